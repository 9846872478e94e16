#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r02; mkdir -p $O
for T in S-c3 S-c4m; do
  N=40; if [ $T = S-c4m ]; then N=20; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_stats -- python3 scripts/profile_step.py $T $N --graph > $O/${T}_stats.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}mixed_stats -- python3 scripts/profile_step.py $T $N --mixed --graph > $O/${T}mixed_stats.log 2>&1
done
