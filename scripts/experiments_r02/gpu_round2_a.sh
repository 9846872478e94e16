#!/bin/bash
# round-2 GPU batch A: test suite + baseline numbers at scale + rocprof kernel stats for S-c3 / S-c4m
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2a; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 600 python bench.py --steps 200 --warmup 20 > $O/bench.json 2> $O/bench.err
timeout 600 python scripts/scale_probe.py S-c3 > $O/scale_c3.log 2>&1
timeout 600 python scripts/scale_probe.py S-c4m > $O/scale_c4m.log 2>&1
for T in S-c3 S-c4m; do
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$T -- python3 scripts/profile_step.py $T 5 > $O/prof_$T.log 2>&1
done
find $O -name "*kernel_stats.csv" | head
