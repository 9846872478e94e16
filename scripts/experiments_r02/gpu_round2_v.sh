#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2v; mkdir -p $O
for rep in 1 2; do
 for V in "SDX_NO_HSCAN=1" "SDX_X=1"; do
  for T in S-c3 S-c4m; do
    echo "== $T $V rep $rep" >> $O/probe.txt
    env $V timeout 400 python scripts/scale_probe.py $T 2>&1 | grep -E "k_line_all|k_hlist|Error" >> $O/probe.txt
  done
 done
done
timeout 300 python bench.py --steps 200 --warmup 20 --no-secondary --no-cpu-baseline > $O/bench.json 2> $O/bench.err
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/c2_WRITE -- python3 scripts/profile_step.py S-c2 3 > $O/c2_write.log 2>&1
