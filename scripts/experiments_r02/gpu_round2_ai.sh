#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2ai; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
for rep in 1 2; do
 for V in "SDX_X=1" "SDX_RT_SEG=0"; do
  echo "== $V rep $rep" >> $O/bench.txt
  env $V timeout 300 python bench.py --steps 400 --warmup 40 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['avg_kernel_ms'])" >> $O/bench.txt
 done
done
timeout 600 python scripts/strong_scaling_probe.py S-c3 1 2 4 8 --balanced >> $O/strong.txt 2>&1
