#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2ap; mkdir -p $O
for rep in 1 2; do
 for V in "SDX_X=1" "SDX_RT_RECIP_KERNEL=1"; do
  echo "== $V" >> $O/bench.txt
  env $V timeout 300 python bench.py --steps 400 --warmup 40 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['avg_kernel_ms'])" >> $O/bench.txt
 done
done
bash scripts/gpu_profiles_r02.sh > $O/prof.log 2>&1
