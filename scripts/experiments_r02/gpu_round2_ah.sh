#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2ah; mkdir -p $O
for rep in 1 2 3; do
 for V in "SDX_X=1" "SDX_RT_LEGACY=1"; do
  echo "== $V rep $rep" >> $O/bench.txt
  env $V timeout 300 python bench.py --steps 400 --warmup 40 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['avg_kernel_ms'])" >> $O/bench.txt
 done
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/seg_stats -- python3 scripts/profile_step.py S-c2 200 --graph > $O/seg_stats.log 2>&1
timeout 600 python scripts/strong_scaling_probe.py S-c3 8 --balanced >> $O/strong.txt 2>&1
timeout 600 python -m pytest tests/test_gpu_engine.py tests/test_gpu_multi.py -m gpu -x -q > $O/pytest_a.log 2>&1
