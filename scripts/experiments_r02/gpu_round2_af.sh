#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2af; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
for V in "SDX_X=1" "SDX_RT_LEGACY=1"; do
  echo "== $V" >> $O/bench.txt
  env $V timeout 300 python bench.py --steps 200 --warmup 20 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['avg_kernel_ms'])" >> $O/bench.txt
  env $V timeout 400 python scripts/scale_probe.py S-c3 2>&1 | grep -E "k_raytrace|flux rel" >> $O/bench.txt
done
