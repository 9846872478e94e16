#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2d; mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
for V in "" "SDX_RT_LEGACY=1" "SDX_RT_B=6" "SDX_RT_B=8" "SDX_RT_B=12" "SDX_RT_FPW=3"; do
  echo "== $V" >> $O/bench_variants.txt
  env $V timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2>>$O/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'])" >> $O/bench_variants.txt
done
for V in "" "SDX_RT_LEGACY=1" "SDX_RT_FPW=1" "SDX_RT_FPW=2"; do
  echo "== S-c3 $V" >> $O/bench_variants.txt
  env $V timeout 300 python scripts/scale_probe.py S-c3 2>&1 | grep -E "wall|k_|flux" >> $O/bench_variants.txt
done
timeout 900 python bench.py > $O/bench_full.json 2> $O/bench_full.err
