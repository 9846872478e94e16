#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2h; mkdir -p $O
export SDX_RECMODE=2
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_parity.py tests/test_gpu_random.py tests/test_gpu_configs.py -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
for V in "SDX_RECMODE=2" "SDX_RECMODE=0"; do
  echo "== $V" >> $O/bench_variants.txt
  env $V timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2>>$O/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'])" >> $O/bench_variants.txt
  for T in S-c3 S-c4m; do
    echo "== $T $V" >> $O/bench_variants.txt
    env $V timeout 400 python scripts/scale_probe.py $T --mixed 2>&1 | grep -E "wall|k_|flux|mixed|Error" >> $O/bench_variants.txt
  done
done
