#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2g; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_parity.py tests/test_gpu_random.py -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
for V in "SDX_RECMODE=0" "SDX_RECMODE=1"; do
  echo "== $V" >> $O/bench_variants.txt
  env $V timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2>>$O/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'])" >> $O/bench_variants.txt
  for T in S-c3 S-c4m; do
    echo "== $T $V" >> $O/bench_variants.txt
    env $V timeout 400 python scripts/scale_probe.py $T 2>&1 | grep -E "wall|k_|flux|mixed|Error" >> $O/bench_variants.txt
  done
done
for M in 0 1; do
  export SDX_RECMODE=$M
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_c3_$M -- python3 scripts/profile_step.py S-c3 2 > $O/pmc_c3_$M.log 2>&1
done
