#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2e; mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
for V in "" "SDX_RT_LEGACY=1"; do
  echo "== $V" >> $O/bench_variants.txt
  env $V timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2>>$O/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'])" >> $O/bench_variants.txt
done
for T in S-c3 S-c4m; do
  echo "== $T" >> $O/bench_variants.txt
  timeout 400 python scripts/scale_probe.py $T --mixed 2>&1 | grep -E "wall|k_|flux|mixed" >> $O/bench_variants.txt
done
# PMC: instruction mix of the S-c2 step, default (k_formal) and legacy raytrace
for V in default legacy; do
  if [ $V = legacy ]; then export SDX_RT_LEGACY=1; else unset SDX_RT_LEGACY; fi
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $O/pmc_$V -- python3 scripts/profile_step.py S-c2 3 > $O/pmc_$V.log 2>&1
done
unset SDX_RT_LEGACY
find $O -name "*counter_collection.csv" | head
