#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2o; mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
for T in S-c3 S-c4m; do
  echo "== $T" >> $O/probe.txt
  timeout 400 python scripts/scale_probe.py $T --mixed 2>&1 | grep -E "wall|k_|flux|mixed|Error" >> $O/probe.txt
done
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary 2>$O/bench.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_kernel_ms'])" >> $O/probe.txt
timeout 600 python scripts/strong_scaling_probe.py S-c3 1 > $O/strong_c3.txt 2>&1
timeout 600 python scripts/strong_scaling_probe.py S-c3 8 --balanced >> $O/strong_c3.txt 2>&1
