#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2t; mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 300 python bench.py --steps 200 --warmup 20 --no-secondary --no-cpu-baseline > $O/bench.json 2> $O/bench.err
for T in S-c3 S-c4m; do
  echo "== $T" >> $O/probe.txt
  timeout 400 python scripts/scale_probe.py $T --mixed 2>&1 | grep -E "wall|k_|flux|mixed|Error" >> $O/probe.txt
done
timeout 600 python scripts/strong_scaling_probe.py S-c3 1 > $O/strong_c3.txt 2>&1
timeout 900 python scripts/strong_scaling_probe.py S-c3 8 --balanced >> $O/strong_c3.txt 2>&1
