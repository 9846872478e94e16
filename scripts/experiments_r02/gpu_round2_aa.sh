#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2aa; mkdir -p $O
timeout 300 python scripts/mixed_debug.py S-c2 > $O/mixed.txt 2>&1
timeout 300 python scripts/mixed_debug.py S-c3 20000 >> $O/mixed.txt 2>&1
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
for T in S-c3 S-c4m; do
  echo "== $T" >> $O/probe.txt
  timeout 400 python scripts/scale_probe.py $T --mixed 2>&1 | grep -E "k_line|k_hlist|k_prepass|k_raytrace|mixed|Error" >> $O/probe.txt
done
timeout 300 python bench.py --steps 200 --warmup 20 --no-secondary --no-cpu-baseline > $O/bench.json 2> $O/bench.err
