#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2ab; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_hot_faddeeva.py tests/test_gpu_configs.py -m gpu -x -q > $O/pytest_a.log 2>&1
timeout 300 python scripts/mixed_debug.py S-c2 > $O/mixed.txt 2>&1
timeout 300 python scripts/mixed_debug.py S-c3 20000 >> $O/mixed.txt 2>&1
for T in S-c3 S-c4m; do
  echo "== $T" >> $O/probe.txt
  timeout 400 python scripts/scale_probe.py $T --mixed 2>&1 | grep -E "k_line|k_hlist|k_prepass|k_raytrace|mixed|Error" >> $O/probe.txt
done
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
