#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2n; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_configs.py tests/test_gpu_engine.py -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
for T in S-c3 S-c4m; do
  echo "== $T" >> $O/probe.txt
  timeout 400 python scripts/scale_probe.py $T --mixed 2>&1 | grep -E "wall|k_|flux|mixed|Error" >> $O/probe.txt
done
