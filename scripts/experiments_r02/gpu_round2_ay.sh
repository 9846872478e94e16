#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2ay; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
for T in S-c3 S-c4m; do
  echo "== $T" >> $O/probe.txt
  timeout 400 python scripts/scale_probe.py $T --mixed 2>&1 | grep -E "k_line|k_hlist|k_prepass|k_raytrace|mixed|Error|flux rel|wall" >> $O/probe.txt
done
timeout 900 python scripts/strong_scaling_probe.py S-c3 1 8 --balanced > $O/strong.txt 2>&1
