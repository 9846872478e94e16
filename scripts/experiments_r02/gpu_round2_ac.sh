#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2ac; mkdir -p $O
for V in "SDX_X=1" "STARDIS_AMD_LIB=$GRAFT_REPO_ROOT/_ab/W6/stardis_amd/lib/libstardis_hip.so"; do
 for T in S-c3 S-c4m; do
  echo "== $T ${V:0:18}" >> $O/probe.txt
  env $V timeout 400 python scripts/scale_probe.py $T --mixed 2>&1 | grep -E "k_line|mixed|Error" >> $O/probe.txt
  echo "== $T ${V:0:18} split" >> $O/probe.txt
  env $V SDX_SPLIT_LAUNCHES=1 timeout 400 python scripts/scale_probe.py $T --mixed 2>&1 | grep -E "k_line|mixed|Error" >> $O/probe.txt
 done
 env $V timeout 300 python bench.py --steps 200 --warmup 20 --no-secondary --no-cpu-baseline 2>/dev/null | head -c 330 >> $O/probe.txt; echo >> $O/probe.txt
done
timeout 600 python -m pytest tests/test_gpu_hot_faddeeva.py tests/test_gpu_parity.py -m gpu -x -q > $O/pytest_a.log 2>&1
