#!/bin/bash
# round 3, experiment D: region IV of the narrow role — cosine short path (default build), real-recurrence polynomials (-DSDX_R4_REAL_RECURRENCE)
for v in default nocos r4; do
  if [ $v = default ]; then unset STARDIS_AMD_LIB; else export STARDIS_AMD_LIB=$PWD/_ab/libstardis_hip_$v.so; fi
  echo "=== $v"
  python -m pytest tests/test_gpu_hot_faddeeva.py tests/test_gpu_parity.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -4
  python scripts/strong_scaling_probe.py S-c2 1 2>&1 | tail -1
  python scripts/strong_scaling_probe.py S-c3 1 2>&1 | tail -1
done
