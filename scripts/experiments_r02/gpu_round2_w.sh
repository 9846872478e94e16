#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2w; mkdir -p $O
for rep in 1 2; do
 for V in "STARDIS_AMD_LIB=$GRAFT_REPO_ROOT/_ab/T/stardis_amd/lib/libstardis_hip.so" "SDX_X=1"; do
  for T in S-c3 S-c4m; do
    echo "== $T ${V:0:18} rep $rep" >> $O/probe.txt
    env $V timeout 400 python scripts/scale_probe.py $T 2>&1 | grep -E "k_line_all|k_hlist|k_prepass|Error" >> $O/probe.txt
  done
 done
done
