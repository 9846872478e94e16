#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2x; mkdir -p $O
export SDX_SPLIT_LAUNCHES=1
for V in "STARDIS_AMD_LIB=$GRAFT_REPO_ROOT/_ab/T/stardis_amd/lib/libstardis_hip.so" "SDX_X=1"; do
  for T in S-c2 S-c3 S-c4m; do
   for M in "" "--mixed"; do
    echo "== $T ${V:0:18} $M" >> $O/probe.txt
    env $V timeout 400 python scripts/scale_probe.py $T $M 2>&1 | grep -E "k_line|Error" >> $O/probe.txt
   done
  done
done
