#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2z; mkdir -p $O
timeout 300 python scripts/mixed_debug.py S-c2 > $O/mixed.txt 2>&1
timeout 300 python scripts/mixed_debug.py S-c3 20000 >> $O/mixed.txt 2>&1
for ORD in 0 1 2; do
 for T in S-c3 S-c4m; do
  echo "== $T order $ORD" >> $O/probe.txt
  SDX_NARROW_ORDER=$ORD SDX_SPLIT_LAUNCHES=1 timeout 400 python scripts/scale_probe.py $T 2>&1 | grep -E "k_line|Error" >> $O/probe.txt
  SDX_NARROW_ORDER=$ORD timeout 400 python scripts/scale_probe.py $T 2>&1 | grep -E "k_line|Error" >> $O/probe.txt
 done
done
