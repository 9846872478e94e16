#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2l; mkdir -p $O
export SDX_CORE_WEIGHT=20
for WB in 105280 210560; do
  echo "== SDX_WIDE_BLOCKS=$WB" >> $O/strong_c3.txt
  SDX_WIDE_BLOCKS=$WB timeout 600 python scripts/strong_scaling_probe.py S-c3 1 >> $O/strong_c3.txt 2>&1
  SDX_WIDE_BLOCKS=$WB timeout 600 python scripts/strong_scaling_probe.py S-c3 8 --balanced >> $O/strong_c3.txt 2>&1
done
