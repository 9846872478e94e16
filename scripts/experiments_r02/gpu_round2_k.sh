#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2k; mkdir -p $O
timeout 600 python scripts/strong_scaling_probe.py S-c3 1 > $O/strong_c3.txt 2>&1
for CW in 12 20 30; do
  SDX_CORE_WEIGHT=$CW timeout 600 python scripts/strong_scaling_probe.py S-c3 8 --balanced >> $O/strong_c3.txt 2>&1
done
