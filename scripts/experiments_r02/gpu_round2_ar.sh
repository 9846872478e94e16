#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2ar; mkdir -p $O
for B in 0 100000000; do
  echo "== SDX_WIDE_BLOCKS=$B" >> $O/strong.txt
  if [ $B = 0 ]; then unset SDX_WIDE_BLOCKS; else export SDX_WIDE_BLOCKS=$B; fi
  timeout 900 python scripts/strong_scaling_probe.py S-c3 1 8 --balanced >> $O/strong.txt 2>&1
  timeout 400 python scripts/scale_probe.py S-c4m 2>&1 | grep -E "k_line_all" >> $O/strong.txt
done
