#!/bin/bash
# strong-scaling probe (emulated 8-way split of S-c3) under experiment knobs: bash scripts/r4/scaling_ab.sh OUT "KNOBS" ["KNOBS" ...]
cd $GRAFT_REPO_ROOT
OUT=$1; shift
: > $OUT
for K in "$@"; do
  echo "== $K" >> $OUT
  env SDX_EXPERIMENT=1 $K python scripts/strong_scaling_probe.py S-c3 1 8 --balanced --verbose >> $OUT 2>&1
done
