#!/bin/bash
# Round-5 A/B driver: the emulated 8-way split of a workload under a list of experiment-knob settings, one block per variant.
#   scripts/r5/ab_shards.sh OUT TAG "PROBE FLAGS" "VAR1=.. VAR2=.." "VAR3=.." ...      ("" = the shipped defaults)
out=$1; tag=$2; flags=$3; shift 3
mkdir -p "$(dirname "$out")"
for v in "$@"; do
  echo "=== $tag $flags variant [$v]" >> "$out"
  env SDX_EXPERIMENT=1 $v python scripts/strong_scaling_probe.py "$tag" 8 --balanced --verbose $flags >> "$out" 2>&1
done
