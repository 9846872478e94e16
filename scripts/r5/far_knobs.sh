#!/bin/bash
# far-field probe under experiment-knob settings: scripts/r5/far_knobs.sh OUT "TAGS [--shard=K/N]" "VAR=.. VAR=.." ...
out=$1; args=$2; shift 2
for v in "$@"; do
  echo "=== [$v] $args" >> "$out"
  env SDX_EXPERIMENT=1 $v python scripts/r5/far_probe.py $args 2>&1 | grep "far_field=1" >> "$out"
done
