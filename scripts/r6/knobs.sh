#!/bin/bash
# one rank of a split under experiment-knob settings: scripts/r6/knobs.sh OUT "TAG WORLD RANK [...]" "VAR=.. VAR=.." ...
out=$1; args=$2; shift 2
for v in "$@"; do
  env SDX_EXPERIMENT=1 $v python scripts/r6/shard_kernels.py $args 2>&1 | tail -1 >> "$out"
done
