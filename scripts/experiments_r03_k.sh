#!/bin/bash
# round 3, experiment K: weights of the shard cost model after the line kernel got cheaper
for FX in 6000 10000 14000 18000; do for CW in 14 20; do echo "== fixed $FX core_weight $CW"; SDX_FIXED=$FX SDX_CORE_WEIGHT=$CW python scripts/strong_scaling_probe.py S-c3 8 --balanced 2>&1 | tail -1 | cut -c1-120; done; done
