#!/bin/bash
# the seeded fuzzers with the far field forced on whatever the grid (SDX_FAR=1): bash scripts/r5/fuzz_far.sh OUT
cd $GRAFT_REPO_ROOT
export SDX_EXPERIMENT=1 SDX_FAR=1
out=$1; : > $out
python scripts/fuzz_random_cases.py 1000 1400 >> $out 2>&1
python scripts/fuzz_long_lists.py 200 360 >> $out 2>&1
python scripts/fuzz_long_lists.py 200 240 --mixed >> $out 2>&1
python scripts/fuzz_long_lists.py 300 330 --ticket >> $out 2>&1
python scripts/fuzz_linelist.py 100 160 >> $out 2>&1
SDX_GROUP_LOOPBACK=1 python scripts/fuzz_group_loopback.py 100 130 >> $out 2>&1
python scripts/fuzz_fused_dropin.py 0 40 >> $out 2>&1
grep -c " ok" $out; grep "failures" $out
