#!/bin/bash
# round-5 evidence with the far field of the line kernels: rocprofv3 summaries, emulated 8-way splits, fuzzers with the option forced on
cd $GRAFT_REPO_ROOT
O=gpurun_out/ev_far; mkdir -p $O
bash scripts/r5/profiles_all.sh > $O/profiles_all.log 2>&1
for T in S-c3 S-c4m; do
  python scripts/strong_scaling_probe.py $T 1 2 4 8 --balanced --verbose > $O/${T}_strong_scaling_probe.txt 2>&1
  python scripts/strong_scaling_probe.py $T 1 8 --balanced --verbose --two-collectives > $O/${T}_strong_scaling_probe_two_collectives.txt 2>&1
done
( python scripts/strong_scaling_probe.py S-c3-R5e5 1 8 --balanced; python scripts/strong_scaling_probe.py S-c3-R5e5 8 --balanced --two-collectives; python scripts/strong_scaling_probe.py S-big 1 8 --balanced; python scripts/strong_scaling_probe.py S-big 8 --balanced --two-collectives ) > $O/asymptote_strong_scaling_probe.txt 2>&1
export SDX_EXPERIMENT=1 SDX_FAR=1
( python scripts/fuzz_long_lists.py 0 40; python scripts/fuzz_long_lists.py 0 16 --mixed; python scripts/fuzz_long_lists.py 100 116 --ticket ) > $O/fuzz_long_lists_far.txt 2>&1
python scripts/fuzz_random_cases.py 0 80 > $O/fuzz_random_far.txt 2>&1
python scripts/fuzz_linelist.py 0 24 > $O/fuzz_linelist_far.txt 2>&1
SDX_GROUP_LOOPBACK=1 python scripts/fuzz_group_loopback.py 0 12 > $O/fuzz_group_far.txt 2>&1
grep -c "ok" $O/fuzz_*_far.txt; grep -h "failures" $O/fuzz_*_far.txt
