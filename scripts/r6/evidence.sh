#!/bin/bash
# round-6 evidence: rocprofv3 summaries (every pass through scripts/r4/prof.sh), emulated 8-way splits in three modes, fuzzers and the
# determinism soak on the round's kernels.   bash scripts/r6/evidence.sh   (on the GPU box; then python scripts/collect_profiles.py r06)
cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r06; E=gpurun_out/ev_r06; mkdir -p $O $E
for T in S-c2 S-c3 S-c4m; do bash scripts/r4/prof.sh $O $T stats SQ FETCH_SIZE WRITE_SIZE > $O/${T}_table.md 2>&1; done
for T in S-c3 S-c4m; do bash scripts/r4/prof.sh $O $T stats SQ -- --mixed > $O/${T}mixed_table.md 2>&1; done
bash scripts/r4/prof.sh $O S-c4m stats SQ FETCH_SIZE WRITE_SIZE -- --linelist > $O/S-c4m-linelist_table.md 2>&1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/S-c3shard8_stats -- python3 scripts/profile_shard.py S-c3 8 5 20 > $O/S-c3shard8_stats.log 2>&1
python3 scripts/r4/prof_table.py $O > $O/tables.md 2>&1
for T in S-c3 S-c4m; do
  python scripts/strong_scaling_probe.py $T 1 2 4 8 --balanced --verbose > $E/${T}_strong_scaling_probe.txt 2>&1
  python scripts/strong_scaling_probe.py $T 1 8 --balanced --verbose --two-collectives > $E/${T}_strong_scaling_probe_two_collectives.txt 2>&1
  python scripts/strong_scaling_probe.py $T 1 8 --balanced --verbose --in-flight=2 > $E/${T}_strong_scaling_probe_two_in_flight.txt 2>&1
done
( python scripts/fuzz_random_cases.py 6000 6200; python scripts/fuzz_long_lists.py 1200 1300; python scripts/fuzz_long_lists.py 1300 1330 --mixed; python scripts/fuzz_long_lists.py 1330 1350 --ticket; python scripts/fuzz_long_lists.py 1350 1420 --raw ) > $E/fuzz.txt 2>&1
python scripts/fuzz_linelist.py 1000 1150 > $E/fuzz_linelist.txt 2>&1
python scripts/fuzz_continuum.py 0 200 > $E/fuzz_continuum.txt 2>&1
python scripts/fuzz_fused_dropin.py 11000 11400 > $E/fuzz_dropin.txt 2>&1
SDX_EXPERIMENT=1 SDX_GROUP_LOOPBACK=1 python scripts/fuzz_group_loopback.py 2300 2400 > $E/fuzz_group.txt 2>&1
( python scripts/determinism_soak.py S-c2 5000; python scripts/determinism_soak.py S-c3 500; python scripts/determinism_soak.py S-c4m 100; python scripts/determinism_soak.py S-c3 300 --mixed ) > $E/determinism_soak.txt 2>&1
python scripts/r5/far_vs_oracle.py S-c3 S-c4m --stride=53 > $E/far_vs_oracle.txt 2>&1
grep -h -i "fail\|differ" $E/fuzz*.txt $E/determinism_soak.txt | tail -20
find $O -name "*.csv" | wc -l
