#!/bin/bash
# round-5 rocprofv3 evidence, every pass through scripts/r4/prof.sh (one process per pass, counters never combined with a trace):
#   bash scripts/r5/profiles_all.sh        (on the GPU box; python scripts/collect_profiles.py r05 copies the summaries into profiles/)
cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r05; mkdir -p $O
for T in S-c2 S-c3 S-c4m; do bash scripts/r4/prof.sh $O $T stats SQ FETCH_SIZE WRITE_SIZE > $O/${T}_table.md 2>&1; done
for T in S-c3 S-c4m; do bash scripts/r4/prof.sh $O $T stats SQ -- --mixed > $O/${T}mixed_table.md 2>&1; done
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/S-c3shard8_stats -- python3 scripts/profile_shard.py S-c3 8 5 20 > $O/S-c3shard8_stats.log 2>&1
python3 scripts/r4/prof_table.py $O > $O/tables.md 2>&1
find $O -name "*.csv" | wc -l
