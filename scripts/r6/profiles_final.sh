#!/bin/bash
# round 6, last kernels: the rocprofv3 passes of scripts/r6/evidence.sh alone (every pass its own process through scripts/r4/prof.sh;
# counters never combined with a trace).  bash scripts/r6/profiles_final.sh   (on the GPU box; then python scripts/collect_profiles.py r06)
cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r06; rm -rf $O; mkdir -p $O
for T in S-c2 S-c3 S-c4m; do bash scripts/r4/prof.sh $O $T stats SQ FETCH_SIZE WRITE_SIZE > $O/${T}_table.md 2>&1; done
for T in S-c3 S-c4m; do bash scripts/r4/prof.sh $O $T stats SQ -- --mixed > $O/${T}mixed_table.md 2>&1; done
bash scripts/r4/prof.sh $O S-c4m stats SQ FETCH_SIZE WRITE_SIZE -- --linelist > $O/S-c4m-linelist_table.md 2>&1
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/S-c3shard8_stats -- python3 scripts/profile_shard.py S-c3 8 5 20 > $O/S-c3shard8_stats.log 2>&1
python3 scripts/r4/prof_table.py $O > $O/tables.md 2>&1
find $O -name "*.csv" | wc -l
