#!/bin/bash
# SQ counters of single ranks of an 8-way split (eager launches): bash scripts/r4/shard_counters.sh TAG RANK...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
T=$1; shift
O=gpurun_out/prof_r04_shards; mkdir -p $O
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"
for R in "$@"; do
  timeout 600 rocprofv3 --pmc $SQ --output-format csv -d $O/${T}rank${R}_SQ -- python3 scripts/profile_shard.py $T 8 $R 3 --eager > $O/${T}rank${R}_SQ.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}rank${R}_stats -- python3 scripts/profile_shard.py $T 8 $R 20 > $O/${T}rank${R}_stats.log 2>&1
done
python3 scripts/r4/prof_table.py $O
