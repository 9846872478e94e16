#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2aj; mkdir -p $O
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/shard_stats -- python3 scripts/profile_shard.py S-c3 8 5 20 > $O/shard_stats.log 2>&1
