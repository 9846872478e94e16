#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2as; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_engine.py tests/test_gpu_multi.py tests/test_gpu_configs.py -m gpu -x -q > $O/pytest_a.log 2>&1
timeout 900 python scripts/strong_scaling_probe.py S-c3 1 2 4 8 --balanced >> $O/strong.txt 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/shard_stats -- python3 scripts/profile_shard.py S-c3 8 5 20 > $O/shard_stats.log 2>&1
