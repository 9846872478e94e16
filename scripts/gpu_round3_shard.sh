#!/bin/bash
# after the narrow-role F rule was fixed for shards: the strong-scaling probe, one rank's kernel stats, the 2-rank self-launched bench
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r03; mkdir -p $O
python -m pytest tests/test_gpu_engine.py tests/test_gpu_multi.py tests/test_gpu_group.py -x -q 2>&1 | grep -E "passed|failed" | tail -2
timeout 1200 python scripts/strong_scaling_probe.py S-c3 1 2 4 8 --balanced > $O/strong.txt 2>&1
rm -rf $O/S-c3shard8_stats
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/S-c3shard8_stats -- python3 scripts/profile_shard.py S-c3 8 5 20 > $O/S-c3shard8_stats.log 2>&1
SDX_BENCH_BACKEND=gloo SDX_BENCH_SINGLE_DEVICE=1 timeout 900 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3final/bench_2rank.json 2> gpurun_out/r3final/bench_2rank.err
cat $O/strong.txt
