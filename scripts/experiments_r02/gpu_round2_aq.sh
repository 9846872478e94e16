#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2aq; mkdir -p $O
timeout 1200 python bench.py > $O/bench_full.json 2> $O/bench_full.err
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err
timeout 900 python scripts/strong_scaling_probe.py S-c3 1 2 4 8 --balanced > $O/strong.txt 2>&1
timeout 600 python bench.py --gpus 2 --steps 100 --warmup 10 --no-cpu-baseline --no-secondary > $O/bench_2.json 2> $O/bench_2.err
