#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2ax; mkdir -p $O
for rep in 1 2 3 4 5; do
  timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['config']['untimed_settle_steps_after_warmup'])" >> $O/bench20.txt
done
timeout 1200 python bench.py > $O/bench_full.json 2> $O/bench_full.err
