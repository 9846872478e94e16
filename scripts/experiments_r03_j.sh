#!/bin/bash
# round 3, experiment J: line subsets per workgroup at S-c2 after the cheaper far-wing loop (SDX_WIDE_BLOCKS: target workgroups)
for B in 2560 5000 6720; do echo "=== SDX_WIDE_BLOCKS=$B"; SDX_WIDE_BLOCKS=$B python bench.py --no-secondary --no-cpu-baseline --steps 200 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['avg_kernel_ms'])"; done
