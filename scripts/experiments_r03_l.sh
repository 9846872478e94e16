#!/bin/bash
# round 3, experiment L: XCD-aware dynamic tile queues for the wide role (SDX_WIDE_QUEUE=1) against the static order
export SDX_WIDE_QUEUE=1
python -m pytest tests/test_gpu_engine.py tests/test_gpu_parity.py -x -q 2>&1 | grep -E "passed|failed" | tail -2
for T in S-c2 S-c3 S-c4m; do python scripts/strong_scaling_probe.py $T 1 2>&1 | tail -1; done
python scripts/strong_scaling_probe.py S-c3 8 --balanced 2>&1 | tail -1
unset SDX_WIDE_QUEUE
echo "== static"
for T in S-c2 S-c3 S-c4m; do python scripts/strong_scaling_probe.py $T 1 2>&1 | tail -1; done
python scripts/strong_scaling_probe.py S-c3 8 --balanced 2>&1 | tail -1
