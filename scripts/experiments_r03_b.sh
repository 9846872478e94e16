#!/bin/bash
# round 3, experiment B: culled pre-pass prepares margin lines through the gather blocks (xlist)
python -m pytest tests/test_gpu_engine.py tests/test_gpu_group.py tests/test_gpu_multi.py tests/test_gpu_random.py -x -q 2>&1 | grep -E "passed|failed|error" | tail -3
python scripts/strong_scaling_probe.py S-c3 1 2 4 8 --balanced
