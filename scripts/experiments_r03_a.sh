#!/bin/bash
# round 3, experiment A: classification stream as a role of the pre-pass launch; XCD grouping of the wide role's tiles
python -m pytest tests/test_gpu_engine.py tests/test_gpu_group.py tests/test_gpu_multi.py -x -q 2>&1 | tail -5
echo "== overlap (default)"; python scripts/strong_scaling_probe.py S-c3 1 8 --balanced
echo "== no overlap"; SDX_NO_CLASSIFY_OVERLAP=1 python scripts/strong_scaling_probe.py S-c3 8 --balanced
for g in 1 2 4; do echo "== wide group $g"; SDX_WIDE_GROUP=$g python scripts/strong_scaling_probe.py S-c3 1 8 --balanced; done
