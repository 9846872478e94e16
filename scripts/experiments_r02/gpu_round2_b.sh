#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2b; mkdir -p $O
timeout 300 scripts/_build/issue_cost > $O/issue_cost.txt 2>&1
timeout 1500 python -m pytest tests/test_gpu_hot_faddeeva.py tests/test_gpu_configs.py tests/test_gpu_engine.py tests/test_gpu_random.py -q -m gpu > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
