#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2m; mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
export SDX_CORE_WEIGHT=20
timeout 600 python scripts/strong_scaling_probe.py S-c3 1 > $O/strong_c3.txt 2>&1
timeout 600 python scripts/strong_scaling_probe.py S-c3 2 4 8 --balanced >> $O/strong_c3.txt 2>&1
