#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2i; mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 600 python scripts/strong_scaling_probe.py S-c3 1 2 4 8 --balanced > $O/strong_c3.txt 2>&1
SDX_NO_CULL=1 timeout 600 python scripts/strong_scaling_probe.py S-c3 1 8 --balanced > $O/strong_c3_nocull.txt 2>&1
timeout 300 python bench.py --steps 200 --warmup 20 --no-cpu-baseline --no-secondary > $O/bench.json 2>$O/bench.err
