#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2bc; mkdir -p $O
python scripts/dropin_profile.py S-c1 2>&1 | head -12 > $O/dropin.txt
python scripts/dropin_profile.py S-c2 2>&1 | head -30 >> $O/dropin.txt
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
timeout 1200 python bench.py > $O/bench_full.json 2> $O/bench_full.err
