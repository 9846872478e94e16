#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2s; mkdir -p $O
timeout 1800 python -m pytest tests -q -m gpu -x > $O/pytest.log 2>&1; echo "pytest rc=$?" >> $O/pytest.log
timeout 900 python bench.py > $O/bench_full.json 2> $O/bench_full.err
timeout 300 python bench.py --steps 20 --warmup 5 --no-secondary --no-cpu-baseline > $O/bench_20.json 2> $O/bench_20.err
bash scripts/gpu_profiles_r02.sh > $O/profiles.log 2>&1
timeout 600 python scripts/strong_scaling_probe.py S-c3 1 > $O/strong_c3.txt 2>&1
timeout 900 python scripts/strong_scaling_probe.py S-c3 2 4 8 --balanced >> $O/strong_c3.txt 2>&1
