#!/bin/bash
# final evidence of round 3: the whole GPU test suite, profiles, the full bench line, the driver-style short run, a 2-rank self-launched
# bench (both ranks on this one GPU, gloo), determinism soaks and the fuzz cases
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3final; mkdir -p $O
python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -15 > $O/pytest.log
bash scripts/gpu_profiles_r03.sh > $O/prof.log 2>&1
timeout 900 python bench.py > $O/bench_full.json 2> $O/bench_full.err
timeout 600 python bench.py --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err
SDX_BENCH_BACKEND=gloo SDX_BENCH_SINGLE_DEVICE=1 timeout 900 python bench.py --gpus 2 --steps 20 --warmup 5 --no-cpu-baseline > $O/bench_2rank.json 2> $O/bench_2rank.err
timeout 900 python scripts/determinism_soak.py S-c2 20000 > $O/soak.txt 2>&1
timeout 900 python scripts/determinism_soak.py S-c3 1000 >> $O/soak.txt 2>&1
timeout 900 python scripts/determinism_soak.py S-c4m 300 >> $O/soak.txt 2>&1
timeout 900 python scripts/determinism_soak.py S-c3 500 --mixed >> $O/soak.txt 2>&1
timeout 1500 python scripts/fuzz_random_cases.py 24 224 > $O/fuzz.txt 2>&1
tail -3 $O/pytest.log; tail -4 $O/soak.txt; tail -2 $O/fuzz.txt; cut -c1-400 $O/bench_20.json
