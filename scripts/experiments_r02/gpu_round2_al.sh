#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2al; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_engine.py tests/test_gpu_random.py -m gpu -x -q > $O/pytest_a.log 2>&1
for rep in 1 2; do
  timeout 300 python bench.py --steps 400 --warmup 40 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['avg_kernel_ms'])" >> $O/bench.txt
done
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $O/SQ -- python3 scripts/profile_step.py S-c2 3 > $O/SQ.log 2>&1
timeout 300 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/FETCH -- python3 scripts/profile_step.py S-c2 3 > $O/F.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/WRITE -- python3 scripts/profile_step.py S-c2 3 > $O/W.log 2>&1
