#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2ad; mkdir -p $O
export SDX_SPLIT_LAUNCHES=1
for M in "" "--mixed"; do
  timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $O/S-c3_SQ$M -- python3 scripts/profile_step.py S-c3 2 $M > $O/S-c3_SQ$M.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/S-c3_kt$M -- python3 scripts/profile_step.py S-c3 3 $M > $O/S-c3_kt$M.log 2>&1
done
unset SDX_SPLIT_LAUNCHES
for B in 2560 6000 12000; do
  echo "== SDX_WIDE_BLOCKS=$B" >> $O/bench.txt
  SDX_WIDE_BLOCKS=$B timeout 300 python bench.py --steps 200 --warmup 20 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['avg_kernel_ms'])" >> $O/bench.txt
done
timeout 900 python -m pytest tests/test_gpu_hot_faddeeva.py tests/test_gpu_parity.py tests/test_gpu_engine.py -m gpu -x -q > $O/pytest_a.log 2>&1
