#!/bin/bash
# round 3, last pass: the whole GPU suite at HEAD, the fp32-mixed profiles again (its narrow walk changed), the driver-style bench
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r03; mkdir -p $O gpurun_out/r3final
python -m pytest tests -m gpu -q 2>&1 | grep -v "^RCCL\|^HIP version\|^ROCm\|^Hostname\|^Librccl" | tail -8 > gpurun_out/r3final/pytest.log
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"
for T in S-c3 S-c4m; do
  N=40; if [ $T = S-c4m ]; then N=20; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}mixed_stats -- python3 scripts/profile_step.py $T $N --mixed --graph > $O/${T}mixed_stats.log 2>&1
  timeout 600 rocprofv3 --pmc $SQ --output-format csv -d $O/${T}mixed_SQ -- python3 scripts/profile_step.py $T 3 --mixed > $O/${T}mixed_SQ.log 2>&1
done
timeout 900 python bench.py > gpurun_out/r3final/bench_full.json 2> gpurun_out/r3final/bench_full.err
timeout 600 python bench.py --steps 20 --warmup 5 > gpurun_out/r3final/bench_20.json 2> gpurun_out/r3final/bench_20.err
timeout 900 python scripts/determinism_soak.py S-c3 500 --mixed > gpurun_out/r3final/soak_mixed.txt 2>&1
tail -3 gpurun_out/r3final/pytest.log; cat gpurun_out/r3final/soak_mixed.txt; cut -c1-300 gpurun_out/r3final/bench_20.json
