#!/bin/bash
# final evidence of the round: profiles, then the full bench line, the driver-style short run and the strong-scaling probe
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2final; mkdir -p $O
bash scripts/gpu_profiles_r02.sh > $O/prof.log 2>&1
export SDX_SPLIT_LAUNCHES=1
for M in "" "--mixed"; do
  timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY --output-format csv -d $O/roles_SQ$M -- python3 scripts/profile_step.py S-c3 2 $M > $O/roles_SQ$M.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/roles_kt$M -- python3 scripts/profile_step.py S-c3 3 $M > $O/roles_kt$M.log 2>&1
done
unset SDX_SPLIT_LAUNCHES
timeout 900 python scripts/strong_scaling_probe.py S-c3 1 2 4 8 --balanced > $O/strong.txt 2>&1
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench_20.json 2> $O/bench_20.err
