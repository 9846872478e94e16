#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2q; mkdir -p $O
for RM in 4 8; do
 for T in S-c3 S-c4m; do
  echo "== $T R_MIXED=$RM" >> $O/probe.txt
  SDX_R_MIXED=$RM timeout 400 python scripts/scale_probe.py $T --mixed 2>&1 | grep -E "k_line_all|mixed|Error" >> $O/probe.txt
 done
done
for M in fp64 mixed; do
  FLAG=""; if [ $M = mixed ]; then FLAG="--mixed"; fi
  timeout 300 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $O/pmc_c3_$M -- python3 scripts/profile_step.py S-c3 2 $FLAG > $O/pmc_c3_$M.log 2>&1
done
