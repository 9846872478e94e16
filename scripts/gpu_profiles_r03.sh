#!/bin/bash
# round-3 rocprofv3 evidence: kernel stats + SQ / FETCH_SIZE / WRITE_SIZE passes (separate runs) for S-c2, S-c3, S-c4m,
# kernel stats + SQ of the fp32-mixed mode at S-c3 / S-c4m, kernel stats of one rank of an 8-way frequency split of S-c3,
# the two roles of the line kernel launched apart (SQ + FETCH / WRITE), the strong-scaling probe
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_r03; mkdir -p $O
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"
for T in S-c2 S-c3 S-c4m; do
  N=40; G="--graph"; if [ $T = S-c2 ]; then N=200; fi; if [ $T = S-c4m ]; then N=20; fi  # enough replays for warm clocks
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}_stats -- python3 scripts/profile_step.py $T $N $G > $O/${T}_stats.log 2>&1
  timeout 600 rocprofv3 --pmc $SQ --output-format csv -d $O/${T}_SQ -- python3 scripts/profile_step.py $T 3 > $O/${T}_SQ.log 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${T}_FETCH_SIZE -- python3 scripts/profile_step.py $T 3 > $O/${T}_FETCH.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${T}_WRITE_SIZE -- python3 scripts/profile_step.py $T 3 > $O/${T}_WRITE.log 2>&1
done
for T in S-c3 S-c4m; do
  N=40; if [ $T = S-c4m ]; then N=20; fi
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}mixed_stats -- python3 scripts/profile_step.py $T $N --mixed --graph > $O/${T}mixed_stats.log 2>&1
  timeout 600 rocprofv3 --pmc $SQ --output-format csv -d $O/${T}mixed_SQ -- python3 scripts/profile_step.py $T 3 --mixed > $O/${T}mixed_SQ.log 2>&1
done
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/S-c3shard8_stats -- python3 scripts/profile_shard.py S-c3 8 5 20 > $O/S-c3shard8_stats.log 2>&1
# the two roles of k_line_all launched apart
export SDX_SPLIT_LAUNCHES=1
for T in S-c3 S-c4m; do
  timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_INST_ANY --output-format csv -d $O/${T}roles_SQ -- python3 scripts/profile_step.py $T 2 > $O/${T}roles_SQ.log 2>&1
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/${T}roles_FETCH -- python3 scripts/profile_step.py $T 2 > $O/${T}roles_FETCH.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/${T}roles_WRITE -- python3 scripts/profile_step.py $T 2 > $O/${T}roles_WRITE.log 2>&1
  timeout 600 rocprofv3 --kernel-trace --output-format csv -d $O/${T}roles_kt -- python3 scripts/profile_step.py $T 3 > $O/${T}roles_kt.log 2>&1
done
unset SDX_SPLIT_LAUNCHES
python3 scripts/roles_summary.py $O > $O/roles.txt 2>&1
timeout 1200 python scripts/strong_scaling_probe.py S-c3 1 2 4 8 --balanced > $O/strong.txt 2>&1
find $O -name "*.csv" | wc -l
