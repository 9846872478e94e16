#!/bin/bash
# rocprofv3 evidence for one workload, one parameterised runner (replaces the one-shot experiment scripts of rounds 2 and 3):
#   bash scripts/r4/prof.sh OUTDIR TAG [stats] [SQ] [FETCH_SIZE] [WRITE_SIZE] [-- extra arguments of scripts/profile_step.py]
# kernel stats come from graph replays (warm clocks), counters from 3 eager steps; every pass is its own process, counters are
# never combined with a trace.  Environment (e.g. SDX_EXPERIMENT=1 SDX_SPLIT_LAUNCHES=1) is inherited by the profiled program.
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=$1; T=$2; shift 2
PASSES=(); EXTRA=()
while [ $# -gt 0 ]; do if [ "$1" = "--" ]; then shift; EXTRA=("$@"); break; fi; PASSES+=("$1"); shift; done
[ ${#PASSES[@]} -eq 0 ] && PASSES=(stats SQ FETCH_SIZE WRITE_SIZE)
mkdir -p $O
N=40; case $T in S-c2) N=200;; S-c4m|S-big) N=20;; esac
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY"
SUF=""; for a in "${EXTRA[@]}"; do [ "$a" = "--mixed" ] && SUF="mixed"; [ "$a" = "--linelist" ] && SUF="-linelist"; done
for P in "${PASSES[@]}"; do
  case $P in
    stats) timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/${T}${SUF}_stats -- python3 scripts/profile_step.py $T $N --graph "${EXTRA[@]}" > $O/${T}${SUF}_stats.log 2>&1;;
    SQ) timeout 600 rocprofv3 --pmc $SQ --output-format csv -d $O/${T}${SUF}_SQ -- python3 scripts/profile_step.py $T 3 "${EXTRA[@]}" > $O/${T}${SUF}_SQ.log 2>&1;;
    FETCH_SIZE|WRITE_SIZE) timeout 600 rocprofv3 --pmc $P --output-format csv -d $O/${T}${SUF}_$P -- python3 scripts/profile_step.py $T 3 "${EXTRA[@]}" > $O/${T}${SUF}_$P.log 2>&1;;
  esac
done
python3 scripts/r4/prof_table.py $O
