#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2at; mkdir -p $O
for D in -1 0 2 4; do
  echo "== SDX_CONT_DGS=$D" >> $O/bench.txt
  if [ $D = -1 ]; then unset SDX_CONT_DGS; else export SDX_CONT_DGS=$D; fi
  timeout 300 python bench.py --steps 400 --warmup 40 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['avg_kernel_ms'])" >> $O/bench.txt
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats$D -- python3 scripts/profile_step.py S-c2 200 --graph > $O/stats$D.log 2>&1
  python - <<PY >> $O/bench.txt
import csv,glob
fs=glob.glob('$O/stats$D/*/*kernel_stats.csv')
for r in csv.DictReader(open(max(fs))):
    if 'prepass' in r['Name']: print('  prepass in-graph us', round(float(r['AverageNs'])/1e3,1))
PY
done
