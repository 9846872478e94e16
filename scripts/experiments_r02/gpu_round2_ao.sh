#!/bin/bash
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/r2ao; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1
for rep in 1 2; do
  timeout 300 python bench.py --steps 400 --warmup 40 --no-secondary --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'], d['roofline']['avg_kernel_ms'])" >> $O/bench.txt
done
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 scripts/profile_step.py S-c2 200 --graph > $O/stats.log 2>&1
for T in S-c3 S-c4m; do
  echo "== $T" >> $O/probe.txt
  timeout 400 python scripts/scale_probe.py $T --mixed 2>&1 | grep -E "k_line|k_hlist|k_prepass|k_raytrace|mixed|Error|flux rel" >> $O/probe.txt
done
