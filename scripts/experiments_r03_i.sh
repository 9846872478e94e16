#!/bin/bash
# round 3, experiment I: the fp32 narrow walk with F frequencies per wave (mixed-precision mode)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_configs.py tests/test_gpu_hot_faddeeva.py tests/test_gpu_engine.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -4
python scripts/fuzz_random_cases.py 24 64 2>&1 | tail -2
for F in 0 1; do
  echo "=== SDX_NARROW_F=$F (0 = auto)"
  for T in S-c3 S-c4m; do
    N=40; if [ $T = S-c4m ]; then N=20; fi
    O=gpurun_out/prof_r03i_${T}_$F; rm -rf $O; mkdir -p $O
    SDX_NARROW_F=$F timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 scripts/profile_step.py $T $N --mixed --graph > $O.log 2>&1
    python3 - <<PY
import csv, glob
for f in glob.glob("$O/**/*kernel_stats.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k_line_all" in r["Name"]: print("$T mixed", r["Name"][:40], "avg us", round(float(r["AverageNs"])/1e3, 1), "calls", r["Calls"])
PY
  done
done
