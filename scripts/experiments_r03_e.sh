#!/bin/bash
# round 3, experiment E: list-ordered copies of the huge lines' records and of the wlist scan words (k_compact_lists)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_engine.py tests/test_gpu_configs.py tests/test_gpu_random.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -4
for v in compact gather; do
  if [ $v = gather ]; then export SDX_NO_COMPACT=1; else unset SDX_NO_COMPACT; fi
  echo "=== $v"
  python scripts/strong_scaling_probe.py S-c3 1 2>&1 | tail -1
  python scripts/strong_scaling_probe.py S-c4m 1 2>&1 | tail -1
  O=gpurun_out/prof_r03e_$v; mkdir -p $O
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/FETCH -- python3 scripts/profile_step.py S-c4m 3 > $O/fetch.log 2>&1
  python3 - <<PY
import csv, glob, collections
for f in glob.glob("$O/FETCH/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == "FETCH_SIZE": acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        if sum(v)/len(v) > 1000: print("FETCH_SIZE KB/launch", k, round(sum(v)/len(v)), "launches", len(v))
PY
done
