#!/bin/bash
# round 3, experiment H: byte half-widths for the narrow records, scan words only for listed lines, F = 8
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_engine.py tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_random.py tests/test_gpu_group.py tests/test_gpu_hot_faddeeva.py tests/test_gpu_linelist.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -4
run() { echo "=== $1"; for T in S-c2 S-c3 S-c4m; do python scripts/strong_scaling_probe.py $T 1 2>&1 | tail -1; done; }
run "default (F auto = 4 at these sizes)"
SDX_NARROW_F=8 run "F=8"
python scripts/strong_scaling_probe.py S-c3 8 --balanced 2>&1 | tail -1
O=gpurun_out/prof_r03h; mkdir -p $O
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/FETCH -- python3 scripts/profile_step.py S-c4m 3 > $O/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/WRITE -- python3 scripts/profile_step.py S-c4m 3 > $O/write.log 2>&1
python3 - <<PY
import csv, glob, collections
for name in ("FETCH", "WRITE"):
  for f in glob.glob("$O/%s/**/*counter_collection.csv" % name, recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name + "_SIZE": acc[r["Kernel_Name"][:70]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        if sum(v)/len(v) > 1000: print("S-c4m", name, "KB/launch", k, round(sum(v)/len(v)), "launches", len(v))
PY
