#!/bin/bash
# round 3, experiment G: narrow role with F frequencies per wave; 6 vs 7 waves per SIMD for k_line_all; list-ordered copies on / off
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
python -m pytest tests/test_gpu_engine.py tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_random.py tests/test_gpu_group.py -x -q 2>&1 | grep -E "passed|failed|Error|assert" | tail -4
run() { echo "=== $1"; for T in S-c2 S-c3 S-c4m; do python scripts/strong_scaling_probe.py $T 1 2>&1 | tail -1; done; }
run "default (F auto, 7 waves, compact)"
SDX_NARROW_F=1 run "F=1"
SDX_NARROW_F=2 run "F=2"
SDX_NO_COMPACT=1 run "no compact"
SDX_NO_WLSCAN=1 run "hrec only"
STARDIS_AMD_LIB=$PWD/_ab/libstardis_hip_w6.so run "6 waves"
O=gpurun_out/prof_r03g; mkdir -p $O
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
