#!/bin/bash
# round 3, experiment F: where the line kernel's HBM fetch comes from at S-c4m — the two roles launched apart
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
export SDX_SPLIT_LAUNCHES=1
for T in S-c4m S-c3; do
O=gpurun_out/prof_r03f_$T; mkdir -p $O
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/FETCH -- python3 scripts/profile_step.py $T 3 > $O/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/WRITE -- python3 scripts/profile_step.py $T 3 > $O/write.log 2>&1
python3 - <<PY
import csv, glob, collections
for name in ("FETCH", "WRITE"):
  for f in glob.glob("$O/%s/**/*counter_collection.csv" % name, recursive=True):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name + "_SIZE": acc[r["Kernel_Name"][:70]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        if sum(v)/len(v) > 1000: print("$T", name, "KB/launch", k, round(sum(v)/len(v)), "launches", len(v))
PY
done
