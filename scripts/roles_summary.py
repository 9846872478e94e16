"""Per-role counters of k_line_all launched as two kernels (SDX_SPLIT_LAUNCHES=1): python scripts/roles_summary.py DIR
Dispatches alternate wide, narrow; values are per launch."""
import collections, csv, glob, os, sys

root = sys.argv[1]
for tag in ("S-c3", "S-c4m"):
    vals = collections.defaultdict(lambda: collections.defaultdict(list))
    for kind in ("SQ", "FETCH", "WRITE"):
        for f in glob.glob(os.path.join(root, f"{tag}roles_{kind}", "**", "*counter_collection.csv"), recursive=True):
            rows = [r for r in csv.DictReader(open(f)) if "k_line_all" in r["Kernel_Name"]]
            ids = sorted({int(r["Dispatch_Id"]) for r in rows})
            role = {d: ("wide", "narrow")[k % 2] for k, d in enumerate(ids)}
            for r in rows:
                vals[role[int(r["Dispatch_Id"])]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    times = collections.defaultdict(list)
    for f in glob.glob(os.path.join(root, f"{tag}roles_kt", "**", "*kernel_trace.csv"), recursive=True):
        rows = sorted((r for r in csv.DictReader(open(f)) if "k_line_all" in r["Kernel_Name"]), key=lambda r: int(r["Start_Timestamp"]))
        for k, r in enumerate(rows):
            times[("wide", "narrow")[k % 2]].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    print(f"{tag}: k_line_all launched per role (SDX_SPLIT_LAUNCHES=1), per launch")
    for role in ("wide", "narrow"):
        v = {c: sum(x) / len(x) for c, x in vals[role].items()}
        t = " ".join(f"{x:.0f}" for x in times[role])
        print(f"  {role:6s} VALU {v.get('SQ_INSTS_VALU', 0) / 1e6:8.1f} M  SALU {v.get('SQ_INSTS_SALU', 0) / 1e6:8.1f} M  wave-cycles {v.get('SQ_WAVE_CYCLES', 0) / 1e6:8.0f} M  "
              f"wait-issue {v.get('SQ_WAIT_INST_ANY', 0) / 1e6:8.0f} M  waves {v.get('SQ_WAVES', 0):8.0f}  FETCH {v.get('FETCH_SIZE', 0) / 1e3:8.0f} MB  WRITE {v.get('WRITE_SIZE', 0) / 1e3:7.0f} MB  time us [{t}]")
