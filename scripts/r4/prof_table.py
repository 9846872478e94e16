"""Per-kernel table of the rocprofv3 passes under a directory written by scripts/r4/prof.sh: python scripts/r4/prof_table.py DIR"""
import collections, csv, glob, os, sys

root = sys.argv[1]
short = lambda n: n.replace("void ", "").replace("sdx::", "").split("(")[0]
tags = sorted({os.path.basename(d).rsplit("_", 1)[0].replace("_FETCH", "").replace("_WRITE", "") for d in glob.glob(os.path.join(root, "*_*")) if os.path.isdir(d)})
for tag in tags:
    dur, pmc = {}, collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(root, tag + "_stats", "*", "*kernel_stats.csv")):
        dur = {short(r["Name"]): (int(r["Calls"]), float(r["AverageNs"]) / 1e3) for r in csv.DictReader(open(f)) if not r["Name"].startswith("__amd")}
    for c in ("SQ", "FETCH_SIZE", "WRITE_SIZE"):
        for f in glob.glob(os.path.join(root, f"{tag}_{c}", "*", "*counter_collection.csv")):
            for r in csv.DictReader(open(f)):
                if not r["Kernel_Name"].startswith("__amd"):
                    pmc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"\n**{tag}**\n\n| kernel | calls | avg us | waves | VALU M | SALU M | LDS M | wave-cycles M | wait-issue M | FETCH KB | WRITE KB |\n|---|---|---|---|---|---|---|---|---|---|---|")
    for k in sorted(set(dur) | set(pmc), key=lambda k: -dur.get(k, (0, 0))[1]):
        calls, us = dur.get(k, (0, float("nan")))
        p = {c: sum(v) / len(v) for c, v in pmc.get(k, {}).items()}
        f = lambda c, s=1e6, n=2: f"{p[c] / s:.{n}f}" if c in p else ""
        print(f"| `{k}` | {calls} | {us:.1f} | {f('SQ_WAVES', 1, 0)} | {f('SQ_INSTS_VALU')} | {f('SQ_INSTS_SALU')} | {f('SQ_INSTS_LDS')} | {f('SQ_WAVE_CYCLES')} | {f('SQ_WAIT_INST_ANY')} | {f('FETCH_SIZE', 1, 0)} | {f('WRITE_SIZE', 1, 0)} |")
