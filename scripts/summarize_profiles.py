"""Markdown summary of the committed rocprofv3 files of one round: python scripts/summarize_profiles.py r02
Per workload and kernel: average duration (kernel_stats), waves, VALU / SALU / LDS wave-instructions (SQ pass), FETCH_SIZE and
WRITE_SIZE (separate passes; KB as rocprofv3 reports them)."""
import collections, csv, glob, os, sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")


def short(name):
    name = name.replace("void ", "").replace("sdx::", "")
    return name.split("(")[0]


for stats in sorted(glob.glob(os.path.join(root, f"{rnd}_*_kernel_stats.csv"))):
    tag = os.path.basename(stats)[len(rnd) + 1:-len("_kernel_stats.csv")]
    dur = {short(r["Name"]): (int(r["Calls"]), float(r["AverageNs"]) / 1e3) for r in csv.DictReader(open(stats)) if not r["Name"].startswith("__amd")}
    pmc = collections.defaultdict(lambda: collections.defaultdict(list))
    for c in ("SQ", "FETCH_SIZE", "WRITE_SIZE"):
        path = os.path.join(root, f"{rnd}_{tag}_pmc_{c}.csv")
        if os.path.exists(path):
            for r in csv.DictReader(open(path)):
                if not r["Kernel_Name"].startswith("__amd"):
                    pmc[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    print(f"\n**{tag}**\n")
    print("| kernel | calls | avg µs | waves | VALU M | SALU M | LDS M | wave-cycles M | wait-issue M | FETCH KB | WRITE KB |")
    print("|---|---|---|---|---|---|---|---|---|---|---|")
    for k, (calls, us) in sorted(dur.items(), key=lambda kv: -kv[1][1]):
        p = {c: sum(v) / len(v) for c, v in pmc.get(k, {}).items()}
        f = lambda c, s=1e6, n=2: f"{p[c] / s:.{n}f}" if c in p else ""
        print(f"| `{k}` | {calls} | {us:.1f} | {f('SQ_WAVES', 1, 0)} | {f('SQ_INSTS_VALU')} | {f('SQ_INSTS_SALU')} | {f('SQ_INSTS_LDS')} | {f('SQ_WAVE_CYCLES')} | "
              f"{f('SQ_WAIT_INST_ANY')} | {f('FETCH_SIZE', 1, 0)} | {f('WRITE_SIZE', 1, 0)} |")
