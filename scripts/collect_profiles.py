"""Copy the rocprofv3 outputs of scripts/gpu_profiles_r02.sh (gpurun_out/prof_r02/) into profiles/ under the round's names:
python scripts/collect_profiles.py r02 [gpurun_out/prof_r02]"""
import glob, os, shutil, sys

rnd = sys.argv[1] if len(sys.argv) > 1 else "r02"
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = sys.argv[2] if len(sys.argv) > 2 else os.path.join(root, "gpurun_out", "prof_" + rnd)
n = 0
for d in sorted(os.listdir(src)):
    path = os.path.join(src, d)
    if not os.path.isdir(path):
        continue
    tag, _, kind = d.rpartition("_") if not d.endswith("_SIZE") else (d[:-len("_FETCH_SIZE")], "_", d[-len("FETCH_SIZE"):]) if d.endswith("FETCH_SIZE") else (d[:-len("_WRITE_SIZE")], "_", "WRITE_SIZE")
    pattern, name = ("*kernel_stats.csv", f"{rnd}_{tag}_kernel_stats.csv") if kind == "stats" else ("*counter_collection.csv", f"{rnd}_{tag}_pmc_{kind}.csv")
    files = glob.glob(os.path.join(path, "*", pattern))
    if files:
        shutil.copy(max(files, key=os.path.getmtime), os.path.join(root, "profiles", name))
        n += 1
        print(name)
print(n, "files")
