"""GPU experiment: the same step replayed many times must give the same bits (no atomics in any sum): every 50th replay of the
graph is compared with the first one.  python scripts/determinism_soak.py TAG REPLAYS [--mixed]"""
import os, sys, zlib
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stardis_amd import synth
from stardis_amd.engine import SpectralSynthesizer

tag = sys.argv[1] if len(sys.argv) > 1 else "S-c2"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
w = synth.make_workload(tag)
atm = w["atm"]
syn = SpectralSynthesizer(w["nus"], atm["temperatures"], atm["dist"], w["thetas"], w["weights"], w["lines"], w["cont"], track_evaluations=False)
if "--mixed" in sys.argv:
    syn.ctx.set_option("mixed_precision", 1)
syn.capture()
syn.step(); syn.synchronize()
ref_F, ref_a = syn.F_nu().copy(), syn.total_alphas().copy()
bad = 0
for k in range(1, n + 1):
    syn.step()
    if k % 50 == 0:
        syn.synchronize()
        if not (np.array_equal(syn.F_nu(), ref_F) and np.array_equal(syn.total_alphas(), ref_a)):
            bad += 1
            print("replay", k, "differs")
print(tag, "mixed" if "--mixed" in sys.argv else "fp64", "replays", n, "checks", n // 50, "differing", bad, "crc", zlib.crc32(ref_F.tobytes()))
