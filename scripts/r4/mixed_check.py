"""fp32-mixed tolerance path against fp64 at full size: python scripts/r4/mixed_check.py [TAG]  (step times, flux difference)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from stardis_amd import synth, _lib
from stardis_amd.engine import SpectralSynthesizer

tag = sys.argv[1] if len(sys.argv) > 1 else "S-c3"
w = synth.make_workload(tag)
atm = w["atm"]
res = {}
for mode in (0, 1):
    ctx = _lib.Context(0)
    ctx.set_option("mixed_precision", mode)
    syn = SpectralSynthesizer(w["nus"], atm["temperatures"], atm["dist"], w["thetas"], w["weights"], w["lines"], w["cont"], ctx=ctx, track_evaluations=False, keep_line=False)
    syn.capture()
    for _ in range(10): syn.step()
    syn.synchronize()
    best = 1e9
    for _ in range(4):
        t0 = time.perf_counter()
        for _ in range(10): syn.step()
        syn.synchronize()
        best = min(best, (time.perf_counter() - t0) / 10)
    res[mode] = (best, syn.F_nu())
    syn.close(); ctx.close()
F0, F1 = res[0][1], res[1][1]
print(f"{tag}: fp64 step {res[0][0] * 1e3:.3f} ms, mixed {res[1][0] * 1e3:.3f} ms ({res[0][0] / res[1][0]:.2f}x); emergent flux max rel diff {np.max(np.abs(F1[-1] - F0[-1]) / np.abs(F0[-1])):.2e}, "
      f"all depths {np.max(np.abs(F1[1:] - F0[1:]) / np.abs(F0[1:])):.2e}")
