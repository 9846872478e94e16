"""GPU experiment: the formal solution's kernel time inside the fused step (eager, HIP events; scripts/ab_libs.sh alternates two builds).
python scripts/r6/rt_ab.py [TAG ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import bench
from stardis_amd import synth, _lib
from stardis_amd.engine import SpectralSynthesizer

for tag in (sys.argv[1:] or ["S-c3"]):
    w = synth.make_workload(tag)
    atm = w["atm"]
    ctx = _lib.Context(0)
    syn = SpectralSynthesizer(w["nus"], atm["temperatures"], atm["dist"], w["thetas"], w["weights"], w["lines"], w["cont"], ctx=ctx, track_evaluations=False, keep_line=False)
    for _ in range(5):
        syn.step()
    ctx.synchronize()
    best = None
    for _ in range(3):
        k = bench.kernel_times(ctx, syn, 20)
        best = k if best is None or k["k_raytrace"] < best["k_raytrace"] else best
    import hashlib
    print(tag, {a: round(b * 1e3, 1) for a, b in best.items() if b}, "F sha", hashlib.sha256(np.ascontiguousarray(syn.F_nu()[-1]).tobytes()).hexdigest()[:12], flush=True)
    syn.close(); ctx.close()
