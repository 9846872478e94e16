"""Per-kernel times of graph-replayed steps: python scripts/r4/step_times.py [TAG ...] [--parity]
(--parity: flux of a strided subset of columns against the CPU oracle, from the GPU's own total opacity)"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from stardis_amd import synth, _lib
from stardis_amd.engine import SpectralSynthesizer

KERNELS = ("k_dnu_partial", "k_classify", "k_prepass_continuum", "k_line_prepass", "k_hlist", "k_line_all", "k_line_wide", "k_line_narrow", "k_raytrace")
tags = [a for a in sys.argv[1:] if not a.startswith("--")] or ["S-c2"]
for tag in tags:
    w = synth.make_workload(tag)
    atm = w["atm"]
    syn = SpectralSynthesizer(w["nus"], atm["temperatures"], atm["dist"], w["thetas"], w["weights"], w["lines"], w["cont"], track_evaluations=False, keep_line=False)
    ctx = syn.ctx
    syn.capture()
    for _ in range(20): syn.step()
    syn.synchronize()
    best = 1e9
    for rep in range(5):
        n = 200 if w["nus"].size < 20000 else 20
        t0 = time.perf_counter()
        for _ in range(n): syn.step()
        syn.synchronize()
        best = min(best, (time.perf_counter() - t0) / n)
    ctx.call("sdx_profile_enable", 1); ctx.call("sdx_profile_reset")
    for _ in range(5): syn.enqueue()
    ctx.synchronize()
    kern = {}
    for k in KERNELS:
        cnt, ms = C.c_int64(), C.c_double()
        _lib.check(ctx.lib.sdx_profile_get(ctx.handle, k.encode(), C.byref(cnt), C.byref(ms)))
        if cnt.value: kern[k] = round(ms.value / 5 * 1e3, 1)
    ctx.call("sdx_profile_enable", 0); ctx.call("sdx_profile_reset")
    print(f"{tag}: step {best * 1e6:.1f} us  kernels(eager) {kern}", flush=True)
    if "--parity" in sys.argv:
        import oracle
        F, total = syn.F_nu(), syn.total_alphas()
        cols = np.arange(0, w["nus"].size, max(1, w["nus"].size // 400))
        F_ref, _ = oracle.raytrace(w["nus"][cols], atm["temperatures"], atm["dist"], w["thetas"], w["weights"], np.ascontiguousarray(total[:, cols]))
        err = np.max(np.abs(F[1:, cols] - F_ref[1:]) / np.abs(F_ref[1:]))
        print(f"   flux vs oracle on {cols.size} columns: {err:.2e}", flush=True)
    syn.close()
