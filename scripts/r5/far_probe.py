"""GPU experiment: the far field of the line kernels (k_line_far) against the direct sum — deviation of the line opacity and of
the flux, step time and per-kernel times with the option off and on.
python scripts/r5/far_probe.py [TAG ...] [--shard=K/N]"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from stardis_amd import synth, _lib, parallel
from stardis_amd.engine import SpectralSynthesizer

tags = [a for a in sys.argv[1:] if not a.startswith("--")] or ["S-c3"]
shard_arg = next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("--shard=")), None)
KERNELS = ("k_classify", "k_prepass_continuum", "k_line_prepass", "k_hlist", "k_far_ranges", "k_line_all", "k_line_wide", "k_line_narrow", "k_line_far", "k_raytrace")


def rel(a, b):
    a, b = np.asarray(a), np.asarray(b)
    m = b != 0
    return float(np.max(np.abs(a[m] - b[m]) / np.abs(b[m]))) if m.any() else 0.0


for tag in tags:
    w = synth.make_workload(tag)
    atm, nus = w["atm"], w["nus"]
    shard = None
    if shard_arg:
        k, n = (int(x) for x in shard_arg.split("/"))
        shard = parallel.balanced_shards(parallel.column_cost(nus, w["lines"]), n)[k]
    out = {}
    for far in (0, 1):
        ctx = _lib.default_context()
        ctx.set_option("far_field", far)
        syn = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], w["thetas"], w["weights"], w["lines"], w["cont"], shard=shard,
                                  track_evaluations=False, keep_line=True)
        syn.enqueue(); syn.synchronize()
        line, F = syn.alpha_line().copy(), syn.F_nu().copy()
        syn.capture()
        syn.step(); syn.synchronize()
        t_end = time.perf_counter() + 0.3
        while time.perf_counter() < t_end:
            for _ in range(10): syn.step()
            syn.synchronize()
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(20): syn.step()
            syn.synchronize()
            best = min(best, (time.perf_counter() - t0) / 20)
        ctx.call("sdx_profile_enable", 1); ctx.call("sdx_profile_reset")
        for _ in range(3): syn.enqueue()
        ctx.synchronize()
        kern = {}
        for name in KERNELS:
            cnt, ms = C.c_int64(), C.c_double()
            _lib.check(ctx.lib.sdx_profile_get(ctx.handle, name.encode(), C.byref(cnt), C.byref(ms)))
            if cnt.value: kern[name] = round(ms.value / 3 * 1e3, 1)
        ctx.call("sdx_profile_enable", 0); ctx.call("sdx_profile_reset")
        syn.close()
        out[far] = (line, F, best, kern)
        print(f"{tag} shard {shard} far_field={far}: step {best * 1e3:.3f} ms   kernels [us] {kern}", flush=True)
        ctx.set_option("far_field", -1)
    print(f"   line opacity: max rel deviation far vs direct {rel(out[1][0], out[0][0]):.2e};  flux {rel(out[1][1][1:], out[0][1][1:]):.2e};  step ratio {out[0][2] / out[1][2]:.2f}x", flush=True)
