"""GPU experiment: per-kernel times of the un-fused step (each role in its own launch) for S-c2."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stardis_amd import synth, _lib
from stardis_amd.engine import SpectralSynthesizer
w = synth.make_workload(sys.argv[1] if len(sys.argv) > 1 else "S-c2")
atm = w["atm"]
syn = SpectralSynthesizer(w["nus"], atm["temperatures"], atm["dist"], w["thetas"], w["weights"], w["lines"], w["cont"])
ctx = syn.ctx
for mode in ("enqueue_unfused", "enqueue"):
    fn = getattr(syn, mode)
    fn(); ctx.synchronize()
    ctx.call("sdx_profile_enable", 1); ctx.call("sdx_profile_reset")
    for _ in range(20): fn()
    ctx.synchronize()
    out = {}
    for k in ("k_dnu_partial", "k_prepass_continuum", "k_line_prepass", "k_line_all", "k_reduce_partials", "k_total_alphas", "k_raytrace"):
        cnt, ms = C.c_int64(), C.c_double()
        _lib.check(ctx.lib.sdx_profile_get(ctx.handle, k.encode(), C.byref(cnt), C.byref(ms)))
        if cnt.value: out[k] = round(ms.value / cnt.value * 1e3, 1)
    ctx.call("sdx_profile_enable", 0)
    print(mode, out)
