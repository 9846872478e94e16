"""GPU: graph-replayed step time and line-kernel time of a workload in the fp32-mixed mode.  python scripts/mixed_time.py TAG..."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stardis_amd import synth, _lib
from stardis_amd.engine import SpectralSynthesizer

for tag in sys.argv[1:] or ["S-c3"]:
    w = synth.make_workload(tag)
    atm = w["atm"]
    syn = SpectralSynthesizer(w["nus"], atm["temperatures"], atm["dist"], w["thetas"], w["weights"], w["lines"], w["cont"], track_evaluations=False, keep_line=False)
    ctx = syn.ctx
    ctx.set_option("mixed_precision", 1)
    syn.capture()
    for _ in range(10): syn.step()
    syn.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): syn.step()
    syn.synchronize()
    ms = (time.perf_counter() - t0) / 20 * 1e3
    ctx.call("sdx_profile_enable", 1); ctx.call("sdx_profile_reset")
    for _ in range(3): syn.enqueue()
    ctx.synchronize()
    cnt, t = C.c_int64(), C.c_double()
    _lib.check(ctx.lib.sdx_profile_get(ctx.handle, b"k_line_all", C.byref(cnt), C.byref(t)))
    ctx.call("sdx_profile_enable", 0)
    print(f"{tag} mixed: step {ms:.3f} ms, k_line_all {t.value / 3 * 1e3:.1f} us", flush=True)
    syn.close()
    ctx.set_option("mixed_precision", 0)
