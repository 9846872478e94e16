"""Can HIP events recorded while a stream is being captured time the kernels of the REPLAYED graph?  (bench.py wants per-kernel
times from the same regime as the step it times.)  python scripts/r5/graph_events_probe.py [TAG]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from stardis_amd import synth, _lib
from stardis_amd.engine import SpectralSynthesizer

tag = sys.argv[1] if len(sys.argv) > 1 else "S-c2"
w = synth.make_workload(tag)
atm, nus = w["atm"], w["nus"]
syn = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], w["thetas"], w["weights"], w["lines"], w["cont"], track_evaluations=False, keep_line=False)
ctx = syn.ctx
KERN = ("k_dnu_partial", "k_classify", "k_prepass_continuum", "k_line_prepass", "k_hlist", "k_line_all", "k_raytrace")


def read():
    out = {}
    for k in KERN:
        cnt, ms = C.c_int64(), C.c_double()
        _lib.check(ctx.lib.sdx_profile_get(ctx.handle, k.encode(), C.byref(cnt), C.byref(ms)))
        if cnt.value: out[k] = round(ms.value / cnt.value * 1e3, 2)
    return out


syn.step(); ctx.synchronize()
# eager, warm
t_end = time.perf_counter() + 0.3
while time.perf_counter() < t_end:
    syn.enqueue()
ctx.synchronize()
ctx.call("sdx_profile_enable", 1); ctx.call("sdx_profile_reset")
for _ in range(20): syn.enqueue()
eager = read()
buf = C.create_string_buffer(64)
ctx.call("sdx_profile_variant", b"k_raytrace", buf, 64)
print("eager events  :", eager, "sum", round(sum(eager.values()), 2), "variant", buf.value.decode())
ctx.call("sdx_profile_reset")
# events recorded during capture
try:
    syn.capture()   # (profiling still on: the event records become graph nodes)
    ctx.call("sdx_profile_enable", 0)
    for _ in range(2000): syn.step()
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(200): syn.step()
    ctx.synchronize()
    step = (time.perf_counter() - t0) / 200 * 1e6
    g = read()
    print("graph events  :", g, "sum", round(sum(g.values()), 2), "step", round(step, 2))
except Exception as e:  # noqa: BLE001
    print("graph events: FAILED", type(e).__name__, e)
