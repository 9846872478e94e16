"""GPU experiment: what bounds the wide role on a small grid (S-c2: 7634 frequencies, 2000 lines)?  The same grid with line lists
of different mixes — weak only (the wide waves scan and never hit), no full-grid lines, only full-grid lines, the default.
SDX_SPLIT_LAUNCHES=1 python scripts/wide_floor_probe.py"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stardis_amd import synth, _lib
from stardis_amd.engine import SpectralSynthesizer

w = synth.make_workload("S-c2")
atm, nus = w["atm"], w["nus"]
for name, mix in (("weak only", (1.0, 0.0, 0.0)), ("weak + medium", (0.9, 0.1, 0.0)), ("weak + full-grid", (0.99, 0.0, 0.01)), ("default", (0.9, 0.09, 0.01)),
                  ("medium only", (0.0, 1.0, 0.0))):
    lines = synth.synth_lines(nus, atm, 2000, seed=synth.SEED, mix=mix)
    syn = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], w["thetas"], w["weights"], lines, w["cont"], track_evaluations=True, keep_line=False)
    ctx = syn.ctx
    if "SDX_INDEXED_MIN" in os.environ: ctx.set_option("indexed_min_lines", int(os.environ["SDX_INDEXED_MIN"]))  # the indexed wide path on a short list
    syn.step()
    ev = syn.evaluations()
    syn.capture()
    for _ in range(50): syn.step()
    syn.synchronize()
    ctx.call("sdx_profile_enable", 1); ctx.call("sdx_profile_reset")
    for _ in range(5): syn.enqueue()
    ctx.synchronize()
    kern = {}
    for k in ("k_line_all", "k_line_wide", "k_line_narrow", "k_raytrace", "k_prepass_continuum", "k_hlist"):
        cnt, ms = C.c_int64(), C.c_double()
        _lib.check(ctx.lib.sdx_profile_get(ctx.handle, k.encode(), C.byref(cnt), C.byref(ms)))
        if cnt.value: kern[k] = round(ms.value / 5 * 1e3, 1)
    ctx.call("sdx_profile_enable", 0)
    syn.close()
    print(f"{name:18s} mix {mix}: evaluations {ev:.3e} {kern}", flush=True)
