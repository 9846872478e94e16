"""GPU experiment: step time and parity on a larger workload (default S-c3 with fewer lines).
python scripts/scale_probe.py TAG N_LINES"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stardis_amd import synth, _lib
from stardis_amd.engine import SpectralSynthesizer

tag = sys.argv[1] if len(sys.argv) > 1 else "S-c3"
n_lines = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else None
t0 = time.time()
w = synth.make_workload(tag, n_lines=n_lines)
atm = w["atm"]
print(f"{tag}: N_nu={w['nus'].size} N_l={w['lines']['line_nus'].size} built in {time.time()-t0:.1f}s", flush=True)
syn = SpectralSynthesizer(w["nus"], atm["temperatures"], atm["dist"], w["thetas"], w["weights"], w["lines"], w["cont"])
ctx = syn.ctx
syn.step(); ctx.synchronize()
print("evals", syn.evaluations(), flush=True)
syn.count_evaluations = False
ctx.call("sdx_profile_enable", 1); ctx.call("sdx_profile_reset")
t0 = time.time()
for _ in range(3): syn.enqueue()
ctx.synchronize()
print(f"wall per step {(time.time()-t0)/3*1e3:.2f} ms")
for k in ("k_dnu_partial", "k_classify", "k_prepass_continuum", "k_line_prepass", "k_hlist", "k_line_all", "k_line_wide", "k_line_narrow", "k_total_alphas", "k_raytrace"):
    cnt, ms = C.c_int64(), C.c_double()
    _lib.check(ctx.lib.sdx_profile_get(ctx.handle, k.encode(), C.byref(cnt), C.byref(ms)))
    if cnt.value: print(f"  {k:16s} {ms.value/cnt.value*1e3:10.1f} us")
ctx.call("sdx_profile_enable", 0)
# parity on a strided subset of columns against the CPU oracle (same global window rule)
import oracle
F, total, line = syn.F_nu(), syn.total_alphas(), syn.alpha_line()
cols = np.arange(0, w["nus"].size, max(1, w["nus"].size // 200))
Fr, _ = oracle.raytrace(w["nus"][cols], atm["temperatures"], atm["dist"], w["thetas"], w["weights"], np.ascontiguousarray(total[:, cols]))
print("flux rel err on subset (GPU total -> oracle raytrace):", float(np.max(np.abs(F[1:, cols] - Fr[1:]) / np.abs(Fr[1:]))))
if w["lines"]["line_nus"].size <= 20000:
    t0 = time.time()
    ref = oracle.calc_alan_entries(56, w["nus"], w["lines"]["line_nus"], w["lines"]["doppler_widths"], w["lines"]["gammas"], w["lines"]["alphas"])
    print(f"oracle line opacity {time.time()-t0:.1f}s; rel err", float(np.max(np.abs(line - ref) / np.maximum(np.abs(ref), 1e-300))))
if "--mixed" in sys.argv:
    ctx.set_option("mixed_precision", 1)
    syn.enqueue(); ctx.synchronize()
    ctx.call("sdx_profile_enable", 1); ctx.call("sdx_profile_reset")
    for _ in range(3): syn.enqueue()
    ctx.synchronize()
    for k in ("k_line_all", "k_line_wide", "k_line_narrow", "k_prepass_continuum"):
        cnt, ms = C.c_int64(), C.c_double()
        _lib.check(ctx.lib.sdx_profile_get(ctx.handle, k.encode(), C.byref(cnt), C.byref(ms)))
        if cnt.value: print(f"mixed precision: {k} {ms.value/cnt.value*1e3:.1f} us")
    print(f"mixed precision: line opacity rel diff vs fp64 {float(np.max(np.abs(syn.alpha_line()-line)/np.maximum(np.abs(line),1e-300))):.2e}")
    ctx.call("sdx_profile_enable", 0); ctx.set_option("mixed_precision", 0)
