"""GPU experiment for the f1 path: the same synthesis fed with per-line scalars (parameters generated in the pre-pass)
and with the three dense (N_l, N_d) tables (made from the same scalars on the device, downloaded, re-uploaded as the
reference layout).  Prints input bytes, upload time, per-kernel times and the flux difference (expected: 0).
python scripts/linelist_probe.py TAG [N_LINES]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stardis_amd import synth, _lib, linelist as LL
from stardis_amd.engine import SpectralSynthesizer

tag = sys.argv[1] if len(sys.argv) > 1 else "S-c2"
cfg = synth.WORKLOADS[tag]
n_lines = int(sys.argv[2]) if len(sys.argv) > 2 else cfg["n_lines"]
atm = synth.solar_atmosphere()
nus = synth.tracing_grid(cfg["lam0"], cfg["lam1"], cfg.get("R"), cfg.get("step"))
thetas, weights = synth.thetas_and_weights()
cont = synth.synth_continuum_state(atm)
spec = synth.synth_linelist(nus, atm, n_lines)
ctx = _lib.default_context()
print(f"{tag}: N_nu={nus.size} N_l={n_lines}; line list {spec.bytes_per_line()} B/line = {spec.bytes_per_line()*n_lines/1e6:.1f} MB, "
      f"dense tables {24*spec.n_depth} B/line = {24*spec.n_depth*n_lines/1e6:.1f} MB", flush=True)


def kernel_times(syn, reps=3):
    ctx.call("sdx_profile_enable", 1); ctx.call("sdx_profile_reset")
    t0 = time.time()
    for _ in range(reps): syn.enqueue()
    ctx.synchronize()
    wall = (time.time() - t0) / reps * 1e3
    out = {}
    for k in ("k_dnu_partial", "k_prepass_continuum", "k_build_lists", "k_line_all", "k_raytrace"):
        cnt, ms = C.c_int64(), C.c_double()
        _lib.check(ctx.lib.sdx_profile_get(ctx.handle, k.encode(), C.byref(cnt), C.byref(ms)))
        if cnt.value: out[k] = ms.value / cnt.value * 1e3
    ctx.call("sdx_profile_enable", 0)
    return wall, out


t0 = time.time(); dev = spec.upload(ctx); ctx.synchronize(); t_up_list = time.time() - t0
syn = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], thetas, weights, dev, cont)
syn.enqueue(); ctx.synchronize()
ev = syn.evaluations(); syn.count_evaluations = False
wall, k = kernel_times(syn)
print(f"generated: upload {t_up_list*1e3:.1f} ms, step {wall:.3f} ms, evaluations {ev}, kernels us:", {a: round(b, 1) for a, b in k.items()}, flush=True)
F_gen = syn.F_nu()

t0 = time.time(); a, g, d = LL.line_params(dev, ctx); t_tab = time.time() - t0
dense = dict(line_nus=spec.nu, doppler_widths=d, gammas=g, alphas=a)
t0 = time.time()
syn2 = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], thetas, weights, dense, cont)
ctx.synchronize(); t_up_dense = time.time() - t0
syn2.enqueue(); ctx.synchronize(); syn2.count_evaluations = False
wall2, k2 = kernel_times(syn2)
print(f"dense:     upload {t_up_dense*1e3:.1f} ms (tables formed + downloaded in {t_tab*1e3:.0f} ms), step {wall2:.3f} ms, kernels us:",
      {a_: round(b, 1) for a_, b in k2.items()}, flush=True)
F_dense = syn2.F_nu()
print("flux identical:", bool(np.array_equal(F_gen, F_dense)), " max rel diff", float(np.max(np.abs(F_gen[1:] - F_dense[1:]) / np.abs(F_dense[1:]))))
