"""GPU experiment: a line list far beyond what dense tables allow, through the f1 path, checked by a size-independent
property — the line opacity is linear in the list, so a list must give the sum of its even- and odd-numbered halves.
python scripts/big_linelist_check.py [N_LINES] [TAG]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stardis_amd import synth, _lib, linelist as LL

n_lines = int(float(sys.argv[1])) if len(sys.argv) > 1 else 10_000_000
tag = sys.argv[2] if len(sys.argv) > 2 else "S-c4"
cfg = synth.WORKLOADS[tag]
atm = synth.solar_atmosphere()
nus = synth.tracing_grid(cfg["lam0"], cfg["lam1"], cfg.get("R"), cfg.get("step"))
t0 = time.time()
spec = synth.synth_linelist(nus, atm, n_lines)
print(f"{tag}: N_nu={nus.size}, {n_lines} lines built in {time.time() - t0:.1f} s; list {spec.bytes_per_line() * n_lines / 1e9:.2f} GB, "
      f"dense tables would be {24 * spec.n_depth * n_lines / 1e9:.1f} GB", flush=True)
ctx = _lib.default_context()


def subset(sel):
    kw = {k: getattr(spec, k)[sel] for k in ("g_lo", "atomic_number", "ion_number", "ionization_energy", "upper_energy", "lower_energy",
                                            "A_ul", "stark", "waals")}
    return LL.LineList(spec.nu[sel], spec.e_low_ev[sel], spec.strength[sel], spec.pop_row[sel], spec.pop, spec.mass[sel], spec.temperature,
                       microturbulence=spec.microturbulence, gamma_mode=spec.gamma_mode, flags=spec.flags,
                       electron_density=spec.electron_density, h_density=spec.h_density, **kw)


def run(s, label):
    t0 = time.time()
    out, ev = LL.line_opacity(nus, s, ctx, return_evaluations=True)
    first = time.time() - t0
    dev = s.upload(ctx)
    d_nus = ctx.upload(nus)
    buf = ctx.empty((s.n_depth, nus.size))
    ctx.synchronize()
    t0 = time.time()
    ctx.call("sdx_line_opacity_linelist_dev", s.n_depth, nus.size, d_nus.ptr, 0, nus.size, dev.byref(), buf.ptr, nus.size, 0, None)
    ctx.synchronize()
    print(f"{label}: {s.n_lines} lines, {ev:.3e} evaluations, {time.time() - t0:.3f} s resident ({ev / (time.time() - t0):.3e} evaluations/s); "
          f"first call incl. upload and allocation {first:.2f} s", flush=True)
    return out


full = run(spec, "full list")
even = run(subset(slice(0, None, 2)), "even lines")
odd = run(subset(slice(1, None, 2)), "odd lines")
s = even + odd
rel = np.abs(full - s) / np.maximum(np.abs(s), 1e-300)
print(f"linearity: max |full - (even + odd)| / (even + odd) = {rel.max():.3e}; all finite: {bool(np.isfinite(full).all())}")
free, total = C.c_size_t(), C.c_size_t()
hip = C.CDLL("libamdhip64.so")
hip.hipMemGetInfo(C.byref(free), C.byref(total))
print(f"HBM in use at the end: {(total.value - free.value) / 1e9:.1f} of {total.value / 1e9:.0f} GB")
