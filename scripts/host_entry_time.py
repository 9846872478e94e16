"""GPU: wall time of the host-buffer entry point sdx_synthesize_f64 (every array handed over as numpy memory, F_nu back on the host)
at a workload's full size — the PCIe-inclusive figure of the C boundary.  python scripts/host_entry_time.py [TAG] [--planes]"""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stardis_amd import _lib, synth
from stardis_amd import constants as K
from stardis_amd.group import host_continuum

tag = next((a for a in sys.argv[1:] if not a.startswith("-")), "S-c2")
w = synth.make_workload(tag)
atm, nus, lines, cont, th, wt = w["atm"], w["nus"], w["lines"], w["cont"], w["thetas"], w["weights"]
ctx = _lib.default_context()
nd, n_nu = atm["temperatures"].size, nus.size
keep = []
c = host_continuum(cont, nus, atm["temperatures"], keep)
ray = np.ascontiguousarray(np.asarray(atm["dist"]).reshape(-1, 1) / np.cos(th))
g = np.ascontiguousarray(lines["gammas"]).reshape(lines["line_nus"].size, -1)
p = lambda a: np.ascontiguousarray(a, dtype=np.float64).ctypes.data
arrs = [np.ascontiguousarray(a, dtype=np.float64) for a in (nus, lines["line_nus"], lines["doppler_widths"], g, lines["alphas"], atm["temperatures"], ray, wt)]
F = np.empty((nd, n_nu))
planes = "--planes" in sys.argv
line = np.empty((nd, n_nu)) if planes else None
total = np.empty((nd, n_nu)) if planes else None


def call():
    _lib.check(ctx.lib.sdx_synthesize_f64(ctx.handle, nd, n_nu, arrs[0].ctypes.data, arrs[1].size, arrs[1].ctypes.data, arrs[2].ctypes.data, arrs[3].ctypes.data,
                                          g.shape[1], arrs[4].ctypes.data, C.byref(c), th.size, arrs[5].ctypes.data, arrs[6].ctypes.data, arrs[7].ctypes.data,
                                          line.ctypes.data if planes else None, total.ctypes.data if planes else None, F.ctypes.data, None))


for _ in range(5): call()
ts = []
for _ in range(30):
    t0 = time.perf_counter(); call(); ts.append(time.perf_counter() - t0)
nbytes_in = sum(a.nbytes for a in arrs)
print(f"{tag}: sdx_synthesize_f64 min {min(ts) * 1e3:.3f} ms median {sorted(ts)[15] * 1e3:.3f} ms; {nbytes_in / 1e6:.1f} MB in, {F.nbytes * (3 if planes else 1) / 1e6:.1f} MB out; "
      f"{n_nu * nd / min(ts):.3e} spectral points/s")
