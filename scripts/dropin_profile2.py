"""GPU experiment: time the pieces of Context.upload inside one drop-in call at S-c2."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stardis_amd import synth, _lib
from stardis_amd.radiation_field import RadiationField
from stardis_amd.radiation_field.opacities.opacities_solvers import calc_alphas
from stardis_amd.radiation_field.radiation_field_solvers import raytrace
from stardis_amd.radiation_field.source_functions.blackbody import blackbody_flux_at_nu

cfg = synth.WORKLOADS["S-c2"]
atm = synth.solar_atmosphere()
nus = synth.tracing_grid(cfg["lam0"], cfg["lam1"], cfg.get("R"), cfg.get("step"))
plasma, model, config, arrays = synth.fake_plasma(nus, atm, 2000, synth.SEED)
log = []
orig_upload = _lib.Context.upload
def upload(self, array, dtype=np.float64):
    t0 = time.perf_counter()
    host = np.ascontiguousarray(array, dtype=dtype)
    t1 = time.perf_counter()
    a = _lib.DeviceArray(self, host.shape, dtype)
    t2 = time.perf_counter()
    _lib.check(self.lib.sdx_memcpy_h2d(self.handle, a.ptr, host.ctypes.data, host.nbytes))
    t3 = time.perf_counter()
    log.append((host.nbytes, (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3))
    return a
_lib.Context.upload = upload
def one():
    field = RadiationField(nus.copy(), blackbody_flux_at_nu, model, synth.N_THETAS)
    calc_alphas(plasma, model, field, config.opacity)
    raytrace(model, field)
for _ in range(3): one()
log.clear()
t0 = time.perf_counter(); one(); print("one call ms", (time.perf_counter() - t0) * 1e3)
for k, r in enumerate(log):
    if max(r[1:]) > 0.3: print(k, "bytes %d  ascontig %.2f  alloc %.2f  memcpy+sync %.2f ms" % r)
print("uploads", len(log), "sum ms", sum(sum(r[1:]) for r in log))
