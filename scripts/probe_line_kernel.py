"""GPU experiment: line-kernel time by line class (weak / medium / strong) on the S-c2 grid."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stardis_amd import synth, _lib
from stardis_amd._lib import default_context

ctx = default_context()
w = synth.make_workload("S-c2")
atm, nus, lines = w["atm"], w["nus"], w["lines"]
amax = lines["alphas"].max(axis=1)
classes = {"all": amax > -1, "weak(<0.1)": amax < 0.1, "medium": (amax >= 0.1) & (amax < 100), "strong(>=100)": amax >= 100}
d_nus = ctx.upload(nus)
out = ctx.empty((56, nus.size))
ev = ctx.zeros((1,), np.int64)
for name, m in classes.items():
    sub = {k: np.ascontiguousarray(v[m]) for k, v in lines.items()}
    n = int(m.sum())
    d = [ctx.upload(sub[k]) for k in ("line_nus", "doppler_widths", "gammas", "alphas")]
    def run():
        ctx.call("sdx_line_opacity_dev", 56, nus.size, d_nus.ptr, 0, nus.size, n, d[0].ptr, d[1].ptr, d[2].ptr, sub["gammas"].shape[1], d[3].ptr, out.ptr, nus.size, 0, ev.ptr)
    run(); ctx.synchronize()
    ctx.call("sdx_profile_enable", 1); ctx.call("sdx_profile_reset")
    for _ in range(10): run()
    ctx.synchronize()
    res = {}
    for k in ("k_line_prepass", "k_line_wide", "k_line_narrow", "k_reduce_partials"):
        cnt, ms = C.c_int64(), C.c_double()
        _lib.check(ctx.lib.sdx_profile_get(ctx.handle, k.encode(), C.byref(cnt), C.byref(ms)))
        res[k] = ms.value / max(cnt.value, 1) * 1e3
    ctx.call("sdx_profile_enable", 0); ctx.call("sdx_profile_reset")
    e = int(ev.numpy()[0])
    print(f"{name:14s} lines={n:5d} evals={e:10d} prepass={res['k_line_prepass']:7.1f}us wide={res['k_line_wide']:7.1f}us narrow={res['k_line_narrow']:7.1f}us reduce={res['k_reduce_partials']:6.1f}us  -> {e/ max(res['k_line_wide']+res['k_line_narrow'],1e-9)/1e3:8.1f} Gevals/s")
