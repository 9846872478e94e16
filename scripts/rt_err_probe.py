"""GPU experiment: intensity / flux error of the formal solution against the oracle on random columns, rough and smooth.
SDX_RT_SEG=0|1 selects the kernel.  python scripts/rt_err_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from stardis_amd import ops, synth

def err(a, b):
    m = b != 0
    return float(np.max(np.abs(a[m] - b[m]) / np.abs(b[m])))

for kind in ("rough", "smooth"):
    n_depth, n_theta, n_nu = 56, 20, 3000
    rng = np.random.default_rng(100 * n_depth + n_theta)
    temps = np.linspace(3900.0, 9500.0, n_depth)
    dist = rng.uniform(2e5, 4e6, n_depth - 1)
    nus = np.linspace(6.5e14, 4.2e14, n_nu)
    if kind == "rough":
        alphas = 10.0 ** rng.uniform(-9.5, -4.5, (n_depth, n_nu)) * np.linspace(0.05, 30.0, n_depth).reshape(-1, 1)
    else:
        alphas = 10.0 ** (np.linspace(-9.5, -4.5, n_depth).reshape(-1, 1) + rng.normal(0.0, 0.15, (n_depth, n_nu)))
    th, w = synth.thetas_and_weights(n_theta)
    rd = dist.reshape(-1, 1) / np.cos(th)
    F, I = ops.raytrace_arrays(nus, temps, rd, w, alphas, track=True)
    ref, Iref = oracle.raytrace(nus, temps, dist, th, w, alphas, track=True)
    e = np.abs(I - Iref) / np.maximum(np.abs(Iref), 1e-300)
    k = np.unravel_index(np.argmax(e), e.shape)
    print(kind, "SEG=" + os.environ.get("SDX_RT_SEG", "auto"), "I err", err(I, Iref), "F err", err(F, ref), "worst at", k, "I", I[k], "ref", Iref[k],
          "max I in column", np.abs(Iref[:, k[1], k[2]]).max())
