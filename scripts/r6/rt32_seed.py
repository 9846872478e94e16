"""GPU experiment: one seed of scripts/fuzz_raytrace.py replayed for the fp32 formal solution of the mixed mode — the worst column, its
optical depths and the flux of both precisions per depth.  python scripts/r6/rt32_seed.py SEED"""
import os, sys
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import numpy as np
from stardis_amd import ops, synth
from stardis_amd._lib import default_context

ctx = default_context()
seed = int(sys.argv[1])
rng = np.random.default_rng(31000 + seed)
n_depth = int(rng.choice([2, 3, 5, 9, 30, 56, 57, 64, 65, 90, 130, 200]))
n_theta = int(rng.choice([1, 2, 3, 7, 20, 21, 33, 64, 65, 70, 140]))
n_nu = int(rng.choice([1, 2, 5, 63, 64, 65, 300, 1500, 6000]))
n_nu = max(1, min(n_nu, int(1.5e6 // (n_depth * n_theta))))
spherical = bool(rng.random() < 0.3)
track = bool(rng.random() < 0.4)
accumulate = bool(rng.random() < 0.3) and not spherical
temps = np.sort(rng.uniform(2500.0, 12000.0, n_depth))
if rng.random() < 0.5:
    temps = temps[::-1].copy()
dist = rng.uniform(2e5, 4e7, n_depth - 1)
nus = np.sort(rng.uniform(2.5e14, 1.2e15, n_nu))[::-1].copy()
regime = rng.integers(0, 5, n_nu)
lo = np.choose(regime, [-30.0, -16.0, -9.0, -4.5, -16.0])
hi = np.choose(regime, [-30.0, -13.0, -5.0, -2.0, -2.0])
alphas = 10.0 ** rng.uniform(lo, hi, (n_depth, n_nu))
alphas[:, regime == 0] = 0.0
if rng.random() < 0.3 and n_depth > 3:
    alphas[int(rng.integers(0, n_depth)), :] = 0.0
th, w = synth.thetas_and_weights(n_theta)
rd = dist.reshape(-1, 1) / np.cos(th)
print("depth", n_depth, "theta", n_theta, "nu", n_nu, spherical, track, accumulate)
F, _ = ops.raytrace_arrays(nus, temps, rd, w, alphas)
ctx.set_option("mixed_precision", 1)
F32, _ = ops.raytrace_arrays(nus, temps, rd, w, alphas)
ctx.set_option("mixed_precision", 0)
fin = np.isfinite(F).all(axis=0)
dev = np.abs(F32 - F) / np.maximum(np.abs(F).max(axis=0, keepdims=True), 1e-300)
dev[:, ~fin] = 0
d, c = np.unravel_index(np.argmax(dev), dev.shape)
print("worst", dev[d, c], "depth", d, "column", c, "regime", regime[c], "nu", nus[c])
print("columns above 1e-5:", int((dev.max(axis=0) > 1e-5).sum()), "of", n_nu, "regimes", np.bincount(regime[dev.max(axis=0) > 1e-5], minlength=5))
a = alphas[:, c]
tau = np.sqrt(a[:-1] * a[1:]) * dist
h, kb, cc = 6.62607015e-27, 1.380649e-16, 2.99792458e10
S = 2 * h * nus[c] ** 3 / cc ** 2 / np.expm1(h * nus[c] / (kb * temps))
np.set_printoptions(linewidth=220, precision=4)
for k in range(n_depth):
    print(k, "T %8.1f" % temps[k], "alpha %.3e" % a[k], "tau(gap k-1, theta 0) %.3e" % (tau[k - 1] if k else 0.0), "S %.4e" % S[k], "F64 %.6e  F32 %.6e  dev %.2e" % (F[k, c], F32[k, c], dev[k, c]))
