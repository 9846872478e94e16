"""GPU experiment: the hot Voigt routine (voigt_term: what the line kernels evaluate per (line, depth, frequency)) and its fp32 twin
on 4 million random arguments — |x| log-uniform in 1e-6 .. 1e7 and uniform near the region boundaries, y log-uniform in 1e-12 .. 1e3 —
against the oracle's faddeeva (voigt.py:17-86, CPython complex rules), per Humlicek region.  Points within 1e-9 of a region boundary
are listed apart (x = delta_nu * (1 / doppler) may fall on the other side of it, a 1e-5 step of the approximation itself).
python scripts/fuzz_faddeeva.py [SEED]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from stardis_amd import ops

SQRT_PI = np.sqrt(np.pi)
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
rng = np.random.default_rng(seed)
n = 1 << 20
xs, ys = [], []
xs.append(10.0 ** rng.uniform(-6, 7, n) * rng.choice([-1.0, 1.0], n)); ys.append(10.0 ** rng.uniform(-12, 3, n))
xs.append(rng.uniform(-16, 16, n)); ys.append(10.0 ** rng.uniform(-6, 1.2, n))  # all four regions densely
s = rng.choice([15.0, 5.5], n) + rng.normal(0.0, 1e-3, n); yy = rng.uniform(0.0, 1.0, n) * s
xs.append((s - yy) * rng.choice([-1.0, 1.0], n)); ys.append(yy)  # around |x| + y = 15 and 5.5
xx = rng.uniform(0.9, 5.5, n); xs.append(xx * rng.choice([-1.0, 1.0], n)); ys.append(np.maximum(0.195 * xx - 0.176 + rng.normal(0.0, 1e-3, n), 1e-9))  # around the III / IV line
x, y = np.concatenate(xs), np.concatenate(ys)
gamma = y * (SQRT_PI * np.pi)
keep = (gamma / (SQRT_PI * np.pi)) == y  # the pre-pass forms y = (gamma / (sqrt(pi) pi)) / dw: keep arguments it returns bit for bit
x, y, gamma = x[keep], y[keep], gamma[keep]
ref = oracle.faddeeva(x + 1j * y).real
got = ops.voigt_term(x, 1.0, gamma, alpha=SQRT_PI)  # amp = 1 -> Re w
got32 = ops.voigt_term(x, 1.0, gamma, alpha=SQRT_PI, fp32=True)
sa = np.abs(x) + y
near = (np.abs(sa - 15.0) < 1e-9) | (np.abs(sa - 5.5) < 1e-9) | (np.abs(y - (0.195 * np.abs(x) - 0.176)) < 1e-9)
near32 = (np.abs(sa - 15.0) < 1e-4) | (np.abs(sa - 5.5) < 1e-4) | (np.abs(y - (0.195 * np.abs(x) - 0.176)) < 1e-4)
regions = {"I": sa >= 15.0, "II": (sa < 15.0) & (sa >= 5.5), "III": (sa < 5.5) & (y >= 0.195 * np.abs(x) - 0.176), "IV": (sa < 5.5) & (y < 0.195 * np.abs(x) - 0.176)}
rel = np.abs(got - ref) / np.maximum(np.abs(ref), 1e-300)
rel32 = np.abs(got32 - ref) / np.maximum(np.abs(ref), 1e-300)
print(f"seed {seed}: {x.size} arguments, {int(near.sum())} within 1e-9 of a region boundary")
bad = 0
for name, m in regions.items():
    a, b = m & ~near, m & ~near32 & (y > 1e-30)
    e64, e32 = float(rel[a].max()), float(rel32[b].max())
    ok = e64 < 2e-13 and e32 < 2e-5
    bad += not ok
    k = np.argmax(np.where(a, rel, 0))
    print(f"   region {name:3s}: {int(a.sum()):8d} points, fp64 worst {e64:.1e} (at x = {x[k]:.6g}, y = {y[k]:.3g}), fp32 worst {e32:.1e}  {'ok' if ok else 'OUT OF BOUNDS'}")
if near.any():
    print(f"   boundary points: fp64 worst {float(rel[near].max()):.1e} (the approximation's own step between regions is ~1e-5)")
    bad += not float(rel[near].max()) < 1e-4
assert np.isfinite(got).all() and np.isfinite(got32).all()
print("failures:", bad)
