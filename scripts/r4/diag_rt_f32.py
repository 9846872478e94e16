"""Where the fp32 formal solution of the mixed mode leaves its tolerance on the random columns of scripts/fuzz_raytrace.py:
python scripts/r4/diag_rt_f32.py SEED...   (fp64 and fp32 kernels on the GPU, worst columns with their optical depths)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from stardis_amd import ops, synth
from stardis_amd._lib import default_context

ctx = default_context()
for seed in map(int, sys.argv[1:]):
    rng = np.random.default_rng(31000 + seed)
    n_depth = int(rng.choice([2, 3, 5, 9, 30, 56, 57, 64, 65, 90, 130, 200]))
    n_theta = int(rng.choice([1, 2, 3, 7, 20, 21, 33, 64, 65, 70, 140]))
    n_nu = int(rng.choice([1, 2, 5, 63, 64, 65, 300, 1500, 6000]))
    n_nu = max(1, min(n_nu, int(1.5e6 // (n_depth * n_theta))))
    spherical = bool(rng.random() < 0.3); track = bool(rng.random() < 0.4); accumulate = bool(rng.random() < 0.3) and not spherical
    temps = np.sort(rng.uniform(2500.0, 12000.0, n_depth))
    if rng.random() < 0.5:
        temps = temps[::-1].copy()
    dist = rng.uniform(2e5, 4e7, n_depth - 1)
    nus = np.sort(rng.uniform(2.5e14, 1.2e15, n_nu))[::-1].copy()
    regime = rng.integers(0, 5, n_nu)
    lo = np.choose(regime, [-30.0, -16.0, -9.0, -4.5, -16.0]); hi = np.choose(regime, [-30.0, -13.0, -5.0, -2.0, -2.0])
    alphas = 10.0 ** rng.uniform(lo, hi, (n_depth, n_nu)); alphas[:, regime == 0] = 0.0
    if rng.random() < 0.3 and n_depth > 3:
        alphas[int(rng.integers(0, n_depth)), :] = 0.0
    th, w = synth.thetas_and_weights(n_theta)
    rd = dist.reshape(-1, 1) / np.cos(th)
    F, _ = ops.raytrace_arrays(nus, temps, rd, w, alphas)
    ctx.set_option("mixed_precision", 1)
    try:
        F32, _ = ops.raytrace_arrays(nus, temps, rd, w, alphas)
    finally:
        ctx.set_option("mixed_precision", 0)
    e = np.abs(np.nan_to_num(F32) - np.nan_to_num(F)) / np.maximum(np.abs(np.nan_to_num(F)).max(axis=0, keepdims=True), 1e-300)
    col_err = e.max(axis=0)
    print(f"seed {seed}: depth {n_depth} theta {n_theta} nu {n_nu}; temps {temps[0]:.0f}..{temps[-1]:.0f}; worst columns:")
    for c in np.argsort(-col_err)[:4]:
        mean = np.sqrt(alphas[1:, c] * alphas[:-1, c]); tau = mean * dist
        d = int(np.argmax(e[:, c]))
        print(f"   column {c} regime {regime[c]} nu {nus[c]:.3e}: error {col_err[c]:.2e} at depth {d}; tau along the vertical ray {tau.min():.1e} .. {tau.max():.1e}; "
              f"F64 there {F[d, c]:.3e} F32 {F32[d, c]:.3e} column max {np.abs(F[:, c]).max():.3e}; zero layers {int((alphas[:, c] == 0).sum())}")
    by = [float(col_err[regime == k].max()) if (regime == k).any() else 0.0 for k in range(5)]
    print("   worst error by regime (0 transparent, 1 thin, 2 moderate, 3 thick, 4 mixed):", ["%.1e" % x for x in by])
