"""Whose error is it?  The formal solution of random columns (the cases of scripts/fuzz_raytrace.py, plane-parallel) three times:
the reference's formulas in 80-bit extended precision (numpy longdouble: the "truth" for these formulas), the double-precision
oracle, and the GPU — errors of the last two against the first, scaled by the largest intensity of the ray / flux of the column.
For 5e-4 <= tau < 50 the reference forms w1 = w0 - tau e^-tau and w2 = 2 w1 - tau^2 e^-tau (radiation_field_solvers/base.py:38-45):
one ulp of exp is amplified by 1 / tau^2 and 1 / tau^3, so ANY double-precision evaluation is ~1e-16 / tau^3 off near tau = 5e-4.
python scripts/r4/rt_truth.py SEED..."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import oracle
from stardis_amd import constants as K, ops, synth

L = np.longdouble


def truth(nus, temps, dist, thetas, weights, alphas, ray_table=None, correction=None):
    """plane-parallel: dist; spherical: ray_table = calculate_spherical_ray(...) (inward sweep first, :141-198) and the
    photospheric correction (:340-344)"""
    nus, temps, alphas = nus.astype(L), temps.astype(L), alphas.astype(L)
    nd, nn, nt = temps.size, nus.size, thetas.size
    # the table is formed in double by the caller (:302-305 / :349-381)
    rd = (dist.reshape(-1, 1) / np.cos(thetas)).astype(L) if ray_table is None else np.asarray(ray_table, dtype=np.float64).astype(L)
    with np.errstate(all="ignore"):
        mean = np.exp((np.log(alphas[1:]) + np.log(alphas[:-1])) * L(0.5))  # (N_g, N_nu)
        tau = mean[:, :, None] * rd[:, None, :]  # (N_g, N_nu, N_theta)
        pre = (2 * L(K.H_CGS) * nus ** 3) / (L(K.C_CGS) ** 2)
        S = pre[None, :] / (np.exp((L(K.H_CGS) * nus)[None, :] / (L(K.K_B_CGS) * temps[:, None])) - 1)
        e = np.exp(-tau)
        w0 = np.where(tau < 5e-4, tau * (1 - tau / 2), np.where(tau < 50, 1 - e, L(1)))
        w1 = np.where(tau < 5e-4, tau ** 2 * (L(0.5) - tau / 3), np.where(tau < 50, (1 - e) - tau * e, L(1)))
        w2 = np.where(tau < 5e-4, tau ** 3 * (L(1) / 3 - tau / 4), np.where(tau < 50, 2 * ((1 - e) - tau * e) - tau ** 2 * e, L(2)))
        I = np.zeros((nd, nn, nt), dtype=L)
        if ray_table is not None:  # the inward sweep; gap 0 wraps to the last gap / depth as the reference's negative index does
            for g in range(nd - 2, -1, -1):
                gm, dm = (g - 1, g - 1) if g > 0 else (nd - 2, nd - 1)
                tg, tm = tau[g], tau[gm]
                sg, sm, sp = S[g][:, None], S[dm][:, None], S[g + 1][:, None]
                second = w1[g] * ((sg - sm) * (tg / tm) - (sg - sp) * (tm / tg)) / (tg + tm)
                third = w2[g] * (((sm - sg) / tm) + ((sp - sg) / tg)) / (tg + tm)
                new = (1 - w0[g]) * I[g + 1] + w0[g] * sg + second + third
                I[g] = np.where((tg == 0) | (tm == 0), I[g + 1], new)
            I[1:] = 0  # (only I[0] of the sweep survives: the outward pass overwrites the other rows)
        for g in range(nd - 2):
            t0, t1 = tau[g], tau[g + 1]
            s0, s1, s2 = S[g][:, None], S[g + 1][:, None], S[g + 2][:, None]
            second = w1[g] * ((s1 - s2) * (t0 / t1) - (s1 - s0) * (t1 / t0)) / (t0 + t1)
            third = w2[g] * (((s2 - s1) / t1) + ((s0 - s1) / t0)) / (t0 + t1)
            new = (1 - w0[g]) * I[g] + w0[g] * s1 + second + third
            I[g + 1] = np.where(t0 == 0, I[g], new)
        g = nd - 2
        t0 = tau[g]
        third = w2[g] * (S[nd - 2][:, None] - S[nd - 1][:, None]) / t0 ** 2
        new = (1 - w0[g]) * I[g] + w0[g] * S[nd - 1][:, None] + third
        I[nd - 1] = np.where(t0 == 0, I[g], new)
    F = (I * weights.astype(L)[None, None, :]).sum(axis=2)
    if correction is not None:
        F = F * L(correction)
    return F, I


def scaled(a, ref):
    a, ref = np.nan_to_num(np.asarray(a, dtype=np.float64)), np.nan_to_num(np.asarray(ref, dtype=np.float64))
    return float(np.max(np.abs(a - ref) / np.maximum(np.abs(ref).max(axis=0, keepdims=True), 1e-300)))


for seed in map(int, sys.argv[1:] if __name__ == "__main__" else []):
    rng = np.random.default_rng(31000 + seed)
    n_depth = int(rng.choice([2, 3, 5, 9, 30, 56, 57, 64, 65, 90, 130, 200]))
    n_theta = int(rng.choice([1, 2, 3, 7, 20, 21, 33, 64, 65, 70, 140]))
    n_nu = int(rng.choice([1, 2, 5, 63, 64, 65, 300, 1500, 6000]))
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
    if spherical or n_depth * n_theta * n_nu > 3e6:
        print(f"seed {seed}: skipped (spherical or large: depth {n_depth} theta {n_theta} nu {n_nu})"); continue
    th, w = synth.thetas_and_weights(n_theta)
    Ft, It = truth(nus, temps, dist, th, w, alphas)
    with np.errstate(all="ignore"):
        Fo, Io = oracle.raytrace(nus, temps, dist, th, w, alphas, track=True)
    Fg, Ig = ops.raytrace_arrays(nus, temps, dist.reshape(-1, 1) / np.cos(th), w, alphas, track=True)
    print(f"seed {seed}: depth {n_depth} theta {n_theta} nu {n_nu}:  oracle vs truth: flux {scaled(Fo, Ft):.1e} intensity {scaled(Io, It):.1e};  "
          f"GPU vs truth: flux {scaled(Fg, Ft):.1e} intensity {scaled(Ig, It):.1e};  GPU vs oracle: flux {scaled(Fg, Fo):.1e} intensity {scaled(Ig, Io):.1e}", flush=True)
