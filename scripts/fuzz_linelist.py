"""GPU experiment: line parameters generated on the device (f1: sdx_line_params_dev, the generating pre-pass) on random line lists:
classic or VALD broadening with a random subset of the four mechanisms (or radiation only, or none), 2 - 90 depth points, 1 - 20000
lines, interpolated atmospheres — alpha, gamma and the Doppler width against the oracle's formulas (broadening.py:550-821,
plasma/base.py:178-455), the line opacity from the generated tables against the oracle, and the generating pre-pass against the
dense-input pre-pass fed with the generated tables, whole grid and three shards, bit for bit.
python scripts/fuzz_linelist.py FIRST LAST"""
import os, sys, traceback
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import oracle
import test_gpu_linelist as TL
from conftest import rel_err
from stardis_amd import linelist as LL, synth
from stardis_amd.engine import SpectralSynthesizer, shard_bounds

bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    try:
        rng = np.random.default_rng(64000 + seed)
        atm0 = synth.cool_dwarf_atmosphere() if seed % 2 else synth.solar_atmosphere()
        nd = int(rng.choice([2, 7, 40, 56, 64, 65, 90]))
        x_old, x_new = np.linspace(0.0, 1.0, atm0["temperatures"].size), np.linspace(0.0, 1.0, nd)
        atm = dict(atm0)
        for k in ("temperatures", "r"):
            atm[k] = np.interp(x_new, x_old, atm0[k])
        for k in ("n_e", "n_h"):
            atm[k] = np.exp(np.interp(x_new, x_old, np.log(atm0[k])))
        atm["dist"] = np.diff(atm["r"])
        lam0 = rng.uniform(3500.0, 9000.0)
        nus = synth.tracing_grid(lam0, lam0 + float(rng.choice([2.0, 25.0, 300.0])), step=float(rng.choice([0.005, 0.02, 0.1])))
        n_lines = int(rng.choice([1, 5, 300, 3000, 9000, 20000]))
        mode = int(rng.choice([LL.GAMMA_CLASSIC, LL.GAMMA_VALD, LL.GAMMA_RADIATION_ONLY, LL.GAMMA_ZERO]))
        spec = TL.seeded_list(n_lines, nus, atm, seed=seed, gamma_mode=mode)
        spec.flags = int(rng.integers(0, 16)) if mode in (LL.GAMMA_CLASSIC, LL.GAMMA_VALD) else spec.flags
        a, gm, d = LL.line_params(spec)
        a_ref = oracle.alpha_line_linelist(spec.e_low_ev, spec.g_lo, spec.strength, spec.nu, spec.pop_row, spec.pop, spec.temperature, spec.alpha_coefficient)
        d_ref = oracle.doppler_widths(spec.nu, spec.mass, spec.temperature, spec.microturbulence)
        ea, ed = rel_err(a, a_ref), rel_err(d, d_ref)
        eg = 0.0
        if mode in (LL.GAMMA_CLASSIC, LL.GAMMA_VALD):
            _, g_ref, _ = TL.oracle_tables(spec)
            eg = rel_err(gm, g_ref)
        elif mode == LL.GAMMA_RADIATION_ONLY:
            assert np.array_equal(np.asarray(gm).reshape(-1), np.asarray(spec.A_ul)), "radiation-only gammas are A_ul"
        else:
            assert not np.asarray(gm).any()
        assert ea < 3e-15 and eg < 1e-13 and ed < 1e-15, (ea, eg, ed)
        cont = synth.synth_continuum_state(atm)
        th, w = synth.thetas_and_weights(int(rng.choice([2, 20])))
        dense = dict(line_nus=spec.nu, doppler_widths=d, gammas=gm, alphas=a)
        gen = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, spec, cont)
        gen.enqueue()
        F, line = gen.F_nu(), gen.alpha_line()
        den = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, dense, cont)
        den.enqueue()
        assert np.array_equal(den.alpha_line(), line) and np.array_equal(den.F_nu(), F), "generated against dense inputs"
        assert gen.evaluations() == den.evaluations()
        if nus.size * n_lines <= 3e7:  # (the oracle on the host bounds this part)
            g_full = np.broadcast_to(np.asarray(gm).reshape(n_lines, -1), (n_lines, nd)) if np.asarray(gm).size else gm
            ref = oracle.calc_alan_entries(nd, nus, spec.nu, d, np.ascontiguousarray(g_full), a)
            assert rel_err(line, ref) < 1e-12
        for r in range(3):
            b, c = shard_bounds(nus.size, 3, r)
            s = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, spec, cont, shard=(b, c))
            s.enqueue()
            assert np.array_equal(s.F_nu(), F[:, b:b + c]) and np.array_equal(s.alpha_line(), line[:, b:b + c]), ("shard", r)
            s.close()
        gen.close(); den.close()
        print(f"seed {seed}: ok  depth {nd} nu {nus.size} lines {n_lines} mode {mode} flags {spec.flags}: alpha {ea:.1e} gamma {eg:.1e} doppler {ed:.1e}", flush=True)
    except Exception:
        bad += 1
        print(f"seed {seed}: FAILED", flush=True)
        traceback.print_exc()
print("failures:", bad)
