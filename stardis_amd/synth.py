"""Seeded synthetic inputs for the hot path (SURVEY.md §8d recipe).

No atomic data, TARDIS or line list is available offline, so tests and
``bench.py`` drive the path with inputs derived from the solar MARCS structure
(``data/sun_marcs_columns.json`` holds the T / Depth / Pe / Pg columns of
docs/quickstart/sun.mod, captured by tests/golden/make_golden.py) plus a random
line list whose strength mixture exercises the three regimes of the reference's
window rule (opacities_solvers/base.py:561-575): hw = 10 floor, medium windows,
and lines spanning the whole grid.

numpy only: this module is also imported by the golden-vector generator under
a different interpreter.
"""
import json
import math
import os

import numpy as np

from . import constants as K

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")

# name -> (lambda_start [AA], lambda_stop [AA], R or None, step [AA] or None, n_lines, gamma per depth)
WORKLOADS = {
    "S-c1": dict(lam0=6560.0, lam1=6570.0, step=0.01, n_lines=2000, gamma_per_depth=True),
    "S-c2": dict(lam0=6500.0, lam1=6600.0, R=5.0e5, n_lines=2000, gamma_per_depth=True),
    "S-c3": dict(lam0=3000.0, lam1=10000.0, R=1.0e5, n_lines=150000, gamma_per_depth=True),
    "S-c3-R5e5": dict(lam0=3000.0, lam1=10000.0, R=5.0e5, n_lines=150000, gamma_per_depth=True),  # S-c3's list at five times the resolving power
    "S-c4": dict(lam0=3000.0, lam1=10000.0, R=1.0e5, n_lines=1000000, gamma_per_depth=False),
    "S-big": dict(lam0=3000.0, lam1=10000.0, R=1.0e6, n_lines=1000000, gamma_per_depth=False),
    # BASELINE config 5: the full solar spectrum of config 3, synthesised with the fp32-mixed tolerance path and followed on
    # the device by the instrumental LSF and the rotational kernel (postprocess.DeviceSpectrum)
    "S-c5": dict(lam0=3000.0, lam1=10000.0, R=1.0e5, n_lines=150000, gamma_per_depth=True, mixed_precision=1, lsf_resolution=5.0e4, v_rot_kms=20.0),
    # BASELINE config 4 on the coolest MARCS structure the reference ships (3800 K dwarf): molecular-style list, gamma (N_l, 1)
    "S-c4m": dict(lam0=3000.0, lam1=10000.0, R=1.0e5, n_lines=1000000, gamma_per_depth=False, atmosphere="cool_dwarf"),
}
SEED = 20250926
N_THETAS = 20  # benchmarks/benchmark_config.yml:19


def solar_atmosphere():
    """Per-depth state of the solar MARCS model (docs/quickstart/sun.mod), innermost point first."""
    return marcs_atmosphere("sun_marcs_columns.json")


def cool_dwarf_atmosphere():
    """The reference's own MARCS test model (io/model/tests/data/marcs_test.mod.gz): Teff 3800 K, log g 4.0,
    2771-7713 K — the coolest structure in the tree, standing in for BASELINE's M-dwarf configuration."""
    return marcs_atmosphere("marcs_t3800_g4_columns.json")


def marcs_atmosphere(columns_file):
    """Per-depth state, innermost point first (io/model/marcs.py:45,204 flips MARCS order)."""
    with open(os.path.join(_DATA, columns_file)) as fh:
        col = json.load(fh)
    t = np.asarray(col["t"], dtype=np.float64)[::-1].copy()
    depth = np.asarray(col["depth"], dtype=np.float64)[::-1].copy()
    pe = np.asarray(col["pe"], dtype=np.float64)[::-1].copy()
    pg = np.asarray(col["pg"], dtype=np.float64)[::-1].copy()
    r = -depth
    n_e = pe / (K.K_B_CGS * t)
    n_h = 0.9 * (pg - pe) / (K.K_B_CGS * t)
    return dict(
        temperatures=t,
        r=r,
        dist=(r[1:] - r[:-1]).copy(),
        n_e=n_e,
        n_h=n_h,
        microturbulence=1.0e5,  # 1 km/s in cgs
    )


def tracing_grid(lam0, lam1, R=None, step=None, n_override=None):
    """Wavelengths ascending -> frequencies DESCENDING (stardis/base.py:34)."""
    if step is not None:
        lam = np.arange(lam0, lam1, step)
    else:
        n = int(math.ceil(math.log(lam1 / lam0) * R)) if n_override is None else int(n_override)
        lam = lam0 * np.exp(np.arange(n, dtype=np.float64) * (math.log(lam1 / lam0) / n))
    return K.C_CGS * 1.0e8 / lam


def thetas_and_weights(n_thetas=N_THETAS):
    """Gauss-Legendre nodes mapped as the reference does (radiation_field/base.py:61-63)."""
    x, w = np.polynomial.legendre.leggauss(n_thetas)
    return x / 2.0 + 0.5 * math.pi / 2.0, w * math.pi / 2.0


def synth_lines(nus, atm, n_lines, seed=SEED, gamma_per_depth=True, mix=(0.90, 0.09, 0.01)):
    """Random line list sorted by frequency, layout as calc_alan_entries takes it (N_l, N_d)."""
    rng = np.random.default_rng(seed)
    t = atm["temperatures"]
    n_e = atm["n_e"]
    nu_lo, nu_hi = float(nus.min()), float(nus.max())
    line_nus = np.sort(nu_lo + (nu_hi - nu_lo) * rng.random(n_lines))
    masses = np.array([1.008, 12.011, 24.305, 55.845, 47.867 + 15.999]) * K.AMU_CGS
    mass = masses[rng.integers(0, len(masses), n_lines)]
    doppler = (
        line_nus[:, None]
        / K.C_CGS
        * np.sqrt(2.0 * K.K_B_CGS * t[None, :] / mass[:, None] + atm["microturbulence"] ** 2)
    )
    g_rad = 10.0 ** (7.0 + 2.0 * rng.random(n_lines))
    g_col = 10.0 ** (-9.0 + 2.0 * rng.random(n_lines))
    if gamma_per_depth:
        gammas = g_rad[:, None] + g_col[:, None] * n_e[None, :]
    else:
        gammas = g_rad[:, None].copy()
    e_low = 5.0 * rng.random(n_lines) * K.EV_CGS
    u = rng.random(n_lines)
    v = rng.random(n_lines)
    s = np.where(
        u < mix[0],
        -6.0 + 5.0 * v,
        np.where(u < mix[0] + mix[1], -1.0 + 3.0 * v, 2.0 + 3.0 * v),
    )
    boltz = np.exp(-e_low[:, None] / (K.K_B_CGS * t[None, :]) + e_low[:, None] / (K.K_B_CGS * t.max()))
    alphas = (10.0**s)[:, None] * boltz
    return dict(
        line_nus=np.ascontiguousarray(line_nus),
        doppler_widths=np.ascontiguousarray(doppler),
        gammas=np.ascontiguousarray(gammas),
        alphas=np.ascontiguousarray(alphas),
    )


def synth_linelist(nus, atm, n_lines, seed=SEED, vald_broadening=True, mix=(0.90, 0.09, 0.01)):
    """The same kind of list as synth_lines, as per-line scalars (stardis_amd.linelist.LineList): VALD-style atoms of
    five elements in two ionisation stages, oscillator strengths scaled so that alpha_line at the hottest depth follows
    the strength mix of synth_lines."""
    from . import linelist as LL

    rng = np.random.default_rng(seed)
    t = np.asarray(atm["temperatures"], dtype=np.float64)
    nu_lo, nu_hi = float(nus.min()), float(nus.max())
    nu = np.sort(nu_lo + (nu_hi - nu_lo) * rng.random(n_lines))
    elements = np.array([1, 6, 12, 20, 26])
    amu = np.array([1.008, 12.011, 24.305, 40.078, 55.845])
    abundance = np.array([1.0, 2.7e-4, 4.0e-5, 2.2e-6, 3.2e-5])
    chi_ev = np.array([[13.598, 13.598], [11.26, 24.383], [7.646, 15.035], [6.113, 11.872], [7.902, 16.199]])
    el = rng.choice(len(elements), n_lines, p=[0.04, 0.16, 0.2, 0.2, 0.4])
    stage = np.where(el == 0, 0, rng.integers(0, 2, n_lines))
    chi = chi_ev[el, stage] * K.EV_TO_ERG_ASTROPY
    e_up = (0.25 + 0.7 * rng.random(n_lines)) * chi
    e_lo = np.maximum(e_up - K.H_CGS * nu, 0.0)
    e_up = e_lo + K.H_CGS * nu
    ok = e_up < chi
    e_up = np.where(ok, e_up, 0.98 * chi)
    e_lo = np.where(ok, e_lo, np.maximum(0.98 * chi - K.H_CGS * nu, 0.0))
    x = 1.0 / (1.0 + np.exp(-(t[None, :] - 5200.0 - 90.0 * elements[:, None]) / 700.0))
    n_h = np.asarray(atm["n_h"], dtype=np.float64)
    pop = np.empty((2 * len(elements), t.size))
    for k in range(len(elements)):
        part = 2.0 + 0.35 * (t / 5000.0) ** 1.5
        pop[2 * k] = abundance[k] * n_h * (1.0 if k == 0 else (1.0 - x[k])) / part
        pop[2 * k + 1] = abundance[k] * n_h * x[k] / part
    row = (2 * el + stage).astype(np.int32)
    g_lo = rng.integers(1, 11, n_lines).astype(np.float64)
    u, v = rng.random(n_lines), rng.random(n_lines)
    s_ = np.where(u < mix[0], -6.0 + 5.0 * v, np.where(u < mix[0] + mix[1], -1.0 + 3.0 * v, 2.0 + 3.0 * v))
    d_hot = int(np.argmax(t))
    kt = 1.380649e-23 * t[d_hot]
    hot = K.ALPHA_COEFFICIENT * np.exp(-(e_lo / K.EV_TO_ERG_ASTROPY) * 1.602176634e-19 / kt) * pop[row, d_hot] * g_lo * (
        1.0 - np.exp(-6.62607015e-34 * nu / kt))
    f_lu = 10.0**s_ / hot
    pick = rng.random(n_lines)
    waals = np.where(pick < 0.5, -(7.0 + rng.random(n_lines)), rng.integers(150, 900, n_lines) + 0.2 + 0.15 * rng.random(n_lines))
    return LL.LineList(
        nu, e_lo / K.EV_TO_ERG_ASTROPY, f_lu, row, pop, amu[el] * K.AMU_CGS, t, g_lo=g_lo, microturbulence=float(atm["microturbulence"]),
        gamma_mode=LL.GAMMA_VALD if vald_broadening else LL.GAMMA_CLASSIC, flags=15, atomic_number=elements[el], ion_number=stage + 1,
        ionization_energy=chi, upper_energy=e_up, lower_energy=e_lo, A_ul=10.0 ** (7.0 + 2.0 * rng.random(n_lines)),
        stark=-(4.5 + 2.0 * rng.random(n_lines)), waals=waals, electron_density=atm["n_e"], h_density=n_h,
    )


def synth_continuum_state(atm, n_levels=10):
    """Per-depth densities for H I bf/ff, H- bf (Wishart table), Thomson: an LTE-like
    hydrogen state good enough to give a solar-looking continuum."""
    t = atm["temperatures"]
    n_e = atm["n_e"]
    n_h = atm["n_h"]
    kt = K.K_B_CGS * t
    chi = 13.598434 * K.EV_CGS
    lam3 = (K.H_CGS**2 / (2.0 * math.pi * K.M_E_CGS * kt)) ** 1.5
    saha = (1.0 / lam3) * np.exp(-chi / kt) / n_e  # n(HII)/n(HI), g ratio 2*1/2
    n_h1 = n_h / (1.0 + saha)
    n_h2 = n_h - n_h1
    n_hminus = n_h1 * n_e * lam3 * 0.25 * np.exp(0.754 * K.EV_CGS / kt)
    n = np.arange(1, n_levels + 1, dtype=np.float64)
    exc = chi * (1.0 - 1.0 / n**2)
    g = 2.0 * n**2
    lev = g[:, None] * np.exp(-exc[:, None] / kt[None, :])
    lev = lev / lev.sum(axis=0, keepdims=True) * n_h1[None, :]
    with open(os.path.join(_DATA, "hminus_bf_wishart1979.json")) as fh:
        tab = json.load(fh)
    return dict(
        n_e=n_e,
        n_h1=n_h1,
        n_h2=n_h2,
        n_hminus=n_hminus,
        n_he1=0.085 * n_h,
        level_excitation=exc,
        level_density=np.ascontiguousarray(lev),
        ionization_energy=chi,
        hminus_bf_wavelength=np.asarray(tab["wavelength"], dtype=np.float64),
        hminus_bf_cross_section=np.asarray(tab["cross_section"], dtype=np.float64),
    )


def make_workload(tag, n_lines=None, seed=SEED, n_nu_override=None):
    cfg = dict(WORKLOADS[tag])
    atm = cool_dwarf_atmosphere() if cfg.get("atmosphere") == "cool_dwarf" else solar_atmosphere()
    nus = tracing_grid(cfg["lam0"], cfg["lam1"], cfg.get("R"), cfg.get("step"), n_nu_override)
    lines = synth_lines(
        nus, atm, cfg["n_lines"] if n_lines is None else n_lines, seed, cfg["gamma_per_depth"]
    )
    thetas, weights = thetas_and_weights()
    return dict(
        tag=tag, nus=nus, atm=atm, lines=lines, cont=synth_continuum_state(atm), thetas=thetas, weights=weights
    )


def fake_plasma(nus, atm, n_lines, seed=SEED, vald_broadening=False, table_dir=None, n_molecule_lines=0):
    """A pandas stand-in for the TARDIS plasma, the stellar model and the opacity configuration, shaped as the reference's
    calc_alphas / raytrace read them (opacities_solvers/base.py:630-740, radiation_field_solvers/base.py:271-346): the
    continuum state of synth_continuum_state and a VALD-style line list with a dense alpha table (synth_lines' strengths on
    synth_linelist's atoms: both draw the same line frequencies from the same seed).  Every line lies on the grid and none
    auto-ionises, so the reference's selection keeps them all, in this order.
    -> (plasma, model, config, arrays) with arrays = dict(line_nus, alphas) in the kernel's layout for cross-checks."""
    import tempfile
    import types

    import pandas as pd

    NS = types.SimpleNamespace
    t = np.asarray(atm["temperatures"], dtype=np.float64)
    cols = np.arange(t.size)
    cont = synth_continuum_state(atm)
    spec = synth_linelist(nus, atm, n_lines, seed, vald_broadening=vald_broadening)
    dense = synth_lines(nus, atm, n_lines, seed)
    assert np.array_equal(spec.nu, dense["line_nus"])
    lines = pd.DataFrame(dict(
        atomic_number=np.asarray(spec.atomic_number), ion_number=np.asarray(spec.ion_number) - 1, nu=spec.nu,
        ionization_energy=spec.ionization_energy, level_energy_upper=spec.upper_energy, level_energy_lower=spec.lower_energy,
        A_ul=spec.A_ul, stark=spec.stark, waals=spec.waals, e_low=spec.e_low_ev, f_lu=spec.strength, g_lo=spec.g_lo,
    ))
    alpha_table = pd.DataFrame(dense["alphas"], columns=cols)
    alpha_table["nu"] = dense["line_nus"]
    n_lev = cont["level_density"].shape[0]
    ind = pd.MultiIndex.from_tuples([(1, 0), (1, 1), (2, 0), (2, 1)], names=["atomic_number", "ion_number"])
    lev_index = pd.MultiIndex.from_tuples([(1, 0, k) for k in range(n_lev)], names=["atomic_number", "ion_number", "level_number"])
    ionization_data = pd.Series(
        np.array([cont["ionization_energy"], 24.587 * K.EV_CGS, 54.418 * K.EV_CGS]),
        index=pd.MultiIndex.from_tuples([(1, 1), (2, 1), (2, 2)], names=["atomic_number", "ion_number"]), name="ionization_energy",
    )
    plasma = NS(
        ion_number_density=pd.DataFrame(np.vstack([cont["n_h1"], cont["n_h2"], cont["n_he1"], 1e-6 * cont["n_he1"]]), index=ind, columns=cols),
        electron_densities=pd.Series(cont["n_e"], index=cols),
        levels=lev_index,
        excitation_energy=pd.Series(cont["level_excitation"], index=lev_index),
        level_number_density=pd.DataFrame(cont["level_density"], index=lev_index, columns=cols),
        ionization_data=ionization_data,
        h_minus_density=pd.Series(cont["n_hminus"], index=cols),
        h2_density=pd.Series(np.zeros(t.size), index=cols),
        lines_from_linelist=lines,
        alpha_line_from_linelist=alpha_table,
    )
    # what the per-line-scalar route needs when the dense alpha table is dropped (plasma.alpha_line_from_linelist = None):
    # number density / partition function per (Z, ion) such that their ratio is the list's population row
    species = [(z, k) for z in (1, 6, 12, 20, 26) for k in (0, 1)]
    dens_rows = {key: plasma.ion_number_density.loc[key].to_numpy() for key in plasma.ion_number_density.index}
    for j, key in enumerate(species):
        dens_rows.setdefault(key, spec.pop[j])
    full = pd.MultiIndex.from_tuples(sorted(dens_rows), names=["atomic_number", "ion_number"])
    plasma.ion_number_density = pd.DataFrame(np.vstack([dens_rows[k] for k in full]), index=full, columns=cols)
    part = pd.DataFrame(np.ones((len(full), t.size)), index=full, columns=cols)
    for j, key in enumerate(species):
        part.loc[key] = plasma.ion_number_density.loc[key].to_numpy() / np.where(spec.pop[j] > 0, spec.pop[j], 1.0)
    plasma.partition_function = part
    masses = pd.Series(np.array([1.008, 12.011, 15.999, 24.305, 40.078, 47.867, 55.845]) * K.AMU_CGS,
                       index=pd.Index([1, 6, 8, 12, 20, 22, 26], name="atomic_number"))
    if n_molecule_lines:  # a second, molecular list (include_molecules): TiO / H2O style lines, dense table and per-line scalars
        mol = synth_lines(nus, atm, n_molecule_lines, seed + 1, gamma_per_depth=False, mix=(0.97, 0.03, 0.0))
        mspec = synth_linelist(nus, atm, n_molecule_lines, seed + 1, vald_broadening=False, mix=(0.97, 0.03, 0.0))
        names = np.where(np.arange(n_molecule_lines) % 2 == 0, "TiO", "OH")
        plasma.molecule_lines_from_linelist = pd.DataFrame(dict(nu=mol["line_nus"], molecule=names, A_ul=mspec.A_ul, e_low=mspec.e_low_ev,
                                                                 f_lu=mspec.strength, g_lo=mspec.g_lo))
        mtab = pd.DataFrame(mol["alphas"], columns=cols)
        mtab["nu"] = mol["line_nus"]
        plasma.molecule_alpha_line_from_linelist = mtab
        plasma.molecule_ion_map = pd.DataFrame(dict(Ion1=[22, 8], Ion2=[8, 1]), index=["TiO", "OH"])
        plasma.molecule_number_density = pd.DataFrame(np.vstack([1e-7 * cont["n_h1"], 1e-6 * cont["n_h1"]]), index=["TiO", "OH"], columns=cols)
        plasma.molecule_partition_function = pd.DataFrame(np.vstack([50.0 + 0.01 * t, 20.0 + 0.005 * t]), index=["TiO", "OH"], columns=cols)
    r = np.asarray(atm["r"], dtype=np.float64)
    model = NS(
        temperatures=t, no_of_depth_points=t.size, spherical=False,
        geometry=NS(dist_to_next_depth_point=np.asarray(atm["dist"], dtype=np.float64), r=r, reference_r=None),
        composition=NS(nuclide_masses=masses), microturbulence=float(atm["microturbulence"]),
    )
    table_dir = table_dir or tempfile.mkdtemp(prefix="stardis_amd_tables_")
    table_path = os.path.join(table_dir, "h_minus_bf.dat")
    with open(table_path, "w") as fh:
        fh.write("\n".join(f"{x!r},{y!r}" for x, y in zip(cont["hminus_bf_wavelength"].tolist(), cont["hminus_bf_cross_section"].tolist())) + "\n")
    opacity = NS(
        file={"Hminus_bf": table_path}, bf={"H_I": {}}, ff={"H_I": {}}, rayleigh=[], disable_electron_scattering=False,
        line=NS(disable=False, broadening=["linear_stark", "quadratic_stark", "van_der_waals", "radiation"],
                vald_linelist=NS(use_linelist=True, use_vald_broadening=bool(vald_broadening)), include_molecules=False),
    )
    config = NS(opacity=opacity, no_of_thetas=N_THETAS, result_options=NS(return_radiation_field=False))
    return plasma, model, config, dict(line_nus=dense["line_nus"], alphas=dense["alphas"])
