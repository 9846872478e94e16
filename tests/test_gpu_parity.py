"""Parity of the HIP path (through the C ABI) with the CPU oracle and the reference's golden vectors.

Tolerances: index/window work is bit-exact; fp64 opacities agree to <= 1e-12 relative (operation
re-association inside a Faddeeva region, device libm); fluxes to <= 1e-10 relative, the path's stated
tolerance (BASELINE.json north_star) — the floor is the reference's own ill-conditioned quadrature
weights (see tests/test_oracle_golden.py::test_weights_golden).
"""
import json
import os

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, load_golden, rel_err
from stardis_amd import constants as K
from stardis_amd import ops, synth

pytestmark = pytest.mark.gpu

OPACITY_RTOL = 1e-12
FLUX_RTOL = 1e-10


# ------------------------------------------------------------------------------------------------ voigt
def test_faddeeva_known_answers(ctx):
    assert ops.faddeeva(0j) == 1 + 0j  # reference test_voigt.py:22-37
    for dw in (1.0, 3.3, 1e9):
        assert ops.voigt_profile(0.0, dw, 0.0) == 1 / (np.sqrt(np.pi) * dw)  # :151-178
    with pytest.raises(ZeroDivisionError):
        ops.voigt_profile(1.0, 0.0, 1.0)  # :130-148


def test_reference_unit_test_known_answers(ctx):
    """Every sample value of the reference's own unit tests for this path, through the reference-named functions."""
    import reference_known_answers as ka
    from stardis_amd.radiation_field.opacities.opacities_solvers import broadening as B
    from stardis_amd.radiation_field.opacities.opacities_solvers import voigt as V

    fn = {"doppler_width": B.calc_doppler_width, "n_effective": B.calc_n_effective, "gamma_linear_stark": B.calc_gamma_linear_stark,
          "gamma_quadratic_stark": B.calc_gamma_quadratic_stark, "gamma_van_der_waals": B.calc_gamma_van_der_waals}
    for name, args, expected in ka.BROADENING:
        got = fn[name](*args)
        assert np.allclose(got, expected), (name, got, expected)
        assert np.shape(got) == np.shape(expected)
    for z, expected in ka.FADDEEVA:
        assert np.allclose(V.faddeeva(z), expected)
    for args, expected in ka.VOIGT:
        assert np.allclose(V.voigt_profile(*args), expected)
    for dnu in ka.VOIGT_DIVISION_BY_ZERO:
        for gam in ka.VOIGT_DIVISION_BY_ZERO:
            with pytest.raises(ZeroDivisionError):
                V.voigt_profile(dnu, 0, gam)


def test_faddeeva_golden(ctx):
    g = load_golden("g1_faddeeva")
    w = ops.faddeeva(g["z"])
    ref = g["w"]
    # regions I-III are rational functions evaluated in the reference's operation order: bit-exact;
    # region IV calls exp/sin/cos, where device and host libm may differ in the last place
    x, y = g["z"].real, g["z"].imag
    s = np.abs(x) + y
    rational = (s > 5.5) | (y >= 0.195 * np.abs(x) - 0.176)
    assert np.array_equal(w[rational], ref[rational])
    assert rel_err(w.real, ref.real) < 1e-14
    assert np.max(np.abs(w.imag - ref.imag) / np.maximum(np.abs(ref), 1e-300)) < 1e-14


def test_voigt_profile_golden(ctx):
    g = load_golden("g2_voigt")
    phi = ops.voigt_profile(g["delta_nu"], g["doppler_width"], g["gamma"])
    assert rel_err(phi, g["phi"]) < 1e-14


# ------------------------------------------------------------------------------------------------ line opacity
@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_line_opacity_golden(ctx, case):
    g = load_golden("g4_alan_entries")
    args = (56, g[case + "_nus"], g[case + "_line_nus"], g[case + "_doppler_widths"], g[case + "_gammas"], g[case + "_alphas"])
    out, evals = ops.calc_alan_entries(*args, return_evaluations=True)
    ref = g[case + "_alpha_line_at_nu"]
    assert np.array_equal(out == 0, ref == 0)  # same windows
    assert rel_err(out, ref) < OPACITY_RTOL
    _, evals_oracle = oracle.calc_alan_entries(*args, return_evals=True)
    assert evals == evals_oracle


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_window_rule_bit_exact(ctx, case):
    g = load_golden("g4_alan_entries")
    nus, ln, dw, gm, al = (g[case + k] for k in ("_nus", "_line_nus", "_doppler_widths", "_gammas", "_alphas"))
    lo, hi = ops.line_windows(56, nus, ln, dw, gm, al)
    for l in range(len(ln)):
        for d in (0, 17, 55):
            want = oracle.window(nus, ln[l], gm[l, d if gm.shape[1] > 1 else 0], dw[l, d], al[l, d])
            assert (lo[l, d], hi[l, d]) == want


def test_line_opacity_edge_cases(ctx):
    atm = synth.solar_atmosphere()
    nus = synth.tracing_grid(6560.0, 6570.0, step=0.05)
    # no lines
    out = ops.calc_alan_entries(56, nus, [], np.zeros((0, 56)), np.zeros((0, 56)), np.zeros((0, 56)))
    assert out.shape == (56, nus.size) and not out.any()
    # one line, ragged tile (N_nu not a multiple of the tile), gamma as a column
    ln = synth.synth_lines(nus, atm, 1, seed=3, gamma_per_depth=False)
    out = ops.calc_alan_entries(56, nus, ln["line_nus"], ln["doppler_widths"], ln["gammas"], ln["alphas"])
    ref = oracle.calc_alan_entries(56, nus, ln["line_nus"], ln["doppler_widths"], ln["gammas"], ln["alphas"])
    assert rel_err(out, ref) < OPACITY_RTOL
    # F-ordered inputs, as DataFrame.to_numpy() hands them over (base.py:403-407)
    ln = synth.synth_lines(nus, atm, 30, seed=4)
    out_f = ops.calc_alan_entries(56, nus, ln["line_nus"], np.asfortranarray(ln["doppler_widths"]), np.asfortranarray(ln["gammas"]), np.asfortranarray(ln["alphas"]))
    out_c = ops.calc_alan_entries(56, nus, ln["line_nus"], ln["doppler_widths"], ln["gammas"], ln["alphas"])
    assert np.array_equal(out_f, out_c)
    with pytest.raises(ValueError):
        ops.calc_alan_entries(56, nus[::-1], ln["line_nus"], ln["doppler_widths"], ln["gammas"], ln["alphas"])  # ascending grid
    bad = ln["doppler_widths"].copy()
    bad[3, 7] = 0.0
    with pytest.raises(ValueError):
        ops.calc_alan_entries(56, nus, ln["line_nus"], bad, ln["gammas"], ln["alphas"])


def test_line_opacity_synthetic_vs_oracle(ctx):
    """Seeded S-c1-like case with all three window regimes (floor, medium, whole grid)."""
    atm = synth.solar_atmosphere()
    nus = synth.tracing_grid(6560.0, 6570.0, step=0.01)
    ln = synth.synth_lines(nus, atm, 400, seed=7, mix=(0.8, 0.15, 0.05))
    out = ops.calc_alan_entries(56, nus, ln["line_nus"], ln["doppler_widths"], ln["gammas"], ln["alphas"])
    ref = oracle.calc_alan_entries(56, nus, ln["line_nus"], ln["doppler_widths"], ln["gammas"], ln["alphas"])
    assert rel_err(out, ref) < OPACITY_RTOL


def test_line_opacity_is_deterministic(ctx):
    atm = synth.solar_atmosphere()
    nus = synth.tracing_grid(6560.0, 6570.0, step=0.02)
    ln = synth.synth_lines(nus, atm, 300, seed=9)
    a = ops.calc_alan_entries(56, nus, ln["line_nus"], ln["doppler_widths"], ln["gammas"], ln["alphas"])
    b = ops.calc_alan_entries(56, nus, ln["line_nus"], ln["doppler_widths"], ln["gammas"], ln["alphas"])
    assert np.array_equal(a, b)


# ------------------------------------------------------------------------------------------------ broadening
def test_broadening_golden(ctx):
    g = load_golden("g3_broadening")
    z, ion = g["line_atomic_number"], g["line_ion_number"] + 1
    base = (z, ion, g["line_ionization_energy"], g["line_level_energy_upper"], g["line_level_energy_lower"], g["line_A_ul"],
            g["n_e"], g["temperatures"], g["n_h1"])
    for tag, sw in dict(all=(1, 1, 1, 1), no_lin=(0, 1, 1, 1), rad_only=(0, 0, 0, 1), qs_vdw=(0, 1, 1, 0)).items():
        assert rel_err(ops.calc_gamma(*base, *[bool(s) for s in sw]), g["gamma_" + tag]) < 1e-13, tag
    dw = ops.doppler_widths(g["line_nu"], g["line_mass"], g["temperatures"], float(g["microturbulence"]))
    assert rel_err(dw, g["doppler"]) < 1e-15
    va = base[:6] + (g["line_stark"], g["line_waals"], g["line_mass"]) + base[6:]
    assert rel_err(ops.calc_vald_gamma_arrays(*va, True, True, True, True), g["vald_gamma_all"]) < 1e-13
    assert rel_err(ops.calc_vald_gamma_arrays(*va, False, False, True, True), g["vald_gamma_rad_vdw"]) < 1e-13
    # calculate_molecule_broadening(use_vald_broadening=True) (:771-799) on the reference's own call: a molecular table with the
    # VALD columns (the fixture's atomic table + a `molecule` column), five broadening configurations
    import pandas as pd
    from types import SimpleNamespace as NS
    from stardis_amd.radiation_field.opacities.opacities_solvers.broadening import calculate_molecule_broadening

    molv = pd.DataFrame({k[len("line_"):]: g[k] for k in g.files if k.startswith("line_") and k != "line_mass"})
    molv["molecule"] = np.where(g["molv_molecule_is_MgH"] == 1, "MgH", "CH")
    masses = pd.Series(np.array([1.008, 4.0026, 12.011, 24.305, 40.078, 55.845]) * K.AMU_CGS, index=pd.Index([1, 2, 6, 12, 20, 26], name="atomic_number"))
    assert np.array_equal(masses.loc[g["line_atomic_number"]].values, g["line_mass"])
    cols = np.arange(g["temperatures"].size)
    model = NS(temperatures=g["temperatures"], no_of_depth_points=cols.size, microturbulence=float(g["microturbulence"]),
               composition=NS(nuclide_masses=masses))
    plasma = NS(electron_densities=pd.Series(g["n_e"], index=cols),
                ion_number_density=pd.DataFrame(g["n_h1"][None, :], index=pd.MultiIndex.from_tuples([(1, 0)]), columns=cols),
                molecule_ion_map=pd.DataFrame(dict(Ion1=[6, 12], Ion2=[1, 1]), index=["CH", "MgH"]))
    for tag, cfg in dict(all=["linear_stark", "quadratic_stark", "van_der_waals", "radiation"], lin=["linear_stark"], quad=["quadratic_stark"],
                         vdw_rad=["van_der_waals", "radiation"], none=[]).items():
        gam, dop = calculate_molecule_broadening(molv, model, plasma, cfg, use_vald_broadening=True)
        ref = g["molv_gammas_" + tag]
        assert gam.shape == ref.shape and np.array_equal(gam == 0, ref == 0) and rel_err(gam, ref) < 1e-13, tag
    assert rel_err(dop, g["molv_doppler"]) < 1e-15
    # element-wise ufunc counterparts (reference test_broadening.py known answers)
    from stardis_amd.radiation_field.opacities.opacities_solvers import broadening as B

    assert B.calc_doppler_width(K.C_CGS, 0.5, K.K_B_CGS, 0.0) == 1.0  # test_broadening.py:40-71
    assert B.calc_n_effective(1, K.RYDBERG_ENERGY, 0.0) == 1.0  # :146-176
    assert rel_err(B.calc_n_effective(ion[:, None], g["line_ionization_energy"][:, None], g["line_level_energy_upper"][:, None]), g["n_eff_upper"]) < 1e-15
    assert rel_err(B.calc_gamma_linear_stark(g["n_eff_upper"], g["n_eff_lower"], g["n_e"]), g["linear_stark"]) < 1e-13
    assert rel_err(B.calc_gamma_quadratic_stark(ion[:, None], g["n_eff_upper"], g["n_eff_lower"], g["n_e"], g["temperatures"]), g["quadratic_stark"]) < 1e-13
    assert rel_err(B.calc_gamma_van_der_waals(ion[:, None], g["n_eff_upper"], g["n_eff_lower"], g["temperatures"], g["n_h1"]), g["van_der_waals"]) < 1e-13


# ------------------------------------------------------------------------------------------------ continuum
def test_continuum_golden(ctx):
    g = load_golden("g5_continuum")
    with open(os.path.join(os.path.dirname(GOLDEN), "..", "stardis_amd", "data", "hminus_bf_wishart1979.json")) as fh:
        tab = json.load(fh)
    for tag in ("opt", "wide"):
        nus, lam = g[tag + "_nus"], g[tag + "_lambdas"]
        assert np.array_equal(K.nu_to_angstrom(nus), lam)  # host conversion matches astropy's bit for bit
        assert rel_err(ops.alpha_file_1d(lam, tab["wavelength"], tab["cross_section"], g["n_hminus"]).numpy(), g[tag + "_alpha_file_Hminus_bf"]) < 1e-15
        assert np.array_equal(ops.alpha_file_2d(g[tag + "_sigma_Hminus_ff"], g["n_h1"] * g["n_e"]).numpy(), g[tag + "_alpha_file_Hminus_ff"])
        cutoff = (g["ionization_energy"] - g["level_excitation"]) / K.H_CGS
        assert rel_err(ops.alpha_bf(nus, [0, len(cutoff)], [0], cutoff, g["level_density"], 56).numpy(), g[tag + "_alpha_bf"]) < 1e-14
        assert rel_err(ops.alpha_ff(nus, g["temperatures"], [1], g["n_e"] * g["n_h2"]).numpy(), g[tag + "_alpha_ff"]) < 1e-14
        dev, clipped = ops.alpha_rayleigh(nus.copy(), 56, n_h=g["n_h1"], n_he=g["n_he1"], n_h2=g["h2_density"])
        assert rel_err(dev.numpy(), g[tag + "_alpha_rayleigh"]) < 1e-14
        assert np.array_equal(clipped, g[tag + "_nus_after_rayleigh"])
        assert np.array_equal(ops.alpha_electron(len(nus), g["n_e"]).numpy(), g[tag + "_alpha_electron"])


# ------------------------------------------------------------------------------------------------ formal solution
def test_weights_and_blackbody_golden(ctx):
    g = load_golden("g6_weights")
    w0, w1, w2 = ops.calc_weights_parallel(g["tau"])
    o0, o1, o2 = oracle.calc_weights_parallel(g["tau"])
    tau = g["tau"]
    edge = (tau < 5e-4) | (tau >= 50)
    for mine, ref in ((w0, o0), (w1, o1), (w2, o2)):
        assert rel_err(mine[edge], ref[edge]) < 1e-15
        assert np.max(np.abs(mine - ref)) < 6e-16  # one ulp of exp(-tau), see test_oracle_golden.test_weights_golden
    g7 = load_golden("g7_raytrace")
    assert rel_err(ops.blackbody_flux_at_nu(g7["nus"], g7["temperatures"].reshape(-1, 1)), g7["blackbody"]) < 1e-14


@pytest.mark.parametrize("n_theta", [1, 4, 20])
def test_raytrace_golden(ctx, n_theta):
    g = load_golden("g7_raytrace")
    rd = g["dist"].reshape(-1, 1) / np.cos(g[f"thetas_{n_theta}"])
    F, I = ops.raytrace_arrays(g["nus"], g["temperatures"], rd, g[f"weights_{n_theta}"], g["total_alphas"], track=(n_theta == 4))
    assert rel_err(F, g[f"F_nu_{n_theta}"]) < FLUX_RTOL
    if n_theta == 4:
        assert rel_err(I, g["I_nus_4"]) < FLUX_RTOL
    assert np.all(F[:, 7] == 0)  # transparent column, radiation_field_solvers/base.py:203-206


def test_raytrace_accumulates_and_single_theta(ctx):
    from stardis_amd.radiation_field.radiation_field_solvers.base import single_theta_trace_parallel

    g = load_golden("g7_raytrace")
    ok = np.isfinite(g["F_nu_4"]).all(axis=0)
    nus, a = g["nus"][ok], g["total_alphas"][:, ok]
    rd = g["dist"].reshape(-1, 1) / np.cos(g["thetas_4"])
    F, _ = ops.raytrace_arrays(nus, g["temperatures"], rd, g["weights_4"], a)
    F2, _ = ops.raytrace_arrays(nus, g["temperatures"], rd, g["weights_4"], a, F_nu=F.copy())
    assert rel_err(F2, 2 * F) < 1e-15  # F_nu += ..., base.py:336
    one = single_theta_trace_parallel(g["dist"] / np.cos(0.3), g["temperatures"].reshape(-1, 1), g["total_alphas"], g["nus"])
    assert rel_err(one, g["I_single_theta_0p3"]) < FLUX_RTOL


def test_raytrace_many_angles_chunks(ctx):
    """n_theta above one launch's lane budget is traced in chunks that accumulate into F_nu."""
    g = load_golden("g7_raytrace")
    ok = np.isfinite(g["F_nu_4"]).all(axis=0)
    nus, a = g["nus"][ok][:40], g["total_alphas"][:, ok][:, :40]
    th, w = synth.thetas_and_weights(70)
    rd = g["dist"].reshape(-1, 1) / np.cos(th)
    F, _ = ops.raytrace_arrays(nus, g["temperatures"], rd, w, a)
    ref, _ = oracle.raytrace(nus, g["temperatures"], g["dist"], th, w, a)
    assert rel_err(F, ref) < FLUX_RTOL


@pytest.mark.parametrize("n_depth,n_theta,n_nu", [(2, 3, 5), (3, 1, 70), (9, 7, 33), (30, 20, 200), (57, 20, 100), (57, 64, 9),
                                                  (58, 20, 50), (12, 33, 17), (56, 20, 3000)])
def test_raytrace_fresh_flux_shapes(ctx, n_depth, n_theta, n_nu):
    """A flux that is written, not added to, on grids of every shape: small plane-parallel grids take the segmented formal
    solution (the gaps of a ray over the 8 waves of a workgroup: segments of 1..7 gaps, idle trailing waves, angle counts
    that do not divide 64, one frequency in the last workgroup), the rest the one-wave-per-ray kernel — both against the oracle,
    intensities included (radiation_field_solvers/base.py:85-268)."""
    rng = np.random.default_rng(100 * n_depth + n_theta)
    temps = np.linspace(3900.0, 9500.0, n_depth)
    dist = rng.uniform(2e5, 4e6, n_depth - 1)
    nus = np.linspace(6.5e14, 4.2e14, n_nu)
    alphas = 10.0 ** rng.uniform(-9.5, -4.5, (n_depth, n_nu)) * np.linspace(0.05, 30.0, n_depth).reshape(-1, 1)
    if n_nu > 4:
        alphas[:, 3] = 0.0  # a transparent column (:203-206)
    th, w = synth.thetas_and_weights(n_theta)
    rd = dist.reshape(-1, 1) / np.cos(th)
    F, I = ops.raytrace_arrays(nus, temps, rd, w, alphas, track=True)
    ref, Iref = oracle.raytrace(nus, temps, dist, th, w, alphas, track=True)
    # random columns are far rougher than an atmosphere: where the intensity passes through ~1e-6 of its scale the
    # reference's own formulas lose six digits (both kernels and the oracle differ there by the same 1e-9), so the error is
    # measured against the scale of the ray / column
    assert np.max(np.abs(I - Iref) / np.maximum(np.abs(Iref).max(axis=0, keepdims=True), 1e-300)) < FLUX_RTOL
    assert np.max(np.abs(F - ref) / np.maximum(np.abs(ref).max(axis=0, keepdims=True), 1e-300)) < FLUX_RTOL
    assert np.all(F[0] == 0)


# ------------------------------------------------------------------------------------------------ post-processing
def test_rotation_broadening_golden(ctx):
    """reference rotation_broadening (broadening.py:824-877) incl. scipy's reflect boundary and summation order"""
    from stardis_amd.radiation_field.opacities.opacities_solvers.broadening import rotation_broadening

    g = load_golden("g8_rotation")
    vpp, lam, flux = float(g["velocity_per_pix"]), g["wavelength"], g["flux"]
    w0, f0 = rotation_broadening(vpp, lam, flux, 0.0)
    assert f0 is flux and w0 is lam  # identity below 1e-5 km/s (:866-867)
    for v, ld, key in ((20.0, 0.6, "flux_v20"), (500.0, 0.6, "flux_v500"), (35.0, 0.3, "flux_v35_ld0p3")):
        _, out = rotation_broadening(vpp, lam, flux, v, ld)
        assert np.array_equal(out, g[key]), key
        assert rel_err(out, oracle.rotation_broadening(flux, vpp, v, ld)) < 1e-15
    # non-symmetric kernel takes the general path: compare with scipy directly
    from scipy.ndimage import convolve1d

    from stardis_amd.postprocess import convolve1d_reflect

    k = np.array([0.1, 0.5, 0.2, 0.15, 0.05])
    assert np.array_equal(convolve1d_reflect(flux, k), convolve1d(flux, k))
    short = flux[:3]
    assert np.array_equal(convolve1d_reflect(short, np.ones(9) / 9), convolve1d(short, np.ones(9) / 9))  # kernel longer than the data


# ------------------------------------------------------------------------------------------------ spherical geometry
def test_spherical_raytrace_golden(ctx):
    """raytrace with spherical=True: chord lengths (:349-381), inward sweep with the reference's index wrap at gap 0
    (:141-198), outward pass, photospheric correction (:340-344).  Flux to the path tolerance; individual
    intensities of grazing rays are ill-conditioned in the reference itself (optical depths just above the 5e-4
    switch of :28,:38 amplify one ulp of exp to ~1e-8, see tests/test_oracle_golden.py), hence 1e-7 there."""
    import types

    from stardis_amd.radiation_field.radiation_field_solvers.base import calculate_spherical_ray, raytrace, single_theta_trace_parallel
    from stardis_amd.radiation_field.source_functions.blackbody import blackbody_flux_at_nu

    g = load_golden("g10_spherical")
    assert np.array_equal(calculate_spherical_ray(g["thetas"], g["r"]), g["ray_distances"])
    NS = types.SimpleNamespace
    nd, nn, nt = g["temperatures"].size, g["nus"].size, g["thetas"].size
    model = NS(spherical=True, temperatures=g["temperatures"], no_of_depth_points=nd,
               geometry=NS(r=g["r"], reference_r=float(g["reference_r"]), dist_to_next_depth_point=np.diff(g["r"])))
    field = NS(thetas=g["thetas"], I_nus_weights=g["weights"], frequencies=g["nus"], source_function=blackbody_flux_at_nu,
               track_individual_intensities=True, F_nu=np.zeros((nd, nn)), I_nus=np.zeros((nd, nn, nt)),
               opacities=NS(total_alphas=g["total_alphas"]))
    F = raytrace(model, field)
    assert rel_err(F, g["F_nu"]) < FLUX_RTOL
    assert rel_err(field.I_nus, g["I_nus"]) < 1e-7
    with np.errstate(all="ignore"):
        F_o, I_o = oracle.raytrace(g["nus"], g["temperatures"], None, g["thetas"], g["weights"], g["total_alphas"], track=True,
                                   spherical_r=g["r"], reference_r=float(g["reference_r"]))
    assert rel_err(F, F_o) < FLUX_RTOL
    assert rel_err(field.I_nus, I_o) < 1e-7
    one = single_theta_trace_parallel(g["ray_distances"][:, 5].copy(), g["temperatures"].reshape(-1, 1), g["total_alphas"], g["nus"],
                                      blackbody_flux_at_nu, inward_rays=True)
    assert rel_err(one, g["I_single_inward_theta5"]) < 1e-7


def test_gaussian_line_spread_function(ctx):
    """The instrumental LSF of the reference's rotation-broadening walk-through (docs/rotation_broadening cells 7-19):
    gaussian_filter1d to the target resolution, then rotation_broadening — bit for bit what scipy returns."""
    from scipy.ndimage import convolve1d, gaussian_filter1d

    from stardis_amd import postprocess as pp

    g = load_golden("g8_rotation")
    flux = g["flux"]
    fwhm = 6500.0 / 100000 / 0.01  # cells 7-8
    sigma = fwhm / 2.355
    for s_ in (sigma, 0.4, 27.6):
        mine = pp.gaussian_filter1d(flux, s_)
        assert np.array_equal(mine, gaussian_filter1d(flux, s_))
        assert rel_err(mine, oracle.gaussian_filter1d(flux, s_)) < 1e-15
    lsf = pp.gaussian_filter1d(flux, sigma)
    vel_per_pix = 299792.458 / 100000 / fwhm  # cells 15-17
    _, broad = pp.rotation_broadening(vel_per_pix, g["wavelength"], lsf, v_rot=20.0)
    assert np.array_equal(broad, convolve1d(lsf, pp.rotation_profile(vel_per_pix, 20.0)))
