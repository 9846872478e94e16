"""The CPU oracle (oracle/stardis_oracle.c) against golden vectors produced by the reference itself.

This pins the oracle: every function is compared with what the reference's own source returned for the
same inputs (tests/golden/make_golden.py).  Tolerances are stated per check; where they are looser than
1e-13 the reason is the reference's own conditioning, noted inline.
"""
import json
import os

import numpy as np
import pytest

import oracle
from conftest import GOLDEN, load_golden, rel_err, species_arrays_from_g14
from stardis_amd import constants as K


def test_constants_match_reference_bit_for_bit():
    with open(os.path.join(GOLDEN, "constants.json")) as fh:
        ref = {k: float.fromhex(v) for k, v in json.load(fh).items()}
    for k, v in ref.items():
        assert getattr(K, k) == v, k


def test_faddeeva_known_answers():
    # reference test_voigt.py:22-37, :151-178
    assert oracle.faddeeva(np.array([0j]))[0] == 1 + 0j
    assert oracle.voigt_profile(0.0, 1.0, 0.0) == pytest.approx(1 / np.sqrt(np.pi), rel=0, abs=0)
    for dw in (1.0, 3.3, 1e9):
        assert oracle.voigt_profile(0.0, dw, 0.0) == 1 / (np.sqrt(np.pi) * dw)
    with pytest.raises(ZeroDivisionError):
        oracle.voigt_profile(1.0, 0.0, 1.0)


def test_faddeeva_golden_bit_exact():
    g = load_golden("g1_faddeeva")
    w = oracle.faddeeva(g["z"])
    assert np.array_equal(w.real, g["w"].real)
    assert np.array_equal(w.imag, g["w"].imag)


def test_voigt_golden_bit_exact():
    g = load_golden("g2_voigt")
    assert np.array_equal(oracle.voigt_profile(g["delta_nu"], g["doppler_width"], g["gamma"]), g["phi"])


def test_broadening_golden():
    g = load_golden("g3_broadening")
    z, ion = g["line_atomic_number"], g["line_ion_number"] + 1
    args = (z, ion, g["line_ionization_energy"], g["line_level_energy_upper"], g["line_level_energy_lower"], g["line_A_ul"],
            g["n_e"], g["temperatures"], g["n_h1"])
    for tag, flags in dict(all=15, no_lin=14, rad_only=8, qs_vdw=6).items():
        assert np.array_equal(oracle.calc_gamma(*args, flags=flags), g["gamma_" + tag]), tag
    dw = oracle.doppler_widths(g["line_nu"], g["line_mass"], g["temperatures"], float(g["microturbulence"]))
    assert np.array_equal(dw, g["doppler"])
    va = (z, ion, g["line_ionization_energy"], g["line_level_energy_upper"], g["line_level_energy_lower"], g["line_A_ul"],
          g["line_stark"], g["line_waals"], g["line_mass"], g["n_e"], g["temperatures"], g["n_h1"])
    # numpy evaluates 10**x and x**0.38 with its own SIMD pow: 1e-15 level differences from libm
    assert rel_err(oracle.calc_vald_gamma(*va, flags=15), g["vald_gamma_all"]) < 5e-15
    assert rel_err(oracle.calc_vald_gamma(*va, flags=12), g["vald_gamma_rad_vdw"]) < 5e-15
    # calculate_molecule_broadening(use_vald_broadening=True) (:771-799): Stark when linear OR quadratic is configured, not halved (16)
    for tag, flags in dict(all=8 | 2 | 4, lin=2, quad=2, vdw_rad=4 | 8, none=0).items():
        ref = g["molv_gammas_" + tag]
        got = oracle.calc_vald_gamma(*va, flags=flags | 16)
        assert np.array_equal(got == 0, ref == 0) and rel_err(got, ref) < 5e-15, tag
    assert rel_err(2 * oracle.calc_vald_gamma(*va, flags=14), g["molv_gammas_all"]) < 5e-15  # the same sum through the halving path
    assert np.array_equal(oracle.doppler_widths(g["mol_nu"], g["mol_mass"], g["temperatures"], 1e5), g["mol_doppler"])


@pytest.mark.parametrize("case", ["a", "b", "c"])
def test_calc_alan_entries_golden(case):
    g = load_golden("g4_alan_entries")
    out, evals = oracle.calc_alan_entries(
        56, g[case + "_nus"], g[case + "_line_nus"], g[case + "_doppler_widths"], g[case + "_gammas"], g[case + "_alphas"],
        return_evals=True,
    )
    ref = g[case + "_alpha_line_at_nu"]
    assert np.array_equal(out == 0, ref == 0)  # identical windows
    assert rel_err(out, ref) < 1e-14  # per-thread slabs change the summation order (base.py:589-590)
    assert evals > 0


def test_calc_alan_entries_empty_and_shapes():
    nus = np.linspace(5e14, 4e14, 50)
    assert np.array_equal(oracle.calc_alan_entries(3, nus, [], np.zeros((0, 3)), np.zeros((0, 3)), np.zeros((0, 3))), np.zeros((3, 50)))


def test_window_rule_edges():
    nus = np.linspace(5.0e14, 4.9e14, 101)
    # a line exactly on the top grid frequency: closest index 1 (first grid value strictly below it)
    lo, hi = oracle.window(nus, nus[0], 1e8, 1e9, 1e-9)
    assert (lo, hi) == (0, 11)
    lo, hi = oracle.window(nus, nus[-1], 1e8, 1e9, 1e-9)
    assert (lo, hi) == (91, 101)
    lo, hi = oracle.window(nus, nus[50], 1e8, 1e9, 1e30)
    assert (lo, hi) == (0, 101)


def test_continuum_golden():
    g = load_golden("g5_continuum")
    with open(os.path.join(os.path.dirname(GOLDEN), "..", "stardis_amd", "data", "hminus_bf_wishart1979.json")) as fh:
        tab = json.load(fh)
    for tag in ("opt", "wide"):
        nus, lam = g[tag + "_nus"], g[tag + "_lambdas"]
        assert np.array_equal(oracle.alpha_file_1d(lam, tab["wavelength"], tab["cross_section"], g["n_hminus"]), g[tag + "_alpha_file_Hminus_bf"])
        assert np.array_equal(oracle.alpha_file_2d(g[tag + "_sigma_Hminus_ff"], g["n_h1"] * g["n_e"]), g[tag + "_alpha_file_Hminus_ff"])
        assert np.array_equal(oracle.alpha_file_2d(g[tag + "_sigma_H2plus_bf"], g["h2_plus_density"]), g[tag + "_alpha_file_H2plus_bf"])
        cutoff = (g["ionization_energy"] - g["level_excitation"]) / K.H_CGS
        # nu**-3, x**5: numpy's pow vs libm, 1 ulp
        assert rel_err(oracle.alpha_bf(nus, [0, len(cutoff)], [0], cutoff, g["level_density"]), g[tag + "_alpha_bf"]) < 1e-15
        assert rel_err(oracle.alpha_ff(nus, g["temperatures"], [1], g["n_e"] * g["n_h2"]), g[tag + "_alpha_ff"]) < 1e-15
        n2 = nus.copy()
        assert rel_err(oracle.alpha_rayleigh(n2, g["n_h1"], g["n_he1"], g["h2_density"]), g[tag + "_alpha_rayleigh"]) < 5e-15
        assert np.array_equal(n2, g[tag + "_nus_after_rayleigh"])  # in-place clipping (base.py:99)
        assert rel_err(oracle.alpha_rayleigh(nus.copy(), g["n_h1"]), g[tag + "_alpha_rayleigh_H_only"]) < 5e-15
        assert np.array_equal(oracle.alpha_electron(len(nus), g["n_e"]), g[tag + "_alpha_electron"])


def test_continuum_species_golden():
    """bf / ff with several species and Z > 1 (reference :202-271, :301-315): G14."""
    g = load_golden("g14_continuum_species")
    assert K.H_CGS == 6.62607015e-27
    for case, species in zip(g["case_names"], g["case_species"]):
        off, bf_ion, cut, ld, ff_ion, ff_n = species_arrays_from_g14(g, str(species).split(","))
        for tag in ("wide", "uv"):
            nus = g[tag + "_nus"]
            ref_bf, ref_ff = g[f"{tag}_alpha_bf_{case}"], g[f"{tag}_alpha_ff_{case}"]
            assert ref_bf.max() > 0 and ref_ff.max() > 0
            # nu**-3, x**5: numpy's pow vs libm, 1 ulp each
            assert rel_err(oracle.alpha_bf(nus, off, bf_ion, cut, ld), ref_bf) < 1e-15, (case, tag)
            assert rel_err(oracle.alpha_ff(nus, g["temperatures"], ff_ion, ff_n), ref_ff) < 1e-15, (case, tag)


def test_weights_golden():
    g = load_golden("g6_weights")
    w0, w1, w2 = oracle.calc_weights_parallel(g["tau"])
    tau = g["tau"]
    small, big = tau < 5e-4, tau >= 50
    for mine, ref in ((w0, g["w0"]), (w1, g["w1"]), (w2, g["w2"])):
        assert rel_err(mine[small], ref[small]) < 1e-15
        assert np.array_equal(mine[big], ref[big])
    # 5e-4 <= tau < 50 (:38-45): w1, w2 are differences of nearly equal numbers; one ulp of exp(-tau)
    # (numpy's SIMD exp vs libm) is amplified by 1/tau^2, 1/tau^3 — the reference's own conditioning.
    mid = ~small & ~big
    assert np.max(np.abs(w0[mid] - g["w0"][mid])) < 3e-16
    assert np.max(np.abs(w1[mid] - g["w1"][mid])) < 3e-16
    assert np.max(np.abs(w2[mid] - g["w2"][mid])) < 6e-16


def test_raytrace_golden():
    g = load_golden("g7_raytrace")
    assert rel_err(oracle.blackbody_flux_at_nu(g["nus"], g["temperatures"]), g["blackbody"]) < 2e-15
    with np.errstate(all="ignore"):
        for n in (1, 4, 20):
            F, I = oracle.raytrace(g["nus"], g["temperatures"], g["dist"], g[f"thetas_{n}"], g[f"weights_{n}"], g["total_alphas"], track=(n == 4))
            # 1e-10 is the path's stated tolerance; observed ~6e-12, set by the ill-conditioned weights above
            assert rel_err(F, g[f"F_nu_{n}"]) < 5e-11
            if n == 4:
                assert rel_err(I, g["I_nus_4"]) < 5e-11
        one = oracle.single_theta_trace_parallel(g["dist"] / np.cos(0.3), g["temperatures"], g["total_alphas"], g["nus"])
    assert rel_err(one, g["I_single_theta_0p3"]) < 5e-11
    # transparent column: intensity stays 0 (:203-206)
    assert np.all(F[:, 7] == 0)


def test_raytrace_accumulates_in_place():
    g = load_golden("g7_raytrace")
    ok = np.isfinite(g["F_nu_4"]).all(axis=0)
    a = g["total_alphas"][:, ok]
    F, _ = oracle.raytrace(g["nus"][ok], g["temperatures"], g["dist"], g["thetas_4"], g["weights_4"], a)
    F2, _ = oracle.raytrace(g["nus"][ok], g["temperatures"], g["dist"], g["thetas_4"], g["weights_4"], a, F_nu=F.copy())
    assert rel_err(F2, 2 * F) < 1e-15  # radiation_field_solvers/base.py:336


def test_rotation_broadening_golden():
    g = load_golden("g8_rotation")
    vpp = float(g["velocity_per_pix"])
    assert np.array_equal(oracle.rotation_broadening(g["flux"], vpp, 0.0), g["flux_v0"])
    # scipy's symmetric-kernel summation order restated: bit-exact (the kernel itself may differ in the last place
    # where libm pow and numpy's differ, hence the 1e-15 allowance)
    assert rel_err(oracle.rotation_broadening(g["flux"], vpp, 20.0), g["flux_v20"]) < 1e-15
    assert rel_err(oracle.rotation_broadening(g["flux"], vpp, 500.0), g["flux_v500"]) < 1e-15
    assert rel_err(oracle.rotation_broadening(g["flux"], vpp, 35.0, 0.3), g["flux_v35_ld0p3"]) < 1e-15


def test_spherical_raytrace_golden():
    """Spherical branch incl. the inward sweep's index wrap at gap 0.  Fluxes to the path tolerance; single-ray
    intensities only to 1e-7: grazing rays have optical depths just above the 5e-4 switch where the reference's
    w1, w2 lose up to six digits to one ulp of exp (numpy's vs libm's)."""
    g = load_golden("g10_spherical")
    assert np.array_equal(oracle.calculate_spherical_ray(g["thetas"], g["r"]), g["ray_distances"])
    with np.errstate(all="ignore"):
        F, I = oracle.raytrace(g["nus"], g["temperatures"], None, g["thetas"], g["weights"], g["total_alphas"], track=True,
                               spherical_r=g["r"], reference_r=float(g["reference_r"]))
        one = oracle.single_theta_trace_parallel(g["ray_distances"][:, 5].copy(), g["temperatures"], g["total_alphas"], g["nus"], inward_rays=True)
    assert rel_err(F, g["F_nu"]) < 5e-11
    assert rel_err(I, g["I_nus"]) < 1e-7
    assert rel_err(one, g["I_single_inward_theta5"]) < 1e-7


def test_reference_unit_test_known_answers():
    """Every sample value of the reference's own unit tests for this path (tests/reference_known_answers.py)."""
    import reference_known_answers as ka

    L = oracle.lib()
    scalar = {
        "doppler_width": lambda nu, t, m, xi: L.orc_doppler_width(nu, t, m, xi),
        "n_effective": lambda ion, e_ion, e_lev: L.orc_n_effective(int(ion), e_ion, e_lev),
        "gamma_linear_stark": lambda nu_, nl_, ne: L.orc_gamma_linear_stark(nu_, nl_, ne),
        "gamma_quadratic_stark": lambda ion, nu_, nl_, ne, t: L.orc_gamma_quadratic_stark(int(ion), nu_, nl_, ne, t),
        "gamma_van_der_waals": lambda ion, nu_, nl_, t, nh: L.orc_gamma_van_der_waals(int(ion), nu_, nl_, t, nh),
    }
    for name, args, expected in ka.BROADENING:
        got = np.vectorize(scalar[name], otypes=[float])(*args)
        assert np.allclose(got, expected), (name, got, expected)
    for z, expected in ka.FADDEEVA:
        assert np.allclose(oracle.faddeeva(np.atleast_1d(np.asarray(z, dtype=complex))), np.atleast_1d(expected))

    for args, expected in ka.VOIGT:
        got = oracle.voigt_profile(*args)
        assert np.allclose(got, expected)
        assert np.array_equal(got, np.broadcast_to(expected, got.shape))  # in fact exact
    for dnu in ka.VOIGT_DIVISION_BY_ZERO:
        for gam in ka.VOIGT_DIVISION_BY_ZERO:
            for a, b in np.broadcast(dnu, gam):
                with pytest.raises(ZeroDivisionError):
                    oracle.voigt_profile(float(a), 0.0, float(b))


def test_gaussian_line_spread_function_matches_scipy():
    """docs/rotation_broadening (cell 11): scipy.ndimage.gaussian_filter1d(spectrum, sigma).  The reference calls scipy
    itself; the installed scipy is the cross-check (libm exp against numpy's exp: <= 1 ulp on the weights)."""
    from scipy.ndimage import gaussian_filter1d

    g = load_golden("g8_rotation")
    for sigma in (0.4, 2.76, 27.6):
        assert rel_err(oracle.gaussian_filter1d(g["flux"], sigma), gaussian_filter1d(g["flux"], sigma)) < 1e-15


def test_column_subset_oracle_equals_the_full_one_on_its_columns():
    """oracle.calc_alan_entries_columns (the window rule of the whole grid, terms only at the listed columns) against
    oracle.calc_alan_entries — itself pinned to the reference by G4 above — incl. edge lines, empty and single-column lists."""
    g = load_golden("g4_alan_entries")
    keys = sorted({k.split("_")[0] for k in g.files})
    assert keys
    for case in keys:
        nus, ln = g[case + "_nus"], g[case + "_line_nus"]
        dw, gam, al = g[case + "_doppler_widths"], g[case + "_gammas"], g[case + "_alphas"]
        nd = dw.shape[1]
        full, ev_full = oracle.calc_alan_entries(nd, nus, ln, dw, gam, al, return_evals=True)
        assert rel_err(full, g[case + "_alpha_line_at_nu"]) < 1e-14  # the reference's own output
        for cols in (np.arange(0, nus.size, 7), np.array([0, nus.size - 1]), np.array([nus.size // 2]), np.arange(nus.size), np.zeros(0, dtype=np.int64)):
            sub, ev = oracle.calc_alan_entries_columns(cols, nd, nus, ln, dw, gam, al, return_evals=True)
            assert sub.shape == (nd, cols.size)
            # the same terms in the same per-thread order up to the slab reduction: a few ulp
            assert rel_err(sub, full[:, cols]) < 1e-14, case
            if cols.size == nus.size:
                assert ev == ev_full
    with pytest.raises(ValueError):
        oracle.calc_alan_entries_columns([3, 2], nd, nus, ln, dw, gam, al)
