"""BASELINE.json configs[2]-[4] at their FULL sizes, through size-independent properties (the oracle cannot run 10^9-10^10
Voigt evaluations in a test):

* the pre-pass's evaluation count equals the window rule evaluated on the host (opacities_solvers/base.py:524-575);
* the formal solution is column-independent: a strided subset of columns recomputed by the oracle from the GPU's own
  total opacity must match the GPU flux (radiation_field_solvers/base.py:85-346);
* the line opacity with EVERY line of the list present — candidate lists, culling, tile classification of the full-size
  kernels — against the oracle on >= 200 columns (oracle.calc_alan_entries_columns: the whole grid's window rule, terms only
  at those columns; ~1e7 evaluations), and with it the total opacity and the flux of those columns end to end;
* the line opacity is linear in the line list: a strided subset of the lines, run alone on the full grid, equals the
  oracle's calc_alan_entries for that subset (opacities_solvers/base.py:487-592);
* config 5 = config 3 + fp32-mixed synthesis + instrumental LSF + rotational kernel without leaving the device
  (docs/rotation_broadening cells 7-19, broadening.py:824-877), against the fp64 path at a stated tolerance and against
  scipy's own filters applied to the same spectrum.
"""
import numpy as np
import pytest

import oracle
from conftest import rel_err
from stardis_amd import constants as K
from stardis_amd import parallel, postprocess, synth
from stardis_amd.engine import SpectralSynthesizer

pytestmark = pytest.mark.gpu

MIXED_FLUX_TOL = 1e-4  # SURVEY §8d / BASELINE.md §4: fp32-mixed path, relative on the flux


def full_size_checks(ctx, tag, line_stride, col_stride):
    w = synth.make_workload(tag)
    atm, nus, lines = w["atm"], w["nus"], w["lines"]
    cfg = synth.WORKLOADS[tag]
    assert nus.size == 120398 and lines["line_nus"].size == cfg["n_lines"]
    syn = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], w["thetas"], w["weights"], lines, w["cont"], ctx=ctx)
    syn.step()
    assert syn.evaluations() == parallel.window_evaluations(nus, lines["line_nus"], lines["doppler_widths"], lines["gammas"], lines["alphas"])
    F, total, line = syn.F_nu(), syn.total_alphas(), syn.alpha_line()
    assert np.isfinite(F).all() and (F[-1] > 0).all() and (line >= 0).all() and np.all(F[0] == 0)
    cols = np.arange(0, nus.size, col_stride)
    F_ref, _ = oracle.raytrace(nus[cols], atm["temperatures"], atm["dist"], w["thetas"], w["weights"], np.ascontiguousarray(total[:, cols]))
    assert rel_err(F[1:, cols], F_ref[1:]) < 1e-10
    # total = continuum + line, continuum from the oracle on the same strided columns
    cont = w["cont"]
    cutoff = (cont["ionization_energy"] - cont["level_excitation"]) / K.H_CGS
    c_ref = oracle.alpha_file_1d(K.nu_to_angstrom(nus[cols]), cont["hminus_bf_wavelength"], cont["hminus_bf_cross_section"], cont["n_hminus"])
    c_ref = c_ref + oracle.alpha_bf(nus[cols], [0, len(cutoff)], [0], cutoff, cont["level_density"])
    c_ref = c_ref + oracle.alpha_ff(nus[cols], atm["temperatures"], [1], cont["n_e"] * cont["n_h2"])
    c_ref = c_ref + oracle.alpha_electron(cols.size, cont["n_e"])
    assert rel_err(total[:, cols], c_ref + line[:, cols]) < 1e-15
    # every line of the list, on these columns, against the oracle (global window rule, evaluated at the columns only), and the
    # flux of these columns from the oracle's own total: the whole chain without the GPU in it
    line_ref, ev_cols = oracle.calc_alan_entries_columns(cols, 56, nus, lines["line_nus"], lines["doppler_widths"], lines["gammas"], lines["alphas"],
                                                         return_evals=True)
    assert cols.size >= 200 and ev_cols > 1e6
    assert np.array_equal(line[:, cols] == 0, line_ref == 0)
    assert rel_err(line[:, cols], line_ref) < 1e-12
    F_cpu, _ = oracle.raytrace(nus[cols], atm["temperatures"], atm["dist"], w["thetas"], w["weights"], c_ref + line_ref)
    assert rel_err(F[1:, cols], F_cpu[1:]) < 1e-10
    del syn
    sub = {k: np.ascontiguousarray(v[::line_stride]) for k, v in lines.items()}
    s2 = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], w["thetas"], w["weights"], sub, w["cont"], ctx=ctx)
    s2.step()
    ref, evals = oracle.calc_alan_entries(56, nus, sub["line_nus"], sub["doppler_widths"], sub["gammas"], sub["alphas"], return_evals=True)
    assert s2.evaluations() == evals
    got = s2.alpha_line()
    assert np.array_equal(got == 0, ref == 0)
    assert rel_err(got, ref) < 1e-12
    return line


def test_config3_full_spectrum_150k_lines(ctx):
    """BASELINE configs[2]: solar MARCS, 3000-10000 A at R = 1e5 (120 398 frequencies), 1.5e5 lines: the indexed wide path."""
    full_size_checks(ctx, "S-c3", line_stride=40, col_stride=601)


def test_config4_cool_dwarf_million_lines(ctx):
    """BASELINE configs[3]: the coolest MARCS structure the reference ships, 1e6 molecular-style lines, gamma (N_l, 1)."""
    full_size_checks(ctx, "S-c4m", line_stride=400, col_stride=601)


def test_saturation_grid_ten_times_the_resolving_power(ctx):
    """SURVEY §8d's saturation case, S-big: the grid of configs[2] at R = 1e6 (1 203 973 frequencies) with the 1e6-line list of
    configs[3] — 2.9e11 evaluations, every window ten times as many points wide: the largest index arithmetic the path sees."""
    w = synth.make_workload("S-big")
    atm, nus, lines, cont = w["atm"], w["nus"], w["lines"], w["cont"]
    assert nus.size == 1203973 and lines["line_nus"].size == 1000000
    syn = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], w["thetas"], w["weights"], lines, cont, ctx=ctx)
    syn.step()
    assert syn.evaluations() == parallel.window_evaluations(nus, lines["line_nus"], lines["doppler_widths"], lines["gammas"], lines["alphas"])
    cols = np.arange(300, nus.size, nus.size // 200)
    F, total, line = (np.ascontiguousarray(a[:, cols]) for a in (syn.F_nu(), syn.total_alphas(), syn.alpha_line()))
    del syn
    cutoff = (cont["ionization_energy"] - cont["level_excitation"]) / K.H_CGS
    c_ref = oracle.alpha_file_1d(K.nu_to_angstrom(nus[cols]), cont["hminus_bf_wavelength"], cont["hminus_bf_cross_section"], cont["n_hminus"])
    c_ref = c_ref + oracle.alpha_bf(nus[cols], [0, len(cutoff)], [0], cutoff, cont["level_density"])
    c_ref = c_ref + oracle.alpha_ff(nus[cols], atm["temperatures"], [1], cont["n_e"] * cont["n_h2"])
    c_ref = c_ref + oracle.alpha_electron(cols.size, cont["n_e"])
    line_ref = oracle.calc_alan_entries_columns(cols, 56, nus, lines["line_nus"], lines["doppler_widths"], lines["gammas"], lines["alphas"])
    assert np.array_equal(line == 0, line_ref == 0)
    assert rel_err(line, line_ref) < 1e-12
    assert rel_err(total, c_ref + line_ref) < 1e-12
    F_cpu, _ = oracle.raytrace(nus[cols], atm["temperatures"], atm["dist"], w["thetas"], w["weights"], c_ref + line_ref)
    assert rel_err(F[1:], F_cpu[1:]) < 1e-10


def test_config5_mixed_precision_then_lsf_and_rotation_on_the_device(ctx):
    """BASELINE configs[4].  fp64 and fp32-mixed syntheses of the full spectrum; both spectra go through F_lambda ->
    gaussian_filter1d -> rotation_broadening on the device."""
    import scipy.ndimage as ndi

    tag = "S-c5"
    cfg = synth.WORKLOADS[tag]
    w = synth.make_workload(tag)
    atm, nus = w["atm"], w["nus"]
    lam = K.nu_to_angstrom(nus)
    # the walk-through's numbers (cells 7-17) for this grid: dispersion = lambda / R_grid, FWHM in pixels, sigma, km/s per pixel
    fwhm_pix = cfg["R"] / cfg["lsf_resolution"]
    sigma_pix = fwhm_pix / 2.355
    vel_per_pix = K.C_KMS / cfg["lsf_resolution"] / fwhm_pix
    out = {}
    try:
        for mode in (0, 1):
            ctx.set_option("mixed_precision", mode)
            syn = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], w["thetas"], w["weights"], w["lines"], w["cont"], ctx=ctx)
            syn.step()
            spec = postprocess.DeviceSpectrum(syn)
            f_lambda = spec.spectrum_lambda().numpy()
            broad = spec.broadened(sigma_pix=sigma_pix, velocity_per_pix=vel_per_pix, v_rot=cfg["v_rot_kms"]).numpy()
            out[mode] = dict(F=syn.F_nu()[-1], f_lambda=f_lambda, broad=broad, line=syn.alpha_line())
            if mode == 0:  # the fp64 leg of configs[4] itself against the oracle: all lines on 201 columns
                cols = np.arange(300, nus.size, 601)
                ln = w["lines"]
                line_ref = oracle.calc_alan_entries_columns(cols, 56, nus, ln["line_nus"], ln["doppler_widths"], ln["gammas"], ln["alphas"])
                assert cols.size >= 200 and rel_err(out[0]["line"][:, cols], line_ref) < 1e-12
            # the device chain is the host chain of the walk-through, operation for operation
            F = out[mode]["F"]
            assert np.array_equal(f_lambda, F * nus / lam)
            host = ndi.gaussian_filter1d(f_lambda, sigma_pix)
            assert rel_err(spec.broadened(sigma_pix=sigma_pix).numpy(), host) < 1e-15
            host = ndi.convolve1d(host, postprocess.rotation_profile(vel_per_pix, cfg["v_rot_kms"]))
            assert rel_err(broad, host) < 1e-15
            assert rel_err(broad, oracle.rotation_broadening(oracle.gaussian_filter1d(f_lambda, sigma_pix), vel_per_pix, cfg["v_rot_kms"])) < 1e-15
            del syn, spec
    finally:
        ctx.set_option("mixed_precision", 0)
    # the tolerance path against the fp64 path (which test_config3 pins to the oracle): stated tolerance 1e-4 on the flux
    assert rel_err(out[1]["F"], out[0]["F"]) < MIXED_FLUX_TOL
    assert rel_err(out[1]["broad"], out[0]["broad"]) < MIXED_FLUX_TOL
    assert rel_err(out[1]["line"], out[0]["line"]) < 1e-4
    # broadening conserves the spectrum's integral away from the reflecting ends and only smooths it
    core = slice(2000, -2000)
    assert abs(out[0]["broad"][core].sum() / out[0]["f_lambda"][core].sum() - 1.0) < 1e-6
    assert np.abs(np.diff(out[0]["broad"])).sum() < np.abs(np.diff(out[0]["f_lambda"])).sum()
