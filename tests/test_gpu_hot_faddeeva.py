"""The Faddeeva / Voigt routine the LINE KERNELS run (sdx_math.h: region1_re, faddeeva_re_core, voigt_term — real part
only, FMA arithmetic, one refined hardware reciprocal per point), pinned point by point against the reference's own
vectors G1 (voigt.py:17-86, all four Humlicek regions and points hugging every region boundary) and G2 (voigt.py:113-150).
The element-wise entry points tested in test_gpu_parity.py run a different, reference-order routine (faddeeva_full)."""
import numpy as np
import pytest

import oracle
from conftest import load_golden, rel_err
from stardis_amd import ops

pytestmark = pytest.mark.gpu

SQRT_PI = 1.7724538509055159
# re-associated real-part formulas + FMA + the 1-ulp reciprocal: a few ulp of the real part (measured: see the asserts)
TOL = 2e-13


def regions(x, y):
    s = np.abs(x) + y
    r3 = y >= 0.195 * np.abs(x) - 0.176
    return {"I": s > 15, "II": (s <= 15) & (s > 5.5), "III": (s <= 5.5) & r3, "IV": (s <= 5.5) & ~r3}


def test_hot_path_faddeeva_vs_reference_vectors_by_region(ctx):
    """Doppler width 1 makes x = delta_nu * (1 / dw) exact, so every G1 point — the boundary-hugging ones included —
    lands in the reference's region; phi * sqrt(pi) = Re w(z)."""
    g = load_golden("g1_faddeeva")
    x, y, ref = g["z"].real.copy(), g["z"].imag.copy(), g["w"].real
    # y = (gamma / (sqrt(pi) pi)) / 1: choose gamma so that the pre-pass arithmetic returns the golden y bit for bit
    gamma = y * (np.float64(SQRT_PI) * np.float64(np.pi))
    ok = (gamma / (np.float64(SQRT_PI) * np.float64(np.pi))) == y
    assert ok.mean() > 0.5
    x, y, ref, gamma = x[ok], y[ok], ref[ok], gamma[ok]
    got = ops.voigt_term(x, 1.0, gamma, alpha=SQRT_PI)  # amp = sqrt(pi) / (sqrt(pi) * 1) = 1  ->  Re w
    for name, m in regions(x, y).items():
        assert m.sum() > 100, name
        assert rel_err(got[m], ref[m]) < TOL, name


def test_hot_path_voigt_vs_reference_vectors(ctx):
    g = load_golden("g2_voigt")
    got = ops.voigt_term(g["delta_nu"], g["doppler_width"], g["gamma"], alpha=1.0)
    x = g["delta_nu"] / g["doppler_width"]
    y = (g["gamma"] / (np.float64(SQRT_PI) * np.float64(np.pi))) / g["doppler_width"]
    # x = delta_nu * (1 / dw) can differ from the reference's delta_nu / dw by an ulp: a point within an ulp of a region
    # boundary may be evaluated with the neighbouring region's rational (a ~1e-5 step of the Humlicek approximation
    # itself).  Leave such points out of the tight comparison and bound them separately.
    s = np.abs(x) + y
    near = (np.abs(s - 15.0) < 1e-12) | (np.abs(s - 5.5) < 1e-12) | (np.abs(y - (0.195 * np.abs(x) - 0.176)) < 1e-12)
    assert rel_err(got[~near], g["phi"][~near]) < TOL
    if near.any():
        assert rel_err(got[near], g["phi"][near]) < 1e-4


def test_fp32_routine_of_the_mixed_mode_by_region(ctx):
    """voigt_add32 (packed fp32, hardware exp / cos: the narrow role of mixed_precision=1) against the reference's own G1
    vectors, per region.  The W4 approximation is good to 1e-4; the mode's stated tolerance is 1e-4 on the flux; the fp32
    evaluation itself stays below 2e-5 of Re w (measured ~7e-6 in region IV, ~2e-6 elsewhere)."""
    g = load_golden("g1_faddeeva")
    x, y, ref = g["z"].real.copy(), g["z"].imag.copy(), g["w"].real
    # keep away from the region boundaries by more than an fp32 ulp of x, y: a point may take the neighbouring rational
    s = np.abs(x) + y
    far = (np.abs(s - 15.0) > 1e-4) & (np.abs(s - 5.5) > 1e-4) & (np.abs(y - (0.195 * np.abs(x) - 0.176)) > 1e-4) & (y > 1e-30)
    x, y, ref = x[far], y[far], ref[far]
    gamma = y * (np.float64(SQRT_PI) * np.float64(np.pi))
    got = ops.voigt_term(x, 1.0, gamma, alpha=SQRT_PI, fp32=True)
    for name, m in regions(x, y).items():
        assert m.sum() > 100, name
        assert rel_err(got[m], ref[m]) < 2e-5, name


@pytest.mark.parametrize("role", ["wide", "narrow"])
def test_line_kernel_evaluates_every_region_like_the_reference(ctx, role):
    """One line per test column of (Doppler width, gamma), windows forced over a grid whose offsets sweep |x| from 0
    to 40 Doppler widths: out[d, i] must be alpha * voigt_profile(nu_i - nu_l) — here through the kernels themselves
    (wide role: whole-grid windows, tile-level fast path and general path; narrow role: 10-pixel windows on a grid so
    coarse that 10 pixels reach region I)."""
    rng = np.random.default_rng(7)
    n_depth = 3
    nu_l = 4.5e14
    dw = np.array([[1.1e9, 1.6e9, 2.3e9]])
    gam = np.array([[2.0e7, 3.0e8, 6.0e9]])  # y from 1e-3 to 0.6: regions IV and III at the core
    if role == "wide":
        n = 4001
        off = np.linspace(40.0, -40.0, n) * dw[0, 0] + rng.uniform(-1e5, 1e5, n)
        alpha = np.full((1, n_depth), 1e6)  # half-width saturates at N_nu: every point of the grid
    else:
        n = 257
        off = np.linspace(30.0, -30.0, n) * dw[0, 0]  # 0.23 Doppler widths per pixel: +-10 px = |x| <= 2.4 ... use several lines
        alpha = np.full((1, n_depth), 1e-12)  # the 10-pixel floor (base.py:565-567)
    nus = np.sort(nu_l + off)[::-1].copy()
    if role == "wide":
        lines_nu = np.array([nu_l])
        dws, gams, alphas = dw, gam, alpha
    else:
        # narrow windows only reach +-10 px: put 25 lines across the grid, with Doppler widths from 0.02 to 3 px so
        # that 10 px span anything from |x| <= 3 (regions III/IV) to |x| <= 500 (regions I/II)
        k = 25
        lines_nu = np.sort(nu_l + np.linspace(-25.0, 25.0, k) * dw[0, 0] + rng.uniform(-1e7, 1e7, k))
        px = abs(off[1] - off[0])
        dws = np.geomspace(0.02, 3.0, k)[:, None] * px * np.array([[1.0, 1.3, 1.7]])
        gams = np.geomspace(1e-3, 2.0, k)[::-1][:, None] * dws * (SQRT_PI * np.pi) * np.array([[1.0, 0.5, 2.0]])
        alphas = np.full((k, n_depth), 1e-12)
    got = ops.calc_alan_entries(n_depth, nus, lines_nu, dws, gams, alphas)
    ref = oracle.calc_alan_entries(n_depth, nus, lines_nu, dws, gams, alphas)
    assert np.array_equal(got == 0, ref == 0)
    assert rel_err(got, ref) < TOL
    # every region was exercised
    x = (nus[None, :] - lines_nu[:, None]) / dws[:, :1]
    y = (gams[:, :1] / (SQRT_PI * np.pi)) / dws[:, :1]
    inside = ref[0][None, :] != 0 if role == "wide" else np.abs(np.arange(nus.size)[None, :] - (nus.size - np.searchsorted(nus[::-1], lines_nu))[:, None]) <= 10
    hit = regions(x, np.broadcast_to(y, x.shape))
    for name, m in hit.items():
        assert (m & inside).any(), name


def test_recurrence_form_of_regions_iii_and_iv_against_extended_precision(ctx):
    """The hot routine evaluates the polynomials of regions III and IV (voigt.py:60-64, :70-84) by real two-term recurrences
    instead of the reference's complex Horner steps.  Same polynomials, different rounding error: against the formulas
    evaluated in extended precision on the host, region III stays within a few ulp, region IV — argument close to the real
    axis, where the synthetic division loses a digit — within 1.3e-13 (the complex Horner form in fp64: 1.4e-14).  This test
    holds the routine to those bounds on a dense sample, so that a change of the evaluation order shows up here first."""
    L = np.longdouble
    if np.finfo(L).eps > 1e-18:
        pytest.skip("no extended-precision long double on this host")
    rng = np.random.default_rng(20250926)
    n = 400000
    x = rng.uniform(-5.5, 5.5, n)
    y = 10.0 ** rng.uniform(-9.0, 0.74, n)
    s = np.abs(x) + y
    edge = 0.195 * np.abs(x) - 0.176
    keep = (s < 5.5 - 1e-9) & (np.abs(y - edge) > 1e-9)  # clear of the region boundaries (x * (1 / 1) is exact here)
    x, y = x[keep], y[keep]
    three = y >= 0.195 * np.abs(x) - 0.176
    t = y.astype(L) + (-1j) * x.astype(L)
    u = t * t
    p3 = L(16.4955) + t * (L(20.20933) + t * (L(11.96482) + t * (L(3.778987) + t * L(0.5642236))))
    q3 = L(16.4955) + t * (L(38.82363) + t * (L(39.27121) + t * (L(21.69274) + t * (L(6.699398) + t))))
    p4 = L(36183.31) - u * (L(3321.99) - u * (L(1540.787) - u * (L(219.031) - u * (L(35.7668) - u * (L(1.320522) - u * L(0.56419))))))
    q4 = L(32066.6) - u * (L(24322.8) - u * (L(9022.23) - u * (L(2186.18) - u * (L(364.219) - u * (L(61.5704) - u * (L(1.84144) - u))))))
    exact = np.where(three, (p3 / q3).real, (np.exp(u) - t * p4 / q4).real).astype(np.float64)
    gamma = y * (np.float64(SQRT_PI) * np.float64(np.pi))
    same_y = (gamma / (np.float64(SQRT_PI) * np.float64(np.pi))) == y  # the pre-pass arithmetic must return this y bit for bit
    got = ops.voigt_term(x[same_y], 1.0, gamma[same_y], alpha=SQRT_PI)
    exact, three = exact[same_y], three[same_y]
    err = np.abs(got - exact) / np.abs(exact)
    assert three.sum() > 20000 and (~three).sum() > 100000
    assert err[three].max() < 2e-14
    assert err[~three].max() < TOL
    assert np.quantile(err[~three], 0.5) < 2e-14
