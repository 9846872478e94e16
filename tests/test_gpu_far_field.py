"""The far field of the line kernels (k_line_far, context option "far_field"): (line, depth, tile) triples whose 256-point tile lies
wholly inside the line's window, clear of its core and at least three tile widths from its centre are summed at the tile's 16
Chebyshev nodes and carried to the grid points by the degree-15 interpolant.  It must agree with the direct sum far inside the
opacity tolerance (1e-12), keep the zeros of the reference, stay bit-identical under frequency sharding and work on grids whose
spacing jumps, whose length is no multiple of a tile, and with every kind of line list."""
import numpy as np
import pytest

import oracle
from conftest import rel_err
from stardis_amd import _lib, synth
from stardis_amd.engine import SpectralSynthesizer, shard_bounds

pytestmark = pytest.mark.gpu

FAR_VS_DIRECT = 2e-13  # measured <= 7e-15 on the 1e6-line list; the interpolation bound is 6e-18 of an item, the rest is rounding


@pytest.fixture
def far_ctx():
    ctx = _lib.Context(0)
    yield ctx
    ctx.close()


def run(ctx, far, nus, atm, lines, cont, th, w, **kw):
    ctx.set_option("far_field", far)
    ctx.call("sdx_profile_enable", 1)
    ctx.call("sdx_profile_reset")
    syn = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, lines, cont, ctx=ctx, track_evaluations=False, **kw)
    syn.step()
    out = syn.alpha_line().copy(), syn.total_alphas().copy(), syn.F_nu().copy()
    # (the far field is a role of the line kernel's launch — or, under an experiment knob, a launch of its own)
    launched = ctx.profile("k_line_far")[0] > 0 or "far" in ctx.profile_variant("k_line_all")
    ctx.call("sdx_profile_enable", 0)
    syn.close()
    return out, launched


def workload(lam0, lam1, R, n_lines, seed, mix=(0.5, 0.3, 0.2), n_theta=4):
    atm = synth.solar_atmosphere()
    nus = synth.tracing_grid(lam0, lam1, R=R)
    lines = synth.synth_lines(nus, atm, n_lines, seed=seed, mix=mix)
    th, w = synth.thetas_and_weights(n_theta)
    return atm, nus, lines, synth.synth_continuum_state(atm), th, w


@pytest.mark.parametrize("n_lines", [700, 9000])  # short lists are scanned completely; long ones go through hlist / wlist
def test_far_field_agrees_with_the_direct_sum_and_the_oracle(far_ctx, n_lines):
    atm, nus, lines, cont, th, w = workload(4000.0, 4300.0, 1.0e5, n_lines, seed=5)
    assert 4096 < nus.size < 32768  # below the automatic threshold: the option decides
    (line0, total0, F0), ran0 = run(far_ctx, 0, nus, atm, lines, cont, th, w)
    (line1, total1, F1), ran1 = run(far_ctx, 1, nus, atm, lines, cont, th, w)
    assert not ran0 and ran1
    assert not np.array_equal(line0, line1)  # (it is a different sum)
    assert np.array_equal(line0 == 0, line1 == 0)
    assert rel_err(line1, line0) < FAR_VS_DIRECT and rel_err(total1, total0) < FAR_VS_DIRECT
    assert rel_err(F1[1:], F0[1:]) < 1e-11
    ref = oracle.calc_alan_entries(56, nus, lines["line_nus"], lines["doppler_widths"], lines["gammas"], lines["alphas"])
    assert np.array_equal(line1 == 0, ref == 0)
    assert rel_err(line1, ref) < 1e-12


def test_automatic_choice_follows_the_global_grid(far_ctx):
    """-1 (the default): grids of >= 32768 frequencies — decided from the GLOBAL grid, so a narrow shard of a long grid still runs it"""
    atm, nus, lines, cont, th, w = workload(4000.0, 4300.0, 1.0e5, 700, seed=6)
    _, ran = run(far_ctx, -1, nus, atm, lines, cont, th, w)
    assert not ran
    atm, nus, lines, cont, th, w = workload(4000.0, 5800.0, 1.0e5, 700, seed=6)
    assert nus.size >= 32768
    _, ran = run(far_ctx, -1, nus, atm, lines, cont, th, w)
    assert ran
    _, ran = run(far_ctx, -1, nus, atm, lines, cont, th, w, shard=(1000, 700))
    assert ran


@pytest.mark.parametrize("n_lines,world", [(700, 3), (9000, 5)])
def test_shards_reproduce_the_unsharded_far_field_bit_for_bit(far_ctx, n_lines, world):
    atm, nus, lines, cont, th, w = workload(4000.0, 5000.0, 1.0e5, n_lines, seed=7)
    (line, total, F), ran = run(far_ctx, 1, nus, atm, lines, cont, th, w)
    assert ran
    for rank in range(world):
        b, c = shard_bounds(nus.size, world, rank)
        (l, t, f), ran = run(far_ctx, 1, nus, atm, lines, cont, th, w, shard=(b, c))
        assert ran
        assert np.array_equal(l, line[:, b:b + c]) and np.array_equal(t, total[:, b:b + c]) and np.array_equal(f, F[:, b:b + c]), rank
    # ... and shards that begin and end inside a tile, one of them narrower than a tile
    for b, c in ((100, 300), (255, 2), (256 * 7 + 13, 256 * 3 + 1), (nus.size - 77, 77)):
        (l, t, f), _ = run(far_ctx, 1, nus, atm, lines, cont, th, w, shard=(b, c))
        assert np.array_equal(l, line[:, b:b + c]) and np.array_equal(f, F[:, b:b + c]), (b, c)


def test_grids_with_jumps_in_their_spacing_and_ragged_ends(far_ctx):
    """The distance test is made in index space from the tile's own end frequencies: segments of very different resolving power in
    one grid (tiles ten times wider than their neighbours), a grid a few points longer than a whole number of tiles, one shorter
    than a single tile."""
    atm = synth.solar_atmosphere()
    cont = synth.synth_continuum_state(atm)
    th, w = synth.thetas_and_weights(4)
    seg = [synth.tracing_grid(4000.0, 4100.0, R=3.0e5), synth.tracing_grid(4100.0, 4400.0, R=3.0e4)[1:], synth.tracing_grid(4400.0, 4450.0, R=6.0e5)[1:]]
    nus = np.concatenate(seg[::-1]) if seg[0][0] < seg[-1][0] else np.concatenate(seg)
    nus = np.sort(nus)[::-1].copy()
    assert np.all(np.diff(nus) < 0)
    for n in (nus.size, (nus.size // 256) * 256 + 3, 200):
        g = nus[:n].copy()
        lines = synth.synth_lines(g, atm, 600, seed=8, mix=(0.4, 0.3, 0.3))
        (line0, _, F0), _ = run(far_ctx, 0, g, atm, lines, cont, th, w)
        (line1, _, F1), ran = run(far_ctx, 1, g, atm, lines, cont, th, w)
        assert ran
        assert np.array_equal(line0 == 0, line1 == 0)
        assert rel_err(line1, line0) < FAR_VS_DIRECT, n
        assert rel_err(F1[1:], F0[1:]) < 1e-11, n
        if n == 200: assert np.array_equal(line0, line1)  # no full tile: no far field, the third plane is zeros


def test_mixed_precision_keeps_its_tolerance_with_the_far_field(far_ctx):
    atm, nus, lines, cont, th, w = workload(4000.0, 4300.0, 1.0e5, 9000, seed=9)
    (line64, _, F64), _ = run(far_ctx, 0, nus, atm, lines, cont, th, w)
    far_ctx.set_option("mixed_precision", 1)
    try:
        (line0, _, F0), _ = run(far_ctx, 0, nus, atm, lines, cont, th, w)
        (line1, _, F1), ran = run(far_ctx, 1, nus, atm, lines, cont, th, w)
    finally:
        far_ctx.set_option("mixed_precision", 0)
    assert ran
    assert rel_err(F0[1:], F64[1:]) < 1e-4 and rel_err(F1[1:], F64[1:]) < 1e-4
    # (the far items are evaluated in fp64 at the nodes: the tolerance path gets no further from the fp64 sum)
    assert rel_err(line0, line64) < 5e-5 and rel_err(line1, line64) < 5e-5


def test_a_column_of_gammas_and_a_deep_model(far_ctx):
    """gamma as an (N_l, 1) column (the molecular lists), and a model with more depth points than a wave has lanes"""
    atm, nus, lines, cont, th, w = workload(5000.0, 5300.0, 1.0e5, 12000, seed=10)
    lines["gammas"] = np.ascontiguousarray(lines["gammas"][:, :1])
    (line0, _, F0), _ = run(far_ctx, 0, nus, atm, lines, cont, th, w)
    (line1, _, F1), ran = run(far_ctx, 1, nus, atm, lines, cont, th, w)
    assert ran and rel_err(line1, line0) < FAR_VS_DIRECT and rel_err(F1[1:], F0[1:]) < 1e-11
    from test_gpu_engine import deep_atmosphere

    atm = deep_atmosphere(150)
    lines = synth.synth_lines(nus, atm, 500, seed=11, mix=(0.5, 0.3, 0.2))
    cont = synth.synth_continuum_state(atm)
    (line0, _, F0), _ = run(far_ctx, 0, nus, atm, lines, cont, th, w)
    (line1, _, F1), ran = run(far_ctx, 1, nus, atm, lines, cont, th, w)
    assert ran and line1.shape[0] == 150 and rel_err(line1, line0) < FAR_VS_DIRECT and rel_err(F1[1:], F0[1:]) < 1e-11


def test_the_stand_alone_line_opacity_entry_point_takes_the_option_too(far_ctx):
    """sdx_line_opacity_f64 (host buffers; what the mirror's calc_alan_entries calls): three partial planes reduced instead of two,
    the evaluation count — the reference's window sizes, whoever evaluates them — unchanged"""
    from stardis_amd import ops

    atm, nus, lines, cont, th, w = workload(4200.0, 4500.0, 1.0e5, 900, seed=12)
    args = (56, nus, lines["line_nus"], lines["doppler_widths"], lines["gammas"], lines["alphas"])
    far_ctx.set_option("far_field", 0)
    a0, n0 = ops.calc_alan_entries(*args, return_evaluations=True, ctx=far_ctx)
    far_ctx.set_option("far_field", 1)
    a1, n1 = ops.calc_alan_entries(*args, return_evaluations=True, ctx=far_ctx)
    assert n0 == n1 and not np.array_equal(a0, a1)
    assert np.array_equal(a0 == 0, a1 == 0) and rel_err(a1, a0) < FAR_VS_DIRECT
    ref, evals = oracle.calc_alan_entries(*args, return_evals=True)
    assert n1 == evals and rel_err(a1, ref) < 1e-12
