"""Seeded random configurations of the LONG-list path (>= 8192 lines: hlist / wlist candidate lists, the indexed wide walk, F
frequencies per narrow wave, k_raytrace<1> or the segmented kernel by grid size, deep models with two depth blocks in the
pre-pass) against the oracle on a subset of columns with every line present, and sharded — equal and cost-balanced shards, i.e.
the culled pre-pass with its classification stream, range and gather blocks — bit for bit against the unsharded run.
scripts/fuzz_long_lists.py runs further seeds."""
import numpy as np
import pytest

import oracle
from conftest import rel_err
from stardis_amd import constants as K, parallel, synth
from stardis_amd.engine import SpectralSynthesizer, shard_bounds

pytestmark = pytest.mark.gpu


def long_case(seed):
    rng = np.random.default_rng(7000 + seed)
    atm0 = synth.cool_dwarf_atmosphere() if seed % 2 else synth.solar_atmosphere()
    n_depth = int(rng.choice([7, 40, 56, 64, 65, 90]))
    x_old, x_new = np.linspace(0.0, 1.0, atm0["temperatures"].size), np.linspace(0.0, 1.0, n_depth)
    atm = dict(atm0)
    for k in ("temperatures", "r"):
        atm[k] = np.interp(x_new, x_old, atm0[k])
    for k in ("n_e", "n_h"):
        atm[k] = np.exp(np.interp(x_new, x_old, np.log(atm0[k])))
    atm["dist"] = np.diff(atm["r"])
    lam0 = rng.uniform(3200.0, 8000.0)
    R = float(rng.choice([5.0e4, 1.0e5, 3.0e5]))
    n_nu = int(rng.integers(3000, 40000))
    nus = synth.tracing_grid(lam0, lam0 * (1.0 + 1.02 * n_nu / R), R=R)[:n_nu]
    n_lines = int(rng.integers(8192, 30000))
    mix = [(0.7, 0.25, 0.05), (0.9, 0.09, 0.01), (0.5, 0.3, 0.2)][int(rng.integers(0, 3))]
    lines = synth.synth_lines(nus, atm, n_lines, seed=seed, gamma_per_depth=bool(rng.integers(0, 2)), mix=mix)
    n_theta = int(rng.choice([4, 20, 24]))
    th, w = synth.thetas_and_weights(n_theta)
    return atm, nus, lines, synth.synth_continuum_state(atm), th, w


def check_long_case(ctx, seed):
    atm, nus, lines, cont, th, w = long_case(seed)
    nd = atm["temperatures"].size
    syn = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, lines, cont, ctx=ctx, track_evaluations=False)
    syn.step()
    line, total, F = syn.alpha_line(), syn.total_alphas(), syn.F_nu()
    cols = np.unique(np.concatenate([np.arange(3, nus.size, max(1, nus.size // 24)), [0, nus.size - 1]]))
    line_ref = oracle.calc_alan_entries_columns(cols, nd, nus, lines["line_nus"], lines["doppler_widths"], lines["gammas"], lines["alphas"])
    assert np.array_equal(line[:, cols] == 0, line_ref == 0)
    assert rel_err(line[:, cols], line_ref) < 1e-12
    F_cpu, _ = oracle.raytrace(nus[cols], atm["temperatures"], atm["dist"], th, w, total[:, cols])
    assert np.all(F[0] == 0) and rel_err(F[1:, cols], F_cpu[1:]) < 1e-10
    worlds = [(3, None), (5, parallel.column_cost(nus, lines))]
    for world, cost in worlds:
        shards = parallel.balanced_shards(cost, world) if cost is not None else [shard_bounds(nus.size, world, r) for r in range(world)]
        for b, c in shards:
            s = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, lines, cont, ctx=ctx, shard=(b, c), track_evaluations=False)
            s.step()
            assert np.array_equal(s.F_nu(), F[:, b:b + c]), (seed, world, b, c)
            assert np.array_equal(s.alpha_line(), line[:, b:b + c]) and np.array_equal(s.total_alphas(), total[:, b:b + c])
            s.close()
    syn.close()
    return nd, nus.size, lines["line_nus"].size, th.size


def check_long_case_mixed(ctx, seed):
    """the fp32-mixed tolerance path on the same configurations: within its stated 1e-4 of the fp64 result, and — like fp64 —
    the same bits from shards as from the whole grid"""
    atm, nus, lines, cont, th, w = long_case(seed)
    ref = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, lines, cont, ctx=ctx, track_evaluations=False)
    ref.step()
    line64, F64 = ref.alpha_line(), ref.F_nu()
    ref.close()
    ctx.set_option("mixed_precision", 1)
    try:
        syn = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, lines, cont, ctx=ctx, track_evaluations=False)
        syn.step()
        line, F = syn.alpha_line(), syn.F_nu()
        ea = float(np.max(np.abs(line - line64)) / max(np.abs(line64).max(), 1e-300))
        ef = float(np.max(np.abs(F - F64) / np.maximum(np.abs(F64).max(axis=0, keepdims=True), 1e-300)))
        assert ea < 1e-4 and ef < 1e-4, (seed, ea, ef)
        for b, c in [shard_bounds(nus.size, 3, r) for r in range(3)]:
            s = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, lines, cont, ctx=ctx, shard=(b, c), track_evaluations=False)
            s.step()
            assert np.array_equal(s.F_nu(), F[:, b:b + c]) and np.array_equal(s.alpha_line(), line[:, b:b + c]), (seed, b, c)
            s.close()
        syn.close()
    finally:
        ctx.set_option("mixed_precision", 0)
    return ea, ef


@pytest.mark.parametrize("seed", range(4))
def test_random_long_list_configuration(ctx, seed):
    check_long_case(ctx, seed)


# (1306: a shard that owns only part of a far-field unit of eight tiles — round 6's packed-fp32 far role first summed a hit's nodes with
# one rounding when every tile of the unit was far and with two when a tile the shard does not own was not: found by this fuzzer)
@pytest.mark.parametrize("seed", [0, 1, 1306])
def test_random_long_list_configuration_mixed_precision(ctx, seed):
    check_long_case_mixed(ctx, seed)


def test_culled_prepass_drawing_its_work_from_a_counter(ctx):
    """Lists of >= 5e5 lines prepare a shard's lines with as many workgroups as the chip holds, each drawing the blocks that have
    work from a device counter (k_line_prepass_ticket) instead of one workgroup per candidate block.  Forced here on a small long
    list: shards — equal and balanced, fp64 and mixed — still reproduce the unsharded bits."""
    ctx.set_option("prepass_ticket_min_blocks", 0)
    try:
        for seed in (1, 2):
            check_long_case(ctx, seed)
        check_long_case_mixed(ctx, 3)
    finally:
        ctx.set_option("prepass_ticket_min_blocks", 16384)

