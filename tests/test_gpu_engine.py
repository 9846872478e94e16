"""Device-resident fused synthesis (stardis_amd.engine): equals the step-by-step entry points bit for bit,
is invariant under frequency sharding and graph replay, and matches the CPU oracle within the path's tolerance."""
import numpy as np
import pytest

import oracle
from conftest import rel_err
from stardis_amd import constants as K
from stardis_amd import synth
from stardis_amd.engine import SpectralSynthesizer, shard_bounds

pytestmark = pytest.mark.gpu


def small_workload(n_lines=300, step=0.02, seed=11, n_theta=8):
    atm = synth.solar_atmosphere()
    nus = synth.tracing_grid(6560.0, 6570.0, step=step)
    lines = synth.synth_lines(nus, atm, n_lines, seed=seed, mix=(0.8, 0.15, 0.05))
    th, w = synth.thetas_and_weights(n_theta)
    return atm, nus, lines, synth.synth_continuum_state(atm), th, w


def oracle_total(atm, nus, lines, cont):
    nd = atm["temperatures"].size
    cutoff = (cont["ionization_energy"] - cont["level_excitation"]) / K.H_CGS
    total = oracle.alpha_file_1d(K.nu_to_angstrom(nus), cont["hminus_bf_wavelength"], cont["hminus_bf_cross_section"], cont["n_hminus"])
    total = total + oracle.alpha_bf(nus, [0, len(cutoff)], [0], cutoff, cont["level_density"])
    total = total + oracle.alpha_ff(nus, atm["temperatures"], [1], cont["n_e"] * cont["n_h2"])
    total = total + oracle.alpha_electron(nus.size, cont["n_e"])
    line = oracle.calc_alan_entries(nd, nus, lines["line_nus"], lines["doppler_widths"], lines["gammas"], lines["alphas"])
    return total + line, line


def test_fused_matches_oracle(ctx):
    atm, nus, lines, cont, th, w = small_workload()
    syn = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, lines, cont, ctx=ctx)
    syn.step()
    total_ref, line_ref = oracle_total(atm, nus, lines, cont)
    assert rel_err(syn.alpha_line(), line_ref) < 1e-12
    assert rel_err(syn.total_alphas(), total_ref) < 1e-12
    F_ref, _ = oracle.raytrace(nus, atm["temperatures"], atm["dist"], th, w, total_ref)
    F = syn.F_nu()
    assert np.all(F[0] == 0)
    assert rel_err(F[1:], F_ref[1:]) < 1e-10
    _, evals = oracle.calc_alan_entries(56, nus, lines["line_nus"], lines["doppler_widths"], lines["gammas"], lines["alphas"], return_evals=True)
    assert syn.evaluations() == evals


def test_fused_equals_unfused_and_graph_replay(ctx):
    atm, nus, lines, cont, th, w = small_workload(seed=12)
    syn = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, lines, cont, ctx=ctx)
    syn.enqueue()
    a = (syn.alpha_line(), syn.total_alphas(), syn.F_nu())
    syn.enqueue_unfused()
    b = (syn.alpha_line(), syn.total_alphas(), syn.F_nu())
    for x, y in zip(a, b):
        assert np.array_equal(x, y)
    syn.capture()
    for _ in range(3):
        syn.step()
    c = (syn.alpha_line(), syn.total_alphas(), syn.F_nu())
    for x, y in zip(a, c):
        assert np.array_equal(x, y)
    syn.close()


@pytest.mark.parametrize("world", [2, 3, 8])
def test_frequency_shards_reassemble_bit_exactly(ctx, world):
    """Each shard uses the global window rule (global d_nu, global line centres, global clamp): the union of the
    shards is the single-GPU answer."""
    atm, nus, lines, cont, th, w = small_workload(n_lines=200, seed=13)
    full = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, lines, cont, ctx=ctx)
    full.step()
    F_full, tot_full = full.F_nu(), full.total_alphas()
    parts_F, parts_t = [], []
    for rank in range(world):
        begin, count = shard_bounds(nus.size, world, rank)
        s = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, lines, cont, ctx=ctx, shard=(begin, count))
        s.step()
        parts_F.append(s.F_nu())
        parts_t.append(s.total_alphas())
    assert np.array_equal(np.concatenate(parts_F, axis=1), F_full)
    assert np.array_equal(np.concatenate(parts_t, axis=1), tot_full)


def test_no_lines_and_gamma_column(ctx):
    atm, nus, lines, cont, th, w = small_workload(n_lines=50, seed=14)
    empty = dict(line_nus=np.zeros(0), doppler_widths=np.zeros((0, 56)), gammas=np.zeros((0, 1)), alphas=np.zeros((0, 56)))
    syn = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, empty, cont, ctx=ctx)
    syn.step()
    assert not syn.alpha_line().any()
    assert np.isfinite(syn.F_nu()).all() and (syn.F_nu()[-1] > 0).all()
    col = synth.synth_lines(nus, atm, 80, seed=15, gamma_per_depth=False)
    syn = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, col, cont, ctx=ctx)
    syn.step()
    ref = oracle.calc_alan_entries(56, nus, col["line_nus"], col["doppler_widths"], col["gammas"], col["alphas"])
    assert rel_err(syn.alpha_line(), ref) < 1e-12


def test_full_size_properties(ctx):
    """BASELINE configs[1] (S-c2: 7634 frequencies, 2000 lines) at full size through size-independent properties:
    doubling every line strength doubles the line opacity of unchanged windows exactly where windows saturate,
    and the flux equals the oracle on a strided subset of columns recomputed from the GPU's own total opacity."""
    w = synth.make_workload("S-c2")
    atm, nus, lines = w["atm"], w["nus"], w["lines"]
    syn = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], w["thetas"], w["weights"], lines, w["cont"], ctx=ctx)
    syn.step()
    F, total, line = syn.F_nu(), syn.total_alphas(), syn.alpha_line()
    assert np.isfinite(F).all() and (F[-1] > 0).all() and (line >= 0).all()
    # the raytrace is column-independent: recompute a strided subset of columns on the CPU from the same total
    cols = np.arange(0, nus.size, 37)
    F_ref, _ = oracle.raytrace(nus[cols], atm["temperatures"], atm["dist"], w["thetas"], w["weights"], np.ascontiguousarray(total[:, cols]))
    assert rel_err(F[1:, cols], F_ref[1:]) < 1e-10
    # line opacity on a strided subset of lines against the oracle (linearity in the line list: subset sum)
    sub = {k: np.ascontiguousarray(v[::40]) for k, v in lines.items()}
    s2 = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], w["thetas"], w["weights"], sub, w["cont"], ctx=ctx)
    s2.step()
    ref = oracle.calc_alan_entries(56, nus, sub["line_nus"], sub["doppler_widths"], sub["gammas"], sub["alphas"])
    assert rel_err(s2.alpha_line(), ref) < 1e-12


def test_indexed_wide_path_matches_direct_path_and_oracle(ctx):
    """Long line lists go through the compact line lists (`wlist`: medium lines found by centre range through `wrank`;
    `hlist`: huge ones scanned by every tile); forced here on a small list: same windows, same evaluations, results equal to the direct path up to the summation
    order (huge before medium instead of pure line order) and to the oracle within the opacity tolerance."""
    atm, nus, lines, cont, th, w = small_workload(n_lines=700, step=0.005, seed=21)
    ref = oracle.calc_alan_entries(56, nus, lines["line_nus"], lines["doppler_widths"], lines["gammas"], lines["alphas"])
    out = {}
    try:
        for name, minimum in (("indexed", 1), ("direct", 1 << 40)):
            ctx.set_option("indexed_min_lines", minimum)
            syn = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, lines, cont, ctx=ctx)
            syn.step()
            out[name] = (syn.alpha_line(), syn.F_nu())
            assert rel_err(out[name][0], ref) < 1e-12, name
        assert rel_err(out["indexed"][0], out["direct"][0]) < 1e-13
        assert rel_err(out["indexed"][1][1:], out["direct"][1][1:]) < 1e-12
        # shard invariance of the indexed path
        ctx.set_option("indexed_min_lines", 1)
        parts = []
        for rank in range(3):
            s = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, lines, cont, ctx=ctx, shard=shard_bounds(nus.size, 3, rank))
            s.step()
            parts.append(s.alpha_line())
        assert np.array_equal(np.concatenate(parts, axis=1), out["indexed"][0])
    finally:
        ctx.set_option("indexed_min_lines", 8192)


def test_indexed_path_on_a_reused_context_with_a_shorter_list(ctx):
    """A context that ran a long list keeps its class-mask scratch: the tail entries of the last 64-line word are never
    written by the pre-pass of a SHORTER list (n_lines % 64 in 1..32), so stale bits of the earlier list must be masked
    off when the dense lists are counted and built (round-1 advisor finding)."""
    atm = synth.solar_atmosphere()
    nus = synth.tracing_grid(6560.0, 6570.0, step=0.01)
    th, w = synth.thetas_and_weights(4)
    cont = synth.synth_continuum_state(atm)
    try:
        ctx.set_option("indexed_min_lines", 1)
        for n_lines, mix, seed in ((1400, (0.0, 0.5, 0.5), 31), (530, (0.3, 0.4, 0.3), 32), (8192 + 17, (0.9, 0.09, 0.01), 33), (65, (0.0, 0.5, 0.5), 34)):
            lines = synth.synth_lines(nus, atm, n_lines, seed=seed, mix=mix)
            syn = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, lines, cont, ctx=ctx)
            syn.step()
            ref, evals = oracle.calc_alan_entries(56, nus, lines["line_nus"], lines["doppler_widths"], lines["gammas"], lines["alphas"], return_evals=True)
            assert syn.evaluations() == evals, n_lines
            assert rel_err(syn.alpha_line(), ref) < 1e-12, n_lines
    finally:
        ctx.set_option("indexed_min_lines", 8192)


def test_graph_captured_before_a_workspace_reallocation_is_refused_and_recaptured():
    """A hipGraph bakes in the context's scratch pointers.  When a larger synthesis on the same context makes the library
    reallocate its scratch, replaying the old graph would touch freed memory (round-1 advisor finding): the C ABI refuses it
    (SDX_ERR_STALE) and SpectralSynthesizer.step captures again."""
    from stardis_amd import _lib

    own = _lib.Context(0)  # a context of its own: the shared one has already grown to the largest test
    atm, nus, lines, cont, th, w = small_workload(n_lines=100, seed=51)
    small = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, lines, cont, ctx=own)
    small.capture()
    small.step()
    want = (small.alpha_line(), small.F_nu())
    stale_handle = small.graph
    nus_big = synth.tracing_grid(6500.0, 6600.0, R=2.0e5)
    big_lines = synth.synth_lines(nus_big, atm, 3000, seed=52)
    big = SpectralSynthesizer(nus_big, atm["temperatures"], atm["dist"], th, w, big_lines, cont, ctx=own)
    big.step()  # grows every scratch buffer of the context
    own.synchronize()
    with pytest.raises(_lib.StaleGraphError):
        own.call("sdx_graph_launch", stale_handle)
    small.step()  # re-captures
    assert small.graph is not None and small.graph is not stale_handle
    got = (small.alpha_line(), small.F_nu())
    for x, y in zip(want, got):
        assert np.array_equal(x, y)
    big.step()
    ref = oracle.calc_alan_entries(56, nus_big, big_lines["line_nus"], big_lines["doppler_widths"], big_lines["gammas"], big_lines["alphas"])
    assert rel_err(big.alpha_line(), ref) < 1e-12
    small.close()
    own.close()


@pytest.mark.parametrize("world", [2, 5])
def test_shards_of_a_long_list_prepare_only_the_lines_they_need_and_change_nothing(ctx, world):
    """Frequency shards of a long list (>= indexed_min_lines lines, grid too long for the in-block d_nu scan) run a culled
    pre-pass: a classification pass + the lines centred within kMediumHalfWidth of the shard + the lines wide enough to
    reach any column.  The union of the shards must still be the unsharded answer bit for bit (which lines a shard
    prepares does not change what it computes), here with windows of every class: 10-point floors, medium windows whose
    centres lie OUTSIDE the shard they reach into, and lines spanning the whole grid."""
    atm = synth.solar_atmosphere()
    nus = synth.tracing_grid(4000.0, 5000.0, R=1.0e5)
    assert nus.size > 16384
    lines = synth.synth_lines(nus, atm, 9000, seed=61, mix=(0.7, 0.25, 0.05))
    cont = synth.synth_continuum_state(atm)
    th, w = synth.thetas_and_weights(4)
    full = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, lines, cont, ctx=ctx, track_evaluations=False)
    full.step()
    F_full, line_full = full.F_nu(), full.alpha_line()
    ref = oracle.calc_alan_entries(56, nus, lines["line_nus"], lines["doppler_widths"], lines["gammas"], lines["alphas"])
    assert rel_err(line_full, ref) < 1e-12
    parts_F, parts_l = [], []
    for rank in range(world):
        s = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, lines, cont, ctx=ctx, shard=shard_bounds(nus.size, world, rank),
                                track_evaluations=False)  # the evaluation counter needs every window: it switches the culling off
        s.step()
        parts_F.append(s.F_nu())
        parts_l.append(s.alpha_line())
        s.capture()  # the culled pre-pass is decided on the device: it replays as a graph
        s.step()
        assert np.array_equal(s.F_nu(), parts_F[-1])
        s.close()
    assert np.array_equal(np.concatenate(parts_l, axis=1), line_full)
    assert np.array_equal(np.concatenate(parts_F, axis=1), F_full)


def test_culled_shards_on_a_context_dirtied_by_another_line_list(ctx):
    """Tiles of the wide role are aligned to the global grid, so a tile cut by a shard boundary looks up to 255 points beyond
    the shard; a culled pre-pass prepares only the lines that can reach the shard's OWN columns.  The lines centred in that
    band must not be walked: their scan words and records are whatever the workspace held — here those of a different,
    longer list made of wide lines only, run on the same context just before (round-3 advisor finding).  The expected
    planes come from another context."""
    from stardis_amd import _lib

    atm = synth.solar_atmosphere()
    nus = synth.tracing_grid(4000.0, 5000.0, R=1.0e5)
    cont = synth.synth_continuum_state(atm)
    th, w = synth.thetas_and_weights(4)
    lines = synth.synth_lines(nus, atm, 9000, seed=71, mix=(0.6, 0.35, 0.05))
    full = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, lines, cont, ctx=ctx, track_evaluations=False)
    full.step()
    F_full, line_full = full.F_nu(), full.alpha_line()
    own = _lib.Context(0)
    try:
        for world in (3, 7):
            for rank in range(world):
                dirt = synth.synth_lines(nus, atm, 11000 + 13 * rank, seed=72 + rank, mix=(0.0, 0.6, 0.4))
                d = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, dirt, cont, ctx=own, track_evaluations=False)
                d.step()  # unsharded: every scan word and record of the workspace now belongs to `dirt`
                begin, count = shard_bounds(nus.size, world, rank)
                s = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, lines, cont, ctx=own, shard=(begin, count), track_evaluations=False)
                s.step()
                assert np.array_equal(s.alpha_line(), line_full[:, begin:begin + count]), (world, rank)
                assert np.array_equal(s.F_nu(), F_full[:, begin:begin + count]), (world, rank)
    finally:
        own.close()


def test_long_line_list_on_the_wide_grid(ctx):
    """BASELINE configs[2]/[3] shape: the 3000-10000 A grid at R = 1e5 (120 398 frequencies) with a line list long
    enough (20 000 lines, gamma given as an (N_l, 1) column like the molecular case) to take the indexed wide-window
    path by default.  Line opacity against the oracle on the full grid; flux through the column-independence of the
    formal solution (a strided subset of columns recomputed on the CPU from the GPU's own total opacity)."""
    w = synth.make_workload("S-c4", n_lines=20000)
    atm, nus, lines = w["atm"], w["nus"], w["lines"]
    assert lines["gammas"].shape == (20000, 1) and nus.size == 120398
    syn = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], w["thetas"], w["weights"], lines, w["cont"], ctx=ctx)
    syn.step()
    line, total, F = syn.alpha_line(), syn.total_alphas(), syn.F_nu()
    ref, evals = oracle.calc_alan_entries(56, nus, lines["line_nus"], lines["doppler_widths"], lines["gammas"], lines["alphas"], return_evals=True)
    assert syn.evaluations() == evals
    assert np.array_equal(line == 0, ref == 0)
    assert rel_err(line, ref) < 1e-12
    cols = np.arange(0, nus.size, 601)
    F_ref, _ = oracle.raytrace(nus[cols], atm["temperatures"], atm["dist"], w["thetas"], w["weights"], np.ascontiguousarray(total[:, cols]))
    assert rel_err(F[1:, cols], F_ref[1:]) < 1e-10


def deep_atmosphere(n_depth):
    """The solar structure resampled to n_depth points (MESA-like models have hundreds; MARCS always 56)."""
    atm = synth.solar_atmosphere()
    x_old = np.linspace(0.0, 1.0, atm["temperatures"].size)
    x_new = np.linspace(0.0, 1.0, n_depth)
    out = dict(atm)
    for k in ("temperatures", "n_e", "n_h", "r"):
        out[k] = np.interp(x_new, x_old, atm[k]) if k != "n_e" and k != "n_h" else np.exp(np.interp(x_new, x_old, np.log(atm[k])))
    out["dist"] = np.diff(out["r"])
    return out


@pytest.mark.parametrize("n_depth,n_theta", [(150, 8), (56, 70), (200, 20)])
def test_deep_models_and_many_angles(ctx, n_depth, n_theta):
    """More than 64 depth points (several pre-pass depth blocks, narrow kernel in depth chunks, LDS-capped raytrace
    groups) and more than 64 angles (un-fused total, chunked raytrace) against the oracle."""
    atm = deep_atmosphere(n_depth)
    nus = synth.tracing_grid(6560.0, 6566.0, step=0.02)
    lines = synth.synth_lines(nus, atm, 150, seed=41, mix=(0.8, 0.15, 0.05))
    cont = synth.synth_continuum_state(atm)
    th, w = synth.thetas_and_weights(n_theta)
    syn = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, lines, cont, ctx=ctx)
    syn.step()
    ref = oracle.calc_alan_entries(n_depth, nus, lines["line_nus"], lines["doppler_widths"], lines["gammas"], lines["alphas"])
    assert rel_err(syn.alpha_line(), ref) < 1e-12
    cutoff = (cont["ionization_energy"] - cont["level_excitation"]) / K.H_CGS
    total = oracle.alpha_file_1d(K.nu_to_angstrom(nus), cont["hminus_bf_wavelength"], cont["hminus_bf_cross_section"], cont["n_hminus"])
    total = total + oracle.alpha_bf(nus, [0, len(cutoff)], [0], cutoff, cont["level_density"])
    total = total + oracle.alpha_ff(nus, atm["temperatures"], [1], cont["n_e"] * cont["n_h2"])
    total = total + oracle.alpha_electron(nus.size, cont["n_e"]) + ref
    assert rel_err(syn.total_alphas(), total) < 1e-12
    F_ref, _ = oracle.raytrace(nus, atm["temperatures"], atm["dist"], th, w, total)
    assert rel_err(syn.F_nu()[1:], F_ref[1:]) < 1e-10
    # windows through the stand-alone entry point as well (bit-exact)
    from stardis_amd import ops

    lo, hi = ops.line_windows(n_depth, nus, lines["line_nus"], lines["doppler_widths"], lines["gammas"], lines["alphas"])
    for l in (0, 75, 149):
        for d in (0, n_depth // 2, n_depth - 1):
            assert (lo[l, d], hi[l, d]) == oracle.window(nus, lines["line_nus"][l], lines["gammas"][l, d], lines["doppler_widths"][l, d], lines["alphas"][l, d])


def test_fp32_formal_solution_on_a_nearly_isothermal_gap_ahead_of_a_thin_one(ctx):
    """The second-order terms of the recurrence (radiation_field_solvers/base.py:208-249) multiply the source DIFFERENCES by up to
    1 / tau of a thin gap.  Formed from two rounded fp32 source values, a difference of 0.1 % of the source carries 3e-4 of itself
    (scripts/fuzz_raytrace.py, seed 849: 2.9e-4 of the flux); the fp32 kernel stages the differences themselves (planck32_pair:
    S_d E expm1(x' - x) / (E' - 1) with T - T' from the fp64 temperatures) and stays two orders below its stated 1e-4."""
    from stardis_amd import ops

    n_depth, n_theta = 40, 20
    rng = np.random.default_rng(849)
    temps = np.sort(rng.uniform(9000.0, 9700.0, n_depth))[::-1].copy()
    temps[16], temps[17] = 9590.1, 9586.7                     # 0.035 % apart: dS / S = 1.4e-3
    dist = rng.uniform(2e5, 4e7, n_depth - 1)
    nus = np.linspace(9.0e14, 6.0e14, 96)
    alphas = 10.0 ** rng.uniform(-9.0, -7.0, (n_depth, nus.size))
    alphas[15] = 1.4e-3                                       # an opaque point, then two thin gaps: tau ~ 0.4 ahead of tau ~ 1e-6
    alphas[16], alphas[17] = 4e-13, 1.5e-14
    th, w = synth.thetas_and_weights(n_theta)
    rd = dist.reshape(-1, 1) / np.cos(th)
    F64, _ = ops.raytrace_arrays(nus, temps, rd, w, alphas, ctx=ctx)
    try:
        ctx.set_option("mixed_precision", 1)
        ctx.set_option("segmented_raytrace", 0)  # (a grid this small would take the segmented fp64 kernel whatever the mode)
        F32, _ = ops.raytrace_arrays(nus, temps, rd, w, alphas, ctx=ctx)
    finally:
        ctx.set_option("mixed_precision", 0)
        ctx.set_option("segmented_raytrace", -1)
    assert np.isfinite(F64).all() and not np.array_equal(F32, F64)
    assert F64[17].max() > 20 * F64[15].max()                 # the amplified second-order term dominates the flux there
    dev = np.max(np.abs(F32 - F64) / np.abs(F64).max(axis=0, keepdims=True))
    print("fp32 formal solution, amplified source difference: max dev", dev)
    assert dev < 2e-5


def test_mixed_precision_mode_is_a_tolerance_path(ctx):
    """BASELINE config 5's tolerance path: far-wing evaluations and window edges in packed fp32, narrow windows through the
    fp32 Faddeeva routine (all four regions), cores kept by wide windows, continuum and formal solution in fp64.  The mode's
    stated tolerance is 1e-4 on the flux; on this workload opacity and flux stay within 1e-5 (observed ~7e-6 / ~1e-6).
    fp64 remains the default and the parity path."""
    atm, nus, lines, cont, th, w = small_workload(n_lines=400, step=0.005, seed=23)
    ref = oracle.calc_alan_entries(56, nus, lines["line_nus"], lines["doppler_widths"], lines["gammas"], lines["alphas"])
    syn64 = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, lines, cont, ctx=ctx)
    syn64.step()
    a64, F64 = syn64.alpha_line(), syn64.F_nu()
    try:
        ctx.set_option("mixed_precision", 1)
        syn = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, lines, cont, ctx=ctx)
        syn.step()
        a32, F32 = syn.alpha_line(), syn.F_nu()
    finally:
        ctx.set_option("mixed_precision", 0)
    assert not np.array_equal(a32, a64)  # the mode really took the fp32 route
    assert rel_err(a32, ref) < 1e-5
    assert rel_err(F32[1:], F64[1:]) < 1e-5
    print("mixed precision: opacity rel err", rel_err(a32, a64), "flux", rel_err(F32[1:], F64[1:]))


def test_cool_dwarf_structure_matches_oracle(ctx):
    """The same path on the coolest MARCS structure the reference ships (its own test model, Teff 3800 K, 2771-7713 K):
    molecular-style list with gamma of shape (N_l, 1), other optical-depth regimes than the solar model."""
    atm = synth.cool_dwarf_atmosphere()
    nus = synth.tracing_grid(7050.0, 7065.0, step=0.01)  # a TiO band region
    lines = synth.synth_lines(nus, atm, 600, seed=21, gamma_per_depth=False, mix=(0.8, 0.15, 0.05))
    assert lines["gammas"].shape == (600, 1)
    cont = synth.synth_continuum_state(atm)
    th, w = synth.thetas_and_weights(10)
    syn = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, lines, cont, ctx=ctx)
    syn.step()
    total_ref, line_ref = oracle_total(atm, nus, lines, cont)
    assert rel_err(syn.alpha_line(), line_ref) < 1e-12
    assert rel_err(syn.total_alphas(), total_ref) < 1e-12
    F_ref, _ = oracle.raytrace(nus, atm["temperatures"], atm["dist"], th, w, total_ref)
    assert rel_err(syn.F_nu()[1:], F_ref[1:]) < 1e-10


def test_pool_members_equal_their_standalone_results(ctx):
    """Independent syntheses in flight on two streams (different line lists, one on the cool structure): each member's
    result is bit for bit what it computes alone."""
    from stardis_amd.engine import SynthesisPool

    cases = []
    for seed, cool in ((31, False), (32, False), (33, True)):
        atm = synth.cool_dwarf_atmosphere() if cool else synth.solar_atmosphere()
        nus = synth.tracing_grid(6560.0, 6570.0, step=0.02)
        lines = synth.synth_lines(nus, atm, 250, seed=seed, mix=(0.8, 0.15, 0.05))
        th, w = synth.thetas_and_weights(6)
        cases.append((nus, atm["temperatures"], atm["dist"], th, w, lines, synth.synth_continuum_state(atm)))
    alone = []
    for args in cases:
        syn = SpectralSynthesizer(*args, ctx=ctx)
        syn.step()
        alone.append(syn.F_nu())
    pool = SynthesisPool(n_streams=2)
    for args in cases:
        pool.add(*args)
    for _ in range(3):
        pool.step()
    for got, want in zip(pool.fluxes(), alone):
        assert np.array_equal(got, want)
    pool.close()


def test_host_buffer_synthesis_entry_point(ctx):
    """sdx_synthesize_f64: every array, including the continuum description, handed over as host (numpy) memory —
    the results are those of the device-resident engine, bit for bit."""
    import ctypes as C

    from stardis_amd import _lib

    atm, nus, lines, cont, th, w = small_workload(seed=17, n_theta=6)
    nd, n_nu = atm["temperatures"].size, nus.size
    syn = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, lines, cont, ctx=ctx)
    syn.step()
    keep = []

    def host(a, dt=np.float64):
        a = np.ascontiguousarray(a, dtype=dt)
        keep.append(a)
        return a.ctypes.data

    cutoff = (cont["ionization_energy"] - np.asarray(cont["level_excitation"])) / K.H_CGS
    c = _lib.Continuum()
    c.lambdas = host(K.nu_to_angstrom(nus))
    c.n_table = len(cont["hminus_bf_wavelength"])
    c.table_wavelength, c.table_sigma = host(cont["hminus_bf_wavelength"]), host(cont["hminus_bf_cross_section"])
    c.table_density = host(cont["n_hminus"])
    c.bf_n_species, c.bf_n_levels = 1, len(cutoff)
    c.bf_species_offsets, c.bf_species_ion_number = host([0, len(cutoff)], np.int32), host([0], np.int32)
    c.bf_cutoff, c.bf_level_density = host(cutoff), host(cont["level_density"])
    c.ff_n_species = 1
    c.ff_species_ion_number, c.ff_number_density = host([1], np.int32), host(np.asarray(cont["n_e"]) * np.asarray(cont["n_h2"]))
    c.electron_density = host(cont["n_e"])
    ray = np.asarray(atm["dist"]).reshape(-1, 1) / np.cos(th)
    line, total, F = np.empty((nd, n_nu)), np.empty((nd, n_nu)), np.empty((nd, n_nu))
    ev = C.c_int64(0)
    g = np.ascontiguousarray(lines["gammas"]).reshape(lines["line_nus"].size, -1)
    _lib.check(ctx.lib.sdx_synthesize_f64(
        ctx.handle, nd, n_nu, host(nus), lines["line_nus"].size, host(lines["line_nus"]), host(lines["doppler_widths"]), host(g), g.shape[1],
        host(lines["alphas"]), C.byref(c), th.size, host(atm["temperatures"]), host(ray), host(w), line.ctypes.data, total.ctypes.data,
        F.ctypes.data, C.byref(ev)))
    assert ev.value == syn.evaluations()
    assert np.array_equal(line, syn.alpha_line()) and np.array_equal(total, syn.total_alphas()) and np.array_equal(F, syn.F_nu())
    bad = nus[::-1].copy()
    with pytest.raises(ValueError, match="descending"):
        _lib.check(ctx.lib.sdx_synthesize_f64(
            ctx.handle, nd, n_nu, host(bad), lines["line_nus"].size, host(lines["line_nus"]), host(lines["doppler_widths"]), host(g), g.shape[1],
            host(lines["alphas"]), C.byref(c), th.size, host(atm["temperatures"]), host(ray), host(w), None, total.ctypes.data, F.ctypes.data, None))


def test_long_list_on_a_deep_model(ctx):
    """More than 64 depth points AND enough lines for the 32-lines-per-block pre-pass (two items per thread, several depth
    blocks per line group): windows bit-exact against the oracle, opacity within tolerance."""
    from stardis_amd import ops

    atm = deep_atmosphere(130)
    nus = synth.tracing_grid(6560.0, 6564.0, step=0.01)
    lines = synth.synth_lines(nus, atm, 6000, seed=53, mix=(0.85, 0.12, 0.03))
    lo, hi = ops.line_windows(130, nus, lines["line_nus"], lines["doppler_widths"], lines["gammas"], lines["alphas"])
    for k in (0, 1, 17, 2999, 5998, 5999):
        for d in (0, 63, 64, 65, 129):
            ref_lo, ref_hi = oracle.window(nus, lines["line_nus"][k], lines["gammas"][k, d], lines["doppler_widths"][k, d], lines["alphas"][k, d])
            assert (lo[k, d], hi[k, d]) == (ref_lo, ref_hi)
    out, evals = ops.calc_alan_entries(130, nus, lines["line_nus"], lines["doppler_widths"], lines["gammas"], lines["alphas"], return_evaluations=True)
    ref, ref_evals = oracle.calc_alan_entries(130, nus, lines["line_nus"], lines["doppler_widths"], lines["gammas"], lines["alphas"], return_evals=True)
    assert evals == ref_evals and evals == int((hi - lo).clip(min=0).sum())
    assert rel_err(out, ref) < 1e-12


def test_an_empty_line_list_counts_zero_evaluations(ctx):
    """the fused step with no line at all: the evaluation count is written (0), not left as it was (found by
    scripts/fuzz_group_loopback.py: the sharded entry point returned whatever the scratch held)"""
    import ctypes as C

    from stardis_amd import synth

    atm = synth.solar_atmosphere()
    nus = synth.tracing_grid(6560.0, 6562.0, step=0.01)
    lines = {k: v[:0] for k, v in synth.synth_lines(nus, atm, 4, seed=1).items()}
    th, w = synth.thetas_and_weights(4)
    syn = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, lines, synth.synth_continuum_state(atm), ctx=ctx)
    d_ev = ctx.upload(np.array([-12345], dtype=np.int64), dtype=np.int64)
    ctx.call("sdx_synthesize_dev", syn.n_depth, syn.n_nu, syn.d_nus.ptr, 0, syn.n_nu, 0, None, None, None, 1, None, C.byref(syn.cont), syn.n_theta,
             syn.d_t.ptr, syn.d_ray.ptr, syn.d_w.ptr, None, None, syn.flux_ptr, syn.n_nu, d_ev.ptr)
    assert int(d_ev.numpy()[0]) == 0 and (syn.F_nu()[-1] > 0).all()
    syn.close()



def test_several_steps_per_graph_launch(ctx):
    """capture(batch=n): a second hipGraph that holds n consecutive steps; step_batch() enqueues it (-> n).  Every step of a batch is
    the whole step on the resident inputs — an input updated on the device between two launches is seen by the next one — and the
    results are the single-step graph's bits."""
    atm, nus, lines, cont, th, w = small_workload()
    syn = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, lines, cont, ctx=ctx, track_evaluations=False)
    syn.capture(batch=4)
    syn.step()
    one = (syn.F_nu().copy(), syn.total_alphas().copy())
    syn.d_F.zero()
    assert syn.step_batch() == 4
    assert np.array_equal(syn.F_nu(), one[0]) and np.array_equal(syn.total_alphas(), one[1])
    # the batch reads the inputs at every step: halve the line strengths in place, replay
    syn.d_a.set(0.5 * lines["alphas"])
    assert syn.step_batch() == 4
    half = syn.F_nu().copy()
    syn.step()
    assert np.array_equal(syn.F_nu(), half) and not np.array_equal(half, one[0])
    syn.close()
    assert syn.graph_batch is None and syn.step_batch() == 1  # (no graph: one eager step)


def test_f32mix_twins_of_the_host_entry_points(ctx):
    """sdx_line_opacity_f32mix / sdx_raytrace_f32mix / sdx_synthesize_f32mix: the fp64 host entry points with the
    "mixed_precision" option on for the call — the bits of the option path, the option restored afterwards, within the stated 1e-4 of fp64."""
    import ctypes as C

    from stardis_amd import _lib

    atm, nus, lines, cont, th, w = small_workload(n_lines=400, step=0.005, seed=23, n_theta=6)
    nd, n_nu, n_l = atm["temperatures"].size, nus.size, lines["line_nus"].size
    keep = []

    def host(a, dt=np.float64):
        a = np.ascontiguousarray(a, dtype=dt)
        keep.append(a)
        return a.ctypes.data

    g = np.ascontiguousarray(lines["gammas"]).reshape(n_l, -1)
    ray = np.asarray(atm["dist"]).reshape(-1, 1) / np.cos(th)
    c = _lib.Continuum()
    c.electron_density = host(cont["n_e"])
    line_args = (ctx.handle, nd, n_nu, host(nus), n_l, host(lines["line_nus"]), host(lines["doppler_widths"]), host(g), g.shape[1], host(lines["alphas"]))

    def run(fn_line, fn_ray, fn_syn):
        line, F1, total, F2 = (np.zeros((nd, n_nu)) for _ in range(4))  # (F_nu is accumulated into: zeros, like RadiationField.__init__)
        _lib.check(fn_line(*line_args, line.ctypes.data, None))
        plane = line + 1e-9
        _lib.check(fn_ray(ctx.handle, nd, n_nu, th.size, host(nus), host(atm["temperatures"]), host(ray), host(w), host(plane), F1.ctypes.data, None))
        _lib.check(fn_syn(*line_args, C.byref(c), th.size, host(atm["temperatures"]), host(ray), host(w), None, total.ctypes.data, F2.ctypes.data, None))
        return line, F1, total, F2

    lib = ctx.lib
    ref = run(lib.sdx_line_opacity_f64, lib.sdx_raytrace_f64, lib.sdx_synthesize_f64)
    twin = run(lib.sdx_line_opacity_f32mix, lib.sdx_raytrace_f32mix, lib.sdx_synthesize_f32mix)
    again = run(lib.sdx_line_opacity_f64, lib.sdx_raytrace_f64, lib.sdx_synthesize_f64)  # the option was restored: fp64 again
    ctx.set_option("mixed_precision", 1)
    try:
        option = run(lib.sdx_line_opacity_f64, lib.sdx_raytrace_f64, lib.sdx_synthesize_f64)
        inside = run(lib.sdx_line_opacity_f32mix, lib.sdx_raytrace_f32mix, lib.sdx_synthesize_f32mix)  # (on before, on after)
        still_on = run(lib.sdx_line_opacity_f64, lib.sdx_raytrace_f64, lib.sdx_synthesize_f64)
    finally:
        ctx.set_option("mixed_precision", 0)
    for a, b, o, i, s_, r in zip(twin, again, option, inside, still_on, ref):
        assert np.array_equal(b, r)                                  # restored to fp64
        assert np.array_equal(a, o) and np.array_equal(i, o) and np.array_equal(s_, o)  # the twin = the option path, which stays on
    assert not np.array_equal(twin[3], ref[3])
    assert rel_err(twin[3][1:], ref[3][1:]) < 1e-4 and rel_err(twin[1][1:], ref[1][1:]) < 1e-4  # stated tolerance on the flux
    with pytest.raises(ValueError):
        _lib.check(lib.sdx_synthesize_f32mix(None, *line_args[1:], C.byref(c), th.size, host(atm["temperatures"]), host(ray), host(w), None, None, None, None))
