"""Round 5: the optional SECOND collective of strong-scaled long lists (every rank classifies a share of the line list, the shares
are exchanged: include/stardis_hip.h, sdx_synthesize_classify_dev), and the culled pre-pass whose blocks with work come first
in the grid.  Both are pure scheduling: the bits of the one-call step."""
import numpy as np
import pytest

from stardis_amd import parallel, synth
from stardis_amd.engine import SpectralSynthesizer, shard_bounds

pytestmark = pytest.mark.gpu


def long_list_case(seed=3, n_nu=30000, n_lines=20000):
    atm = synth.solar_atmosphere()
    nus = synth.tracing_grid(5000.0, 5000.0 * (1.0 + 1.02 * n_nu / 1.0e5), R=1.0e5)[:n_nu]
    lines = synth.synth_lines(nus, atm, n_lines, seed=seed, mix=(0.85, 0.12, 0.03))
    th, w = synth.thetas_and_weights(8)
    return atm, nus, lines, synth.synth_continuum_state(atm), th, w


@pytest.fixture(params=[-1, 1], ids=["direct", "far_field"])
def far(request, ctx):
    """the 30 000-point grid of these cases is below the far field's automatic threshold: once as it comes, once forced on (the
    tiles' far ranges are then computed by the classification launch of phase 1 and used by phase 2)"""
    ctx.set_option("far_field", request.param)
    yield request.param
    ctx.set_option("far_field", -1)


@pytest.mark.parametrize("world", [3, 8])
def test_two_collective_mode_reproduces_the_one_call_step(ctx, world, far):
    """Each rank classifies 1 / world of the lines, the shares are 'gathered' (here: every share written into one array), the rest
    of the step runs on the gathered array: F_nu, total and line opacity of every shard equal the one-call shard and the unsharded
    run bit for bit — eagerly and as two captured graphs, shares written in place or through a send buffer."""
    atm, nus, lines, cont, th, w = long_list_case()
    args = (nus, atm["temperatures"], atm["dist"], th, w, lines, cont)
    whole = SpectralSynthesizer(*args, ctx=ctx, track_evaluations=False)
    whole.step()
    F, total, line = whole.F_nu(), whole.total_alphas(), whole.alpha_line()
    whole.close()
    n_l = lines["line_nus"].size
    per = -(-n_l // world)
    shares = [(min(r * per, n_l), max(0, min(per, n_l - r * per))) for r in range(world)]
    shards = parallel.balanced_shards(parallel.column_cost(nus, lines), world)
    for rank in (0, world // 2, world - 1):
        b, c = shards[rank]
        m_full = ctx.zeros((per * world,))
        send = ctx.zeros((per,))
        syn = SpectralSynthesizer(*args, ctx=ctx, shard=(b, c), track_evaluations=False, classify_share=shares[rank], m_max=m_full)
        # the "gather": every share, computed on this rank's shard (what a line's largest (gamma + dw) alpha is does not depend on the shard)
        for q in range(world):
            if q != rank:
                syn.classify_share = shares[q]
                syn.enqueue_classify()
        syn.classify_share = shares[rank]
        syn.enqueue_classify()
        syn.enqueue()
        assert np.array_equal(syn.F_nu(), F[:, b:b + c]), (world, rank)
        assert np.array_equal(syn.total_alphas(), total[:, b:b + c]) and np.array_equal(syn.alpha_line(), line[:, b:b + c])
        m_host = m_full.numpy()[:n_l]
        dw, g, al = lines["doppler_widths"], lines["gammas"].reshape(n_l, -1), lines["alphas"]
        assert np.array_equal(m_host, np.max((g + dw) * al, axis=1))  # the reference's operations (:561-575), one rounding each
        # through a send buffer (the all-gather's input): the share lands at its index 0
        syn.m_share_out = send
        syn.enqueue_classify()
        lo, n = shares[rank]
        assert np.array_equal(send.numpy()[:n], m_host[lo:lo + n])
        syn.m_share_out = None
        # two graphs, replayed
        syn.capture()
        for _ in range(3):
            syn.step_classify()
            syn.step()
        assert np.array_equal(syn.F_nu(), F[:, b:b + c])
        syn.close()


def test_two_collective_mode_refuses_what_it_does_not_cover(ctx):
    atm, nus, lines, cont, th, w = long_list_case(n_nu=20000, n_lines=9000)
    args = (nus, atm["temperatures"], atm["dist"], th, w, lines, cont)
    n_l = lines["line_nus"].size
    m = ctx.zeros((n_l,))
    b, c = shard_bounds(nus.size, 4, 1)
    syn = SpectralSynthesizer(*args, ctx=ctx, shard=(b, c), track_evaluations=False, classify_share=(0, n_l), m_max=m)
    with pytest.raises(ValueError, match="sdx_synthesize_classify_dev"):
        syn.enqueue()  # phase 2 without phase 1
    syn.enqueue_classify()
    syn.enqueue()
    other = SpectralSynthesizer(*args, ctx=ctx, shard=shard_bounds(nus.size, 4, 2), track_evaluations=False, classify_share=(0, n_l), m_max=m)
    with pytest.raises(ValueError, match="same grid, shard"):
        other.enqueue()  # phase 1 ran for another shard
    unsharded = SpectralSynthesizer(*args, ctx=ctx, track_evaluations=False, classify_share=(0, n_l), m_max=m)
    with pytest.raises(ValueError, match="two-collective mode is for frequency shards"):
        unsharded.enqueue_classify()
    short = synth.synth_lines(nus, atm, 500, seed=1)
    few = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, short, cont, ctx=ctx, shard=(b, c), track_evaluations=False,
                              classify_share=(0, 500), m_max=ctx.zeros((500,)))
    with pytest.raises(ValueError, match="two-collective mode is for frequency shards"):
        few.enqueue_classify()
    with pytest.raises(ValueError):
        SpectralSynthesizer(*args, ctx=ctx, shard=(b, c), classify_share=(0, n_l))  # share without the array
    for s in (syn, other, unsharded, few):
        s.close()


def test_phase_two_needs_its_own_phase_one(ctx):
    """What phase 1 leaves in the context's scratch (grid spacing, line ranges, continuum plane, far ranges) belongs to ONE phase 2:
    a second phase 2 without a new phase 1, or a phase 2 after any other step on the context has rewritten the scratch, is refused
    (SDX_ERR_ARG) instead of reading another problem's data."""
    atm, nus, lines, cont, th, w = long_list_case(n_nu=20000, n_lines=9000)
    args = (nus, atm["temperatures"], atm["dist"], th, w, lines, cont)
    n_l = lines["line_nus"].size
    m = ctx.zeros((n_l,))
    b, c = shard_bounds(nus.size, 4, 1)
    a = SpectralSynthesizer(*args, ctx=ctx, shard=(b, c), track_evaluations=False, classify_share=(0, n_l), m_max=m)
    a.enqueue_classify()
    a.enqueue()
    F = a.F_nu()
    with pytest.raises(ValueError, match="sdx_synthesize_classify_dev"):
        a.enqueue()  # consumed
    # classify(A), a one-call step of another problem B on the same context, phase 2 (A)
    other_lines = synth.synth_lines(nus, atm, 9000, seed=11, mix=(0.85, 0.12, 0.03))
    bb = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, other_lines, cont, ctx=ctx, shard=shard_bounds(nus.size, 4, 3), track_evaluations=False)
    a.enqueue_classify()
    bb.enqueue()
    with pytest.raises(ValueError, match="sdx_synthesize_classify_dev"):
        a.enqueue()
    # ... and a separate line-opacity call
    a.enqueue_classify()
    bb.enqueue_unfused()
    with pytest.raises(ValueError, match="sdx_synthesize_classify_dev"):
        a.enqueue()
    a.enqueue_classify()
    a.enqueue()
    assert np.array_equal(a.F_nu(), F)
    # buffers are checked where they are handed over
    with pytest.raises(ValueError, match="m_max holds"):
        SpectralSynthesizer(*args, ctx=ctx, shard=(b, c), track_evaluations=False, classify_share=(0, n_l), m_max=ctx.zeros((n_l - 1,)))
    with pytest.raises(ValueError, match="m_share_out holds"):
        SpectralSynthesizer(*args, ctx=ctx, shard=(b, c), track_evaluations=False, classify_share=(0, n_l), m_max=m, m_share_out=ctx.zeros((10,)))
    with pytest.raises(TypeError, match="float64"):
        SpectralSynthesizer(*args, ctx=ctx, shard=(b, c), track_evaluations=False, classify_share=(0, n_l), m_max=ctx.zeros((n_l,), np.float32))
    for s in (a, bb):
        s.close()


def test_far_field_option_may_change_between_the_phases(ctx):
    """Phase 1 with the far field off, phase 2 with it on (and the reverse): phase 2 computes the tiles' far ranges itself when
    phase 1 has not, and ignores them when the far field is off — the bits of the one-call step in that mode."""
    atm, nus, lines, cont, th, w = long_list_case(n_nu=20000, n_lines=9000)
    args = (nus, atm["temperatures"], atm["dist"], th, w, lines, cont)
    n_l = lines["line_nus"].size
    b, c = shard_bounds(nus.size, 4, 2)
    want = {}
    for mode in (0, 1):
        ctx.set_option("far_field", mode)
        one = SpectralSynthesizer(*args, ctx=ctx, shard=(b, c), track_evaluations=False)
        one.step()
        want[mode] = one.F_nu()
        one.close()
    try:
        for first, second in ((1, 0), (0, 1), (1, 0)):  # (the third pair: far_req_done left set by an earlier far-field phase 1)
            syn = SpectralSynthesizer(*args, ctx=ctx, shard=(b, c), track_evaluations=False, classify_share=(0, n_l), m_max=ctx.zeros((n_l,)))
            ctx.set_option("far_field", first)
            syn.enqueue_classify()
            ctx.set_option("far_field", second)
            syn.enqueue()
            assert np.array_equal(syn.F_nu(), want[second]), (first, second)
            syn.close()
    finally:
        ctx.set_option("far_field", -1)


@pytest.mark.parametrize("gamma_per_depth", [True, False])
def test_narrow_role_from_the_callers_tables_gives_the_same_bits(ctx, gamma_per_depth):
    """Option "narrow_records": the narrow role takes 1 / dw, y and the amplitude from records the pre-pass writes (1) or forms them
    itself from the caller's doppler widths, gammas and alphas (0) — the same three operations, so line opacity, total and flux are
    bit-identical: whole grid, a frequency shard, gammas (N_l, N_d) and (N_l, 1), the far field forced on."""
    atm, nus, lines, cont, th, w = long_list_case(seed=5)
    if not gamma_per_depth:
        lines = dict(lines, gammas=np.ascontiguousarray(lines["gammas"][:, :1]))
    args = (nus, atm["temperatures"], atm["dist"], th, w, lines, cont)
    out = {}
    try:
        for mode in (1, 0, -1):
            ctx.set_option("narrow_records", mode)
            for far in (-1, 1):
                ctx.set_option("far_field", far)
                for shard in (None, shard_bounds(nus.size, 4, 1)):
                    syn = SpectralSynthesizer(*args, ctx=ctx, shard=shard, track_evaluations=False)
                    syn.step()
                    out[(mode, far, shard)] = (syn.F_nu(), syn.total_alphas(), syn.alpha_line())
                    syn.close()
    finally:
        ctx.set_option("narrow_records", -1)
        ctx.set_option("far_field", -1)
    for (mode, far, shard), got in out.items():
        for a, b in zip(got, out[(1, far, shard)]):
            assert np.array_equal(a, b), (mode, far, shard)
