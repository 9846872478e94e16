"""GPU experiment: strong scaling of one workload by frequency sharding, emulated on one GPU — each rank's step is timed
alone (ranks are independent: no data-path exchange; the flux gather overlaps the next step), the projected speed-up is
t(1) / max_r t_r(P).  Steps are replayed as hipGraphs, like bench.py; the slowest rank's per-kernel times are printed.
python scripts/strong_scaling_probe.py [TAG] [WORLD ...] [--balanced] [--all-ranks] [--verbose] [--two-collectives] [--in-flight=K]
--two-collectives: every rank classifies 1 / WORLD of the line list and reads the other shares from a buffer filled beforehand (what
the all-gather of m_max would deliver; its cost is NOT in the time printed — profiles/README.md models it)."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stardis_amd import synth, parallel, _lib
from stardis_amd.engine import SpectralSynthesizer, shard_bounds

tag = sys.argv[1] if len(sys.argv) > 1 else "S-c3"
balanced = "--balanced" in sys.argv
worlds = [int(a) for a in sys.argv[2:] if a.isdigit()] or [1, 2, 4, 8]
in_flight = next((int(a.split("=")[1]) for a in sys.argv if a.startswith("--in-flight=")), 1)
w = synth.make_workload(tag)
atm, nus, ln = w["atm"], w["nus"], w["lines"]
# per-column cost: window evaluations + the column's share of the formal solution and continuum (~8000 evaluation-equivalents)
work = parallel.column_cost(nus, ln, **{k: float(os.environ[e]) for k, e in (("scan_weight", "SDX_SCAN_WEIGHT"), ("core_weight", "SDX_CORE_WEIGHT"), ("fixed", "SDX_FIXED"), ("far_weight", "SDX_FAR_WEIGHT"), ("huge_weight", "SDX_HUGE_WEIGHT")) if e in os.environ}) if balanced else None
KERNELS = ("k_dnu_partial", "k_classify", "k_prepass_continuum", "k_line_prepass", "k_hlist", "k_gather", "k_far_ranges", "k_line_all", "k_line_wide", "k_line_narrow", "k_line_far", "k_raytrace")


def rank_time(world, rank, reps=20):
    shard = parallel.balanced_shards(work, world)[rank] if balanced else shard_bounds(nus.size, world, rank)
    two = "--two-collectives" in sys.argv and world > 1
    extra = {}
    if two:
        n_l = int(ln["line_nus"].size)
        per = -(-n_l // world)
        m_full = _lib.default_context().empty((n_l,))
        extra = dict(classify_share=(0, n_l), m_max=m_full)
    syn = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], w["thetas"], w["weights"], w["lines"], w["cont"], shard=shard,
                              track_evaluations=False, keep_line=False, **extra)
    ctx = syn.ctx
    # --in-flight=2: a second synthesis of the same shard on a context (stream, scratch) of its own, the two stepped alternately — what a
    # rank does with a queue of independent syntheses (a grid of models): the stream-bound and launch-bound parts of one step run beside
    # the arithmetic of the other.  The time printed is then per step of the PAIR's throughput.
    twin = None
    twins = []
    for _ in range(in_flight - 1):
        ctx_b = _lib.Context(ctx.device if hasattr(ctx, "device") else 0)
        extra_b = dict(classify_share=(0, n_l), m_max=ctx_b.empty((n_l,))) if two else {}
        twin = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], w["thetas"], w["weights"], w["lines"], w["cont"], shard=shard,
                                   track_evaluations=False, keep_line=False, ctx=ctx_b, **extra_b)
        if two:
            twin.enqueue_classify(); ctx_b.synchronize()
            twin.classify_share = (min(rank * per, n_l), max(0, min(per, n_l - rank * per)))
        twin.capture()
        twins.append(twin)
    if two:  # every share once (the other ranks' part of the gathered array), then this rank's own from now on
        syn.enqueue_classify(); ctx.synchronize()
        syn.classify_share = (min(rank * per, n_l), max(0, min(per, n_l - rank * per)))
    syn.capture()

    flip = [0]

    def one():
        if twin is not None:
            flip[0] = (flip[0] + 1) % in_flight
            which = twins[flip[0] - 1] if flip[0] else syn
            if two: which.step_classify()
            which.step()
            return
        if two: syn.step_classify()
        syn.step()

    def sync():
        syn.synchronize()
        for t_ in twins: t_.synchronize()

    # steady state, like bench.py's timed loop: ~0.2 s of untimed replays (clocks settle), then the best of five blocks of replays
    one(); sync()
    t_end = time.perf_counter() + 0.2
    while time.perf_counter() < t_end:
        for _ in range(reps): one()
        sync()
    t = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(reps): one()
        sync()
        t = min(t, (time.perf_counter() - t0) / reps)
    for t_ in twins:
        t_.close(); t_.ctx.close()
    ctx.call("sdx_profile_enable", 1); ctx.call("sdx_profile_reset")
    for _ in range(3):
        if two: syn.enqueue_classify()
        syn.enqueue()
    ctx.synchronize()
    kern = {}
    for k in KERNELS:
        cnt, ms = C.c_int64(), C.c_double()
        _lib.check(ctx.lib.sdx_profile_get(ctx.handle, k.encode(), C.byref(cnt), C.byref(ms)))
        if cnt.value: kern[k] = round(ms.value / 3 * 1e3, 1)
    ctx.call("sdx_profile_enable", 0); ctx.call("sdx_profile_reset")
    syn.close()
    return t, kern, shard


t1 = None
for world in worlds:
    ranks = range(world) if (balanced or "--all-ranks" in sys.argv) else sorted({0, world // 2, world - 1})
    res = [rank_time(world, r) for r in ranks]
    slow = max(res, key=lambda x: x[0])
    t1 = t1 or slow[0]
    print(f"{tag} world {world}: slowest rank {slow[0] * 1e3:.3f} ms (mean {sum(r[0] for r in res) / len(res) * 1e3:.3f}) -> projected speed-up {t1 / slow[0]:.2f}x; "
          f"slowest rank's shard {slow[2]} kernels [us] {slow[1]}", flush=True)
    if "--verbose" in sys.argv:
        for r, (t, kern, shard) in zip(ranks, res):
            print(f"    rank {r}: {t * 1e3:.3f} ms shard {shard} {kern}")
        print("    sum over ranks [us]:", {k: round(sum(x[1].get(k, 0.0) for x in res), 1) for k in KERNELS})
