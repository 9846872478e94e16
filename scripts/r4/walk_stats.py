"""What the waves of the wide role spend their time on.  Needs the analysis build of the library (csrc compiled with
-DSDX_WALK_STATS into stardis_amd/lib_stats/, scripts/r4/walk_stats.sh does that and sets STARDIS_AMD_LIB): every wide-role wave
leaves its start / end time (100 MHz counter) and its counts of scan chunks, test-free hits, general hits, region-I blocks and
full-Voigt blocks in a device array.  One eager step of the workload (or of one rank of an N-way split), then a least-squares fit
time = a + b chunks + c fast + d general + e blocks_if + f blocks_slow over the waves and the launch's time line.
python scripts/r4/walk_stats.py [TAG] [WORLD RANK]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from stardis_amd import synth, parallel, _lib
from stardis_amd.engine import SpectralSynthesizer

tag = sys.argv[1] if len(sys.argv) > 1 else "S-c2"
w = synth.make_workload(tag)
atm, nus = w["atm"], w["nus"]
shard = None
if len(sys.argv) > 3:
    shard = parallel.balanced_shards(parallel.column_cost(nus, w["lines"]), int(sys.argv[2]))[int(sys.argv[3])]
syn = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], w["thetas"], w["weights"], w["lines"], w["cont"], shard=shard, track_evaluations=False, keep_line=False)
lib = syn.ctx.lib
n_words = (1 << 19) * 8
buf = np.zeros(n_words, dtype=np.uint64)
read = lambda: lib.sdx_walk_stats_read(C.c_void_p(buf.ctypes.data), C.c_longlong(n_words))  # noqa: E731
for _ in range(3): syn.step()
syn.synchronize()
assert read() == 0  # (clears: the warm-up steps are dropped)
syn.step(); syn.synchronize()
assert read() == 0
a_all = buf.reshape(-1, 8)
nar = a_all[(a_all[:, 0] >> np.uint64(62)) == 3]
a = a_all[(a_all[:, 0] >> np.uint64(62)) == 2]
key = a[:, 0]
d = ((key >> np.uint64(40)) & np.uint64(0xFFFF)).astype(float); tile = ((key >> np.uint64(8)) & np.uint64(0xFFFFFF)).astype(float); split = (key & np.uint64(0xFF)).astype(float)
t0, t1 = a[:, 1].astype(float), a[:, 2].astype(float)
flush_us = (a[:, 3] >> np.uint64(32)).astype(float) / 100.0  # (queued walks of the far-field kernels: time spent evaluating queued hits)
a[:, 3] &= np.uint64(0xFFFFFFFF)
chunks, fast, general, b_if, b_slow = (a[:, k].astype(float) for k in range(3, 8))
dur = (t1 - t0) / 100.0
start, end = (t0 - t0.min()) / 100.0, (t1 - t0.min()) / 100.0
print(f"{tag} shard {shard}: {len(a)} wide waves; launch span {end.max():.1f} us; wave duration mean {dur.mean():.1f} max {dur.max():.1f} us; last start {start.max():.1f} us")
X = np.stack([np.ones_like(dur), chunks, fast, general, b_if, b_slow], axis=1)
coef, *_ = np.linalg.lstsq(X, dur, rcond=None)
print("fit [us]: const %.2f, per scan chunk %.3f, per test-free hit %.3f, per general hit %.3f, per region-I block %.3f, per full-Voigt block %.3f" % tuple(coef))
print("queued evaluation: mean %.1f us per wave of %.1f" % (flush_us.mean(), ((t1 - t0) / 100.0).mean()))
print("mean per wave: chunks %.1f fast %.1f general %.1f blocks_if %.1f blocks_slow %.1f" % (chunks.mean(), fast.mean(), general.mean(), b_if.mean(), b_slow.mean()))
print("share of the summed wave time: const %.0f%% chunks %.0f%% fast %.0f%% general %.0f%% if %.0f%% slow %.0f%%" % tuple(100 * coef * X.sum(axis=0) / dur.sum()))
depths = sorted(set(d.astype(int)))
for dd in depths[:: max(1, len(depths) // 8)]:
    m = d == dd
    k = np.argmax(dur * m)
    print(f"  depth {dd:3d}: mean {dur[m].mean():6.1f} us, max {dur[m].max():6.1f} (tile {int(tile[k])}, subset {int(split[k])}: chunks {int(chunks[k])} fast {int(fast[k])} general {int(general[k])} if {int(b_if[k])} slow {int(b_slow[k])}), starts {start[m].min():5.1f}..{start[m].max():5.1f}, ends ..{end[m].max():5.1f}")
late = np.argsort(-end)[:8]
print("last waves to finish (depth, tile, subset, start, end, fast, general, slow):", [(int(d[k]), int(tile[k]), int(split[k]), round(float(start[k]), 1), round(float(end[k]), 1), int(fast[k]), int(general[k]), int(b_slow[k])) for k in late])
hist, edges = np.histogram(end, bins=10)
print("waves finishing per tenth of the span:", hist.tolist())
if len(nar):  # narrow-role waves (one frequency each, F = 1 path only)
    n0, n1 = nar[:, 1].astype(float), nar[:, 2].astype(float)
    base = min(t0.min(), n0.min())
    nd = (n1 - n0) / 100.0
    ns, ne = (n0 - base) / 100.0, (n1 - base) / 100.0
    ch, rel, ev = (nar[:, k].astype(float) for k in range(3, 6))
    print(f"{len(nar)} narrow waves: first start {ns.min():.1f} us, last start {ns.max():.1f}, last end {ne.max():.1f}; duration mean {nd.mean():.1f} max {nd.max():.1f} us; per wave chunks {ch.mean():.1f} relevant lines {rel.mean():.1f} evaluated {ev.mean():.1f}")
    Xn = np.stack([np.ones_like(nd), ch, rel, ev], axis=1)
    cn, *_ = np.linalg.lstsq(Xn, nd, rcond=None)
    print("  fit [us]: const %.2f, per chunk %.3f, per relevant line %.3f, per evaluated line %.3f" % tuple(cn))
    print("  narrow waves starting per tenth of the span:", np.histogram(ns, bins=10, range=(0, max(ne.max(), end.max())))[0].tolist())
    print("  narrow waves finishing per tenth of the span:", np.histogram(ne, bins=10, range=(0, max(ne.max(), end.max())))[0].tolist())
    print("  wide (start-base) min %.1f; wide ends per tenth:" % ((t0.min() - base) / 100.0), np.histogram((t1 - base) / 100.0, bins=10, range=(0, max(ne.max(), end.max())))[0].tolist())

