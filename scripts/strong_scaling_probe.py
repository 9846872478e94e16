"""GPU experiment: strong scaling of one workload by frequency sharding, emulated on one GPU — each rank's step is timed
alone (ranks are independent: no data-path exchange), the projected speed-up is t(1) / max_r t_r(P).
python scripts/strong_scaling_probe.py [TAG] [WORLD ...] [--balanced]   (--balanced: shards of equal estimated work)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stardis_amd import synth
from stardis_amd.engine import SpectralSynthesizer, shard_bounds

tag = sys.argv[1] if len(sys.argv) > 1 else "S-c3"
balanced = "--balanced" in sys.argv
worlds = [int(a) for a in sys.argv[2:] if a.isdigit()] or [1, 2, 4, 8]
w = synth.make_workload(tag)
atm, nus = w["atm"], w["nus"]
from stardis_amd import parallel
ln = w["lines"]
# per-column cost: window evaluations + the column's share of the formal solution and continuum (~6000 evaluation-equivalents)
work = parallel.window_work(nus, ln["line_nus"], ln["doppler_widths"], ln["gammas"], ln["alphas"]) if balanced else None


def rank_time(world, rank, reps=5):
    syn = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], w["thetas"], w["weights"], w["lines"], w["cont"],
                              shard=parallel.balanced_shards(work, world, 6000.0)[rank] if balanced else shard_bounds(nus.size, world, rank),
                              track_evaluations=False)
    syn.step(); syn.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): syn.step()
    syn.synchronize()
    return (time.perf_counter() - t0) / reps


t1 = None
for world in worlds:
    ts = [rank_time(world, r) for r in (range(world) if balanced else sorted({0, world // 2, world - 1}))]
    t1 = t1 or max(ts)
    print(f"{tag} world {world}: slowest of ranks 0/mid/last {max(ts) * 1e3:.3f} ms -> projected speed-up {t1 / max(ts):.2f}x", flush=True)
