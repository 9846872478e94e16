"""GPU experiment: strong scaling of one workload by frequency sharding, emulated on one GPU — each rank's step is timed
alone (ranks are independent: no data-path exchange), the projected speed-up is t(1) / max_r t_r(P).
python scripts/strong_scaling_probe.py [TAG] [WORLD ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stardis_amd import synth
from stardis_amd.engine import SpectralSynthesizer, shard_bounds

tag = sys.argv[1] if len(sys.argv) > 1 else "S-c3"
worlds = [int(a) for a in sys.argv[2:]] or [1, 2, 4, 8]
w = synth.make_workload(tag)
atm, nus = w["atm"], w["nus"]


def rank_time(world, rank, reps=5):
    syn = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], w["thetas"], w["weights"], w["lines"], w["cont"],
                              shard=shard_bounds(nus.size, world, rank), track_evaluations=False)
    syn.step(); syn.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): syn.step()
    syn.synchronize()
    return (time.perf_counter() - t0) / reps


t1 = None
for world in worlds:
    ts = [rank_time(world, r) for r in sorted({0, world // 2, world - 1})]
    t1 = t1 or max(ts)
    print(f"{tag} world {world}: slowest of ranks 0/mid/last {max(ts) * 1e3:.3f} ms -> projected speed-up {t1 / max(ts):.2f}x", flush=True)
