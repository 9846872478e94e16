"""GPU experiment: the step one rank runs in bench.py's weak-scaling set-up (global grid = WORLD x S-c2, this rank's
shard only, no gather), to check that the per-rank step time does not grow with the world size.
python scripts/shard_probe.py [WORLD ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
from stardis_amd import _lib
from stardis_amd.engine import SpectralSynthesizer, shard_bounds

for world in [int(a) for a in sys.argv[1:]] or [1, 2, 4, 8]:
    w = bench.build_workload("S-c2", world)
    nus, atm = w["nus"], w["atm"]
    for rank in sorted({0, world - 1}):
        begin, count = shard_bounds(nus.size, world, rank)
        syn = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], w["thetas"], w["weights"], w["lines"], w["cont"],
                                  shard=(begin, count))
        syn.step(); syn.synchronize()
        ev = syn.evaluations()
        syn.count_evaluations = False
        syn.capture()
        for _ in range(20): syn.step()
        syn.synchronize()
        t0 = time.perf_counter()
        for _ in range(200): syn.step()
        syn.synchronize()
        print(f"world {world} rank {rank}: {count} of {nus.size} columns, {ev} evaluations, {(time.perf_counter() - t0) / 200 * 1e6:.1f} us/step", flush=True)
        syn.close()
