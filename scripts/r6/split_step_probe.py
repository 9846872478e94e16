"""GPU experiment: ONE synthesis as P concurrent column blocks on P contexts (streams), every block a complete step of its columns
(its own pre-pass of the whole list, line kernel, formal solution) — does a small grid's step get shorter when its launches overlap?
python scripts/r6/split_step_probe.py [TAG] [P ...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from stardis_amd import synth, _lib
from stardis_amd.engine import SpectralSynthesizer, shard_bounds

tag = sys.argv[1] if len(sys.argv) > 1 else "S-c2"
parts = [int(a) for a in sys.argv[2:]] or [1, 2, 3, 4]
w = synth.make_workload(tag)
atm, nus = w["atm"], w["nus"]
whole = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], w["thetas"], w["weights"], w["lines"], w["cont"], track_evaluations=False, keep_line=False)
whole.step(); F = whole.F_nu(); whole.close()
for P in parts:
    ctxs = [_lib.Context(0) for _ in range(P)]
    syns = [SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], w["thetas"], w["weights"], w["lines"], w["cont"], shard=shard_bounds(nus.size, P, k),
                                track_evaluations=False, keep_line=False, ctx=ctxs[k]) for k in range(P)]
    for s in syns: s.capture()
    def step():
        for s in syns: s.step()
    def sync():
        for c in ctxs: c.synchronize()
    step(); sync()
    same = all(np.array_equal(s.F_nu(), F[:, s.begin:s.begin + s.count]) for s in syns)
    t_end = time.perf_counter() + 0.3
    while time.perf_counter() < t_end:
        for _ in range(20): step()
        sync()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(200): step()
        sync()
        best = min(best, (time.perf_counter() - t0) / 200)
    # ... and with a host synchronisation after every step (no overlap between consecutive steps)
    t0 = time.perf_counter()
    for _ in range(200):
        step(); sync()
    lat = (time.perf_counter() - t0) / 200
    print(f"{tag} as {P} concurrent column blocks: {best * 1e6:.1f} us per step back to back, {lat * 1e6:.1f} us with a sync per step; bits equal to the whole grid: {same}", flush=True)
    for s in syns: s.close()
    for c in ctxs: c.close()
