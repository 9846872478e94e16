"""Shards of the 1e6-line workload (the counter-driven culled pre-pass at its real size) against the unsharded run, bit for bit:
python scripts/r4/big_shard_check.py [TAG] [WORLD] [RANK ...]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from stardis_amd import synth, parallel
from stardis_amd.engine import SpectralSynthesizer

tag = sys.argv[1] if len(sys.argv) > 1 else "S-c4m"
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
ranks = [int(a) for a in sys.argv[3:]] or [0, world // 2, world - 1]
w = synth.make_workload(tag)
atm, nus = w["atm"], w["nus"]
full = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], w["thetas"], w["weights"], w["lines"], w["cont"], track_evaluations=False, keep_line=False)
full.step()
F = full.F_nu().copy()
full.close()
shards = parallel.balanced_shards(parallel.column_cost(nus, w["lines"]), world)
for r in ranks:
    b, c = shards[r]
    s = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], w["thetas"], w["weights"], w["lines"], w["cont"], shard=(b, c), track_evaluations=False, keep_line=False)
    s.step()
    same = np.array_equal(s.F_nu(), F[:, b:b + c])
    print(f"{tag} rank {r}/{world} shard ({b}, {c}): F_nu {'identical to the unsharded run' if same else 'DIFFERS'}", flush=True)
    s.close()
    assert same
