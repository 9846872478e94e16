"""Run a few graph-replayed steps of ONE frequency shard of a workload (for rocprofv3):
python scripts/profile_shard.py [workload] [world] [rank] [steps] [--eager]   (--eager: plain launches, for counter passes)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stardis_amd import synth, parallel
from stardis_amd.engine import SpectralSynthesizer

ARGS = [a for a in sys.argv[1:] if not a.startswith("--")]
tag = ARGS[0] if len(ARGS) > 0 else "S-c3"
world = int(ARGS[1]) if len(ARGS) > 1 else 8
rank = int(ARGS[2]) if len(ARGS) > 2 else world - 1
steps = int(ARGS[3]) if len(ARGS) > 3 else 20
w = synth.make_workload(tag)
atm = w["atm"]
shard = parallel.balanced_shards(parallel.column_cost(w["nus"], w["lines"]), world)[rank]
syn = SpectralSynthesizer(w["nus"], atm["temperatures"], atm["dist"], w["thetas"], w["weights"], w["lines"], w["cont"], shard=shard,
                          track_evaluations=False, keep_line=False)
if "--eager" not in sys.argv:
    syn.capture()
for _ in range(steps):
    syn.step()
syn.synchronize()
print(tag, "shard", shard, "F[-1][:3]", syn.F_nu()[-1][:3])
