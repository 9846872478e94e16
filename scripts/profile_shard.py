"""Run a few graph-replayed steps of ONE frequency shard of a workload (for rocprofv3):
python scripts/profile_shard.py [workload] [world] [rank] [steps]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stardis_amd import synth, parallel
from stardis_amd.engine import SpectralSynthesizer

tag = sys.argv[1] if len(sys.argv) > 1 else "S-c3"
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8
rank = int(sys.argv[3]) if len(sys.argv) > 3 else world - 1
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 20
w = synth.make_workload(tag)
atm = w["atm"]
shard = parallel.balanced_shards(parallel.column_cost(w["nus"], w["lines"]), world)[rank]
syn = SpectralSynthesizer(w["nus"], atm["temperatures"], atm["dist"], w["thetas"], w["weights"], w["lines"], w["cont"], shard=shard,
                          track_evaluations=False, keep_line=False)
syn.capture()
for _ in range(steps):
    syn.step()
syn.synchronize()
print(tag, "shard", shard, "F[-1][:3]", syn.F_nu()[-1][:3])
