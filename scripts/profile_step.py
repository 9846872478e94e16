"""Run a few fused steps of a workload (for rocprofv3): python scripts/profile_step.py [workload] [steps] [--graph] [--mixed] [--evals] [--linelist]
The evaluation counter (a memset + a copy per step) is off unless --evals is given, like in bench.py's timed loop."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stardis_amd import synth
from stardis_amd.engine import SpectralSynthesizer

tag = sys.argv[1] if len(sys.argv) > 1 else "S-c2"
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
w = synth.make_workload(tag)
atm = w["atm"]
if "--linelist" in sys.argv:  # the line list as per-line scalars (f1): the pre-pass generates alpha, gamma and the Doppler width
    w["lines"] = synth.synth_linelist(w["nus"], atm, int(w["lines"]["line_nus"].size), synth.SEED)
syn = SpectralSynthesizer(w["nus"], atm["temperatures"], atm["dist"], w["thetas"], w["weights"], w["lines"], w["cont"],
                          track_evaluations="--evals" in sys.argv, keep_line=False)
if "--mixed" in sys.argv:
    syn.ctx.set_option("mixed_precision", 1)
if "--graph" in sys.argv:
    syn.capture()
for _ in range(steps):
    syn.step()
syn.synchronize()
print(tag, "evals", syn.evaluations(), "F[-1][:3]", syn.F_nu()[-1][:3])
