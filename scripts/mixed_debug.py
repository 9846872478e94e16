"""GPU experiment: where the fp32-mixed line opacity differs from fp64.  python scripts/mixed_debug.py TAG [N_LINES]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stardis_amd import synth
from stardis_amd.engine import SpectralSynthesizer

tag = sys.argv[1] if len(sys.argv) > 1 else "S-c2"
n_lines = int(sys.argv[2]) if len(sys.argv) > 2 else None
w = synth.make_workload(tag, n_lines=n_lines)
atm = w["atm"]
syn = SpectralSynthesizer(w["nus"], atm["temperatures"], atm["dist"], w["thetas"], w["weights"], w["lines"], w["cont"])
syn.step(); syn.ctx.synchronize()
ref = syn.alpha_line().copy()
syn.ctx.set_option("mixed_precision", 1)
syn.step(); syn.ctx.synchronize()
mix = syn.alpha_line().copy()
rel = np.abs(mix - ref) / np.maximum(np.abs(ref), 1e-300)
print(tag, "max rel", rel.max(), "nan", np.isnan(mix).sum(), "inf", np.isinf(mix).sum(), "points > 1e-4:", int((rel > 1e-4).sum()), "of", rel.size)
bad = np.argwhere(rel > 1e-4)
for d, i in bad[:12]:
    print(f"  d={d} i={i} (tile {i // 256}, block {(i % 256) // 64}, lane {i % 64}) fp64 {ref[d, i]:.6e} mixed {mix[d, i]:.6e}")
if bad.size:
    print("  depths hit:", np.unique(bad[:, 0])[:20], " tiles hit:", np.unique(bad[:, 1] // 256)[:20], " lanes%64:", np.unique(bad[:, 1] % 64)[:70])
