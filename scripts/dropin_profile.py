"""GPU experiment: where the steady-state time of the fused drop-in call (create_stellar_radiation_field on host objects) goes.
python scripts/dropin_profile.py [TAG]      (STARDIS_AMD_FUSED=0: the source-by-source path instead)"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import stardis_amd.radiation_field.base as rf
from stardis_amd import synth

tag = sys.argv[1] if len(sys.argv) > 1 else "S-c2"
cfg = synth.WORKLOADS[tag]
atm = synth.solar_atmosphere()
nus = synth.tracing_grid(cfg["lam0"], cfg["lam1"], cfg.get("R"), cfg.get("step"))
plasma, model, config, arrays = synth.fake_plasma(nus, atm, 2000, synth.SEED)
for _ in range(5):
    rf.create_stellar_radiation_field(nus.copy(), model, plasma, config)
ts = []
for _ in range(50):
    t0 = time.perf_counter()
    rf.create_stellar_radiation_field(nus.copy(), model, plasma, config)
    ts.append(time.perf_counter() - t0)
print(f"{tag}: steady min {min(ts) * 1e3:.3f} ms median {sorted(ts)[len(ts) // 2] * 1e3:.3f} ms")
pr = cProfile.Profile()
pr.enable()
for _ in range(200):
    rf.create_stellar_radiation_field(nus.copy(), model, plasma, config)
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("tottime").print_stats(28)
