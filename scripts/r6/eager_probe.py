import sys, time, os
sys.path.insert(0, os.getcwd())
from stardis_amd import synth
from stardis_amd.engine import SpectralSynthesizer
w = synth.make_workload("S-c2"); atm = w["atm"]
syn = SpectralSynthesizer(w["nus"], atm["temperatures"], atm["dist"], w["thetas"], w["weights"], w["lines"], w["cont"], track_evaluations=False, keep_line=False)
for _ in range(2000): syn.enqueue()
syn.synchronize()
best = 1e9
for _ in range(5):
    t0 = time.perf_counter()
    for _ in range(400): syn.enqueue()
    t_host = time.perf_counter() - t0
    syn.synchronize()
    best = min(best, (time.perf_counter() - t0) / 400)
print(f"eager: {best*1e6:.1f} us per step; host enqueue {t_host/400*1e6:.1f} us per step")
syn.capture()
for _ in range(2000): syn.step()
syn.synchronize()
best = 1e9
for _ in range(5):
    t0 = time.perf_counter()
    for _ in range(400): syn.step()
    t_host = time.perf_counter() - t0
    syn.synchronize()
    best = min(best, (time.perf_counter() - t0) / 400)
print(f"graph: {best*1e6:.1f} us per step; host enqueue {t_host/400*1e6:.1f} us per step")
