import sys, time, os
sys.path.insert(0, "/root/repo" if os.path.exists("/root/repo/stardis_amd") else os.getcwd())
from stardis_amd import synth
from stardis_amd.engine import SpectralSynthesizer
tag = sys.argv[1] if len(sys.argv) > 1 else "S-c2"
w = synth.make_workload(tag); atm = w["atm"]
for batch in (1, 2, 4, 8, 16):
    syn = SpectralSynthesizer(w["nus"], atm["temperatures"], atm["dist"], w["thetas"], w["weights"], w["lines"], w["cont"], track_evaluations=False, keep_line=False)
    syn.capture(batch=batch)
    t_end = time.perf_counter() + 0.4
    while time.perf_counter() < t_end:
        for _ in range(10): syn.step_batch()
        syn.synchronize()
    best = 1e9
    for _ in range(5):
        t0 = time.perf_counter(); n = 0
        for _ in range(max(1, 400 // batch)): n += syn.step_batch()
        syn.synchronize()
        best = min(best, (time.perf_counter() - t0) / n)
    print(f"{tag} batch {batch}: {best * 1e6:.2f} us per step", flush=True)
    syn.close()
