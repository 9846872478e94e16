"""GPU experiment: where the far field starts to pay — S-c3's line density on grids of 8k .. 60k points, far_field 0 against 1.
python scripts/r5/far_threshold_probe.py"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from stardis_amd import synth, _lib
from stardis_amd.engine import SpectralSynthesizer

atm = synth.solar_atmosphere()
cont = synth.synth_continuum_state(atm)
th, w = synth.thetas_and_weights(synth.N_THETAS)
for lam1 in (4350.0, 4750.0, 5200.0, 5600.0, 6500.0, 7300.0):
    nus = synth.tracing_grid(4000.0, lam1, R=1.0e5)
    n_lines = int(1.25 * nus.size)
    lines = synth.synth_lines(nus, atm, n_lines, seed=5)
    res = []
    for far in (0, 1):
        ctx = _lib.default_context()
        ctx.set_option("far_field", far)
        syn = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, lines, cont, track_evaluations=False, keep_line=False)
        syn.enqueue(); syn.synchronize()
        syn.capture()
        t_end = time.perf_counter() + 0.3
        while time.perf_counter() < t_end:
            for _ in range(20): syn.step()
            syn.synchronize()
        best = 1e9
        for _ in range(5):
            t0 = time.perf_counter()
            for _ in range(50): syn.step()
            syn.synchronize()
            best = min(best, (time.perf_counter() - t0) / 50)
        res.append(best)
        syn.close()
        ctx.set_option("far_field", -1)
    print(f"{nus.size:6d} points, {n_lines:6d} lines: direct {res[0] * 1e6:8.1f} us   far field {res[1] * 1e6:8.1f} us   ratio {res[0] / res[1]:.2f}", flush=True)
