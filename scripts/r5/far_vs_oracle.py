"""GPU + CPU check: the far field against the ORACLE (the reference's point-by-point sum in C) on many columns of the full-size
workloads — every line of the list present; far_field 1 (as shipped) and 0 beside each other.
python scripts/r5/far_vs_oracle.py [TAG ...] [--stride=N]   (oracle/ is test infrastructure: this script is a checker)"""
import os, sys, time
import numpy as np
root = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, root)
import oracle
from stardis_amd import synth, _lib
from stardis_amd.engine import SpectralSynthesizer

stride = next((int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--stride=")), 53)
for tag in [a for a in sys.argv[1:] if not a.startswith("--")] or ["S-c3"]:
    w = synth.make_workload(tag)
    atm, nus, lines = w["atm"], w["nus"], w["lines"]
    cols = np.arange(0, nus.size, stride)
    t0 = time.perf_counter()
    ref, evals = oracle.calc_alan_entries_columns(cols, atm["temperatures"].size, nus, lines["line_nus"], lines["doppler_widths"], lines["gammas"],
                                                  lines["alphas"], return_evals=True)
    t_cpu = time.perf_counter() - t0
    for far in (1, 0):
        ctx = _lib.default_context()
        ctx.set_option("far_field", far)
        syn = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], w["thetas"], w["weights"], lines, w["cont"], track_evaluations=False)
        syn.step()
        line = syn.alpha_line()[:, cols]
        m = ref != 0
        dev = np.abs(line[m] - ref[m]) / ref[m]
        print(f"{tag} far_field={far}: {cols.size} columns x {ref.shape[0]} depths, {evals:.3g} oracle evaluations ({t_cpu:.0f} s): "
              f"zeros agree {bool(np.array_equal(line == 0, ref == 0))}, relative deviation of the line opacity max {dev.max():.2e}, "
              f"99.9th percentile {np.quantile(dev, 0.999):.2e}, median {np.median(dev):.2e}", flush=True)
        syn.close()
        ctx.set_option("far_field", -1)
