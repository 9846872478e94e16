"""count the Humlicek region flips over every evaluation of a workload (CPU, the oracle's window walk).
python scripts/r4/region_flips.py [TAG ...]   (S-c2 takes a second, S-c3 — 4.5e9 evaluations — a few minutes on 8 cores)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import oracle
from stardis_amd import synth

for tag in sys.argv[1:] or ["S-c2"]:
    w = synth.make_workload(tag)
    ln = w["lines"]
    t0 = time.time()
    r = oracle.count_region_flips(w["atm"]["temperatures"].size, w["nus"], ln["line_nus"], ln["doppler_widths"], ln["gammas"], ln["alphas"])
    print(f"{tag}: {r}  ({time.time() - t0:.0f} s, {oracle.num_threads()} threads)", flush=True)
