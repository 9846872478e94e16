"""GPU experiment: random configurations of the drop-in call — create_stellar_radiation_field on the pandas stand-in for a TARDIS
plasma (the reference's end-to-end fixture G9 + the VALD atomic and molecular lists of G11) — through the fused device pass and
through the general source-by-source path: every dictionary entry, the total opacity, F_nu and (when tracked) I_nus bit for bit,
same keys in the same order.  Options drawn per seed: which continuum sources are configured, line lists dense or as per-line
scalars, molecules, VALD broadening, the broadening list, spherical geometry, tracked intensities, the number of angles, a slice
of the frequency grid, the line opacity disabled.  python scripts/fuzz_fused_dropin.py FIRST LAST"""
import os, sys, tempfile, traceback, pathlib, types
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import stardis_amd.radiation_field.base as rf

NS = types.SimpleNamespace


import test_gpu_dropin_random as TR

bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    with tempfile.TemporaryDirectory() as tmp:
        try:
            print(f"seed {seed}: ok  [{TR.random_dropin_case(seed, tmp)}]", flush=True)
        except Exception:
            bad += 1
            print(f"seed {seed}: FAILED", flush=True)
            traceback.print_exc()
rf.FUSED = True
print("failures:", bad)
