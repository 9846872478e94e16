"""GPU experiment: the seeded random configurations of tests/test_gpu_random.py beyond the 24 the suite runs, fp64 and (opacity
and flux against fp64 within the mode's tolerance) fp32-mixed.  python scripts/fuzz_random_cases.py FIRST LAST"""
import os, sys, traceback
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import test_gpu_random as T
from conftest import rel_err
from stardis_amd._lib import default_context
from stardis_amd.engine import SpectralSynthesizer

ctx = default_context()
first, last = int(sys.argv[1]), int(sys.argv[2])
bad = 0
for seed in range(first, last):
    try:
        T.test_random_configuration(ctx, seed)
        atm, nus, lines, cont, th, w = T.random_case(seed)
        ref = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, lines, cont, ctx=ctx)
        ref.step()
        a64, F64 = ref.alpha_line(), ref.F_nu()
        ctx.set_option("mixed_precision", 1)
        try:
            syn = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, lines, cont, ctx=ctx)
            syn.step()
            a32, F32 = syn.alpha_line(), syn.F_nu()
        finally:
            ctx.set_option("mixed_precision", 0)
        ea = float(np.max(np.abs(a32 - a64) / np.maximum(np.abs(a64).max(), 1e-300)))
        scale = np.maximum(np.abs(F64).max(axis=0, keepdims=True), 1e-300)
        ef = float(np.max(np.abs(F32 - F64) / scale))
        ok = ea < 1e-4 and ef < 1e-4
        print(f"seed {seed}: ok fp64; mixed opacity {ea:.1e} flux {ef:.1e} {'ok' if ok else 'MIXED OUT OF TOLERANCE'}", flush=True)
        bad += not ok
    except Exception:
        bad += 1
        print(f"seed {seed}: FAILED", flush=True)
        traceback.print_exc()
print("failures:", bad)
