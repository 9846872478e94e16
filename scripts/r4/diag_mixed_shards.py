import sys, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import test_gpu_long_random as T
from stardis_amd._lib import default_context
from stardis_amd.engine import SpectralSynthesizer, shard_bounds
ctx = default_context()
seed = int(sys.argv[1])
atm, nus, lines, cont, th, w = T.long_case(seed)
print("case", atm["temperatures"].size, nus.size, lines["line_nus"].size, th.size)
ctx.set_option("mixed_precision", 1)
syn = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, lines, cont, ctx=ctx, track_evaluations=False)
syn.step()
line, F = syn.alpha_line(), syn.F_nu()
for r in range(3):
    b, c = shard_bounds(nus.size, 3, r)
    s = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, lines, cont, ctx=ctx, shard=(b, c), track_evaluations=False)
    s.step()
    dl = s.alpha_line() != line[:, b:b + c]
    dF = s.F_nu() != F[:, b:b + c]
    cols = np.where(dl.any(axis=0))[0]
    print(f"shard {r} ({b},{c}): line differs at {dl.sum()} points in {cols.size} columns; F differs in {np.where(dF.any(axis=0))[0].size} columns")
    if cols.size:
        print("   columns (global):", (cols + b)[:20], "...", (cols + b)[-5:], " tiles(256):", np.unique((cols + b) // 256)[:20])
        d = np.where(dl[:, cols[0]])[0]
        print("   depths at first column:", d[:10], " rel diff max", np.max(np.abs(s.alpha_line()[dl] - line[:, b:b + c][dl]) / np.abs(line[:, b:b + c][dl])))
    s.close()
