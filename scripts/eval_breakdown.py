"""Host-side breakdown of a workload's Voigt evaluations by the class the line kernel gives them (planning aid):
narrow windows, delegated cores, wide windows split into fast tiles (256 points wholly inside the window and clear of
the core), edge tiles and core tiles.  python scripts/eval_breakdown.py S-c3"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from stardis_amd import synth

tag = sys.argv[1] if len(sys.argv) > 1 else "S-c3"
w = synth.make_workload(tag)
nus = np.asarray(w["nus"]); n = nus.size
L = w["lines"]
d_nu = -np.max(np.diff(nus))
ln = np.asarray(L["line_nus"])
centre = (n - np.searchsorted(nus[::-1], ln))[:, None]
g = np.asarray(L["gammas"]).reshape(ln.size, -1); dw = np.asarray(L["doppler_widths"]); al = np.asarray(L["alphas"])
pix = (g + dw) * al / d_nu * 20.0
hw = np.minimum(np.where(pix > 10, pix, 10), float(n)).astype(np.int64)
lo = np.clip(centre - hw, 0, n); hi = np.clip(centre + hw, 0, n)
ev = hi - lo
narrow = hw <= 64
y = g / (np.sqrt(np.pi) * np.pi) / dw
chw = np.minimum((np.maximum(15.0 - y, 0) * dw / d_nu + 2).astype(np.int64), hw)
clo = np.clip(centre - chw, 0, n); chi = np.clip(centre + chw, 0, n)
deleg = (~narrow) & (chw <= 64)
T = 256
# tiles touched by a wide window, fast tiles: wholly inside [lo,hi) and not intersecting [clo,chi)
t0 = lo // T; t1 = (hi + T - 1) // T
inside0 = (lo + T - 1) // T; inside1 = hi // T          # tiles wholly inside
n_inside = np.maximum(inside1 - inside0, 0)
c0 = clo // T; c1 = (chi + T - 1) // T                     # tiles touching the core
n_core_tiles = np.where(chi > clo, np.maximum(np.minimum(c1, inside1) - np.maximum(c0, inside0), 0), 0)
fast_tiles = np.where(narrow, 0, n_inside - n_core_tiles)
hits = np.where(narrow, 0, t1 - t0)
tot = ev.sum()
print(tag, "items", ev.size, "evaluations %.4g" % tot)
print(" narrow windows: items %.3g  evals %.4g (%.1f %%)" % (narrow.sum(), ev[narrow].sum(), 100 * ev[narrow].sum() / tot))
print(" wide windows:   items %.3g  evals %.4g" % ((~narrow).sum(), ev[~narrow].sum()))
print("   delegated cores: items %.3g  evals %.4g (%.2f %%)" % (deleg.sum(), (chi - clo)[deleg].sum(), 100 * (chi - clo)[deleg].sum() / tot))
kept = (~narrow) & ~deleg
print("   kept cores:      items %.3g  evals %.4g (%.2f %%)" % (kept.sum(), (chi - clo)[kept].sum(), 100 * (chi - clo)[kept].sum() / tot))
print("   tile hits %.4g, fast %.4g (%.1f %% of hits; %.1f %% of all evaluations)" % (hits.sum(), fast_tiles.sum(), 100 * fast_tiles.sum() / hits.sum(), 100 * fast_tiles.sum() * T / tot))
other = hits.sum() - fast_tiles.sum()
print("   non-fast hits %.4g (%.2f per wide item), evaluations in them %.4g (%.1f %%)" % (other, other / (~narrow).sum(), ev[~narrow].sum() - fast_tiles.sum() * T, 100 * (ev[~narrow].sum() - fast_tiles.sum() * T) / tot))
hwm = hw.max(axis=1)
print(" lines: huge (>4096) %d, wide %d, narrow-only %d" % ((hwm > 4096).sum(), ((hwm > 64) & (hwm <= 4096)).sum(), (hwm <= 64).sum()))
for a, b in ((64, 256), (256, 1024), (1024, 4096), (4096, 1 << 30)):
    m = (hw > a) & (hw <= b)
    print("   hw in (%d,%d]: items %.3g evals %.4g (%.1f %%)" % (a, b, m.sum(), ev[m].sum(), 100 * ev[m].sum() / tot))
