"""Phase time line of the pre-pass blocks (analysis build -DSDX_PRE_STATS): python scripts/r5/pre_stats.py [TAG] [WORLD RANK]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from stardis_amd import synth, parallel, _lib
from stardis_amd.engine import SpectralSynthesizer

tag = sys.argv[1] if len(sys.argv) > 1 else "S-c2"
w = synth.make_workload(tag)
atm, nus = w["atm"], w["nus"]
shard = None
if len(sys.argv) > 3:
    shard = parallel.balanced_shards(parallel.column_cost(nus, w["lines"]), int(sys.argv[2]))[int(sys.argv[3])]
syn = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], w["thetas"], w["weights"], w["lines"], w["cont"], shard=shard, track_evaluations=False, keep_line=False)
lib = syn.ctx.lib
lib.sdx_pre_stats_read.argtypes = [C.c_void_p, C.c_longlong]
n = (1 << 14) * 8
buf = np.zeros(n, dtype=np.uint64)
syn.capture()
for _ in range(300): syn.step()          # warm clocks
syn.synchronize()
lib.sdx_pre_stats_read(buf.ctypes.data, n)   # clear
syn.step(); syn.synchronize()
lib.sdx_pre_stats_read(buf.ctypes.data, n)
rec = buf.reshape(-1, 8)
rec = rec[rec[:, 0] != 0]
cont = rec[(rec[:, 0] & 4) != 0]
rec = rec[(rec[:, 0] & 4) == 0]
t = rec[:, 1:8].astype(np.float64) * 0.01    # us (100 MHz clock)
t0 = t[:, 0].min()
print(f"{tag} shard {shard}: {rec.shape[0]} pre-pass blocks with work; first start 0, last start {t[:, 0].max() - t0:.2f} us, last end {t[:, 6].max() - t0:.2f} us")
names = ["sample+lnu in LDS", "centres found", "grid spacing", "arithmetic + item stores issued", "barrier", "end (scan words, summaries)"]
d = np.diff(t, axis=1)
for k, nm in enumerate(names):
    print(f"   {nm:34s} mean {d[:, k].mean():6.2f} us   median {np.median(d[:, k]):6.2f}   max {d[:, k].max():6.2f}")
print(f"   block duration                     mean {(t[:, 6] - t[:, 0]).mean():6.2f} us   max {(t[:, 6] - t[:, 0]).max():6.2f}")
if cont.shape[0]:
    c0, c1 = cont[:, 1].astype(np.float64) * 0.01 - t0, cont[:, 7].astype(np.float64) * 0.01 - t0
    print(f"   {cont.shape[0]} continuum blocks: start min {c0.min():.2f} median {np.median(c0):.2f} max {c0.max():.2f} us; duration mean {(c1 - c0).mean():.2f} max {(c1 - c0).max():.2f}; last end {c1.max():.2f} us")
starts = np.sort(t[:, 0] - t0)
print("   block starts [us] percentiles 0/25/50/75/100:", np.round(np.percentile(starts, [0, 25, 50, 75, 100]), 2))
