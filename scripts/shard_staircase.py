"""GPU experiment: line-kernel time of a shard (0, n) of S-c3 against its width n — the staircase of wave generations.
python scripts/shard_staircase.py BEGIN TILES...   (SDX_SPLIT_LAUNCHES=1 times the two roles apart; SDX_TAG: another workload)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stardis_amd import synth, _lib
from stardis_amd.engine import SpectralSynthesizer

w = synth.make_workload(os.environ.get("SDX_TAG", "S-c3"))
atm, nus = w["atm"], w["nus"]
begin = int(sys.argv[1])
for tiles in [int(a) for a in sys.argv[2:]]:
    shard = (begin, min(256 * tiles, nus.size - begin))
    syn = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], w["thetas"], w["weights"], w["lines"], w["cont"], shard=shard,
                              track_evaluations=False, keep_line=False)
    ctx = syn.ctx
    syn.capture()
    for _ in range(20): syn.step()
    syn.synchronize()
    ctx.call("sdx_profile_enable", 1); ctx.call("sdx_profile_reset")
    for _ in range(3): syn.enqueue()
    ctx.synchronize()
    kern = {}
    for k in ("k_line_all", "k_line_wide", "k_line_narrow", "k_raytrace", "k_prepass_continuum", "k_classify"):
        cnt, ms = C.c_int64(), C.c_double()
        _lib.check(ctx.lib.sdx_profile_get(ctx.handle, k.encode(), C.byref(cnt), C.byref(ms)))
        if cnt.value: kern[k] = round(ms.value / 3 * 1e3, 1)
    ctx.call("sdx_profile_enable", 0)
    syn.close()
    print(f"shard {shard} tiles {tiles} wide waves {tiles * 56 * 4} ({tiles * 56 * 4 / 7168:.2f} generations of 7168): {kern}", flush=True)
