"""Per-kernel times of ONE rank of a balanced N-way split (graph-replayed step + eager event pass), every stage name the library
records: python scripts/r6/shard_kernels.py TAG WORLD RANK [--shard BEGIN COUNT] [--opt NAME=VALUE ...]
Experiment knobs come from the environment (SDX_EXPERIMENT=1 SDX_...); context options from --opt."""
import ctypes as C, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from stardis_amd import synth, parallel, _lib
from stardis_amd.engine import SpectralSynthesizer

tag, world, rank = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
w = synth.make_workload(tag)
atm, nus = w["atm"], w["nus"]
ctx = _lib.default_context()
for i, a in enumerate(sys.argv):
    if a == "--opt":
        k, v = sys.argv[i + 1].split("=")
        ctx.set_option(k, int(v))
shard = parallel.balanced_shards(parallel.column_cost(nus, w["lines"], ctx=ctx), world)[rank] if world > 1 else (0, nus.size)
if "--shard" in sys.argv:
    k = sys.argv.index("--shard")
    shard = (int(sys.argv[k + 1]), int(sys.argv[k + 2]))
syn = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], w["thetas"], w["weights"], w["lines"], w["cont"], shard=shard, track_evaluations=False,
                          keep_line=False, ctx=ctx)
syn.capture()
t_end = time.perf_counter() + 0.3
while time.perf_counter() < t_end:
    for _ in range(10): syn.step()
    syn.synchronize()
best = 1e9
for _ in range(5):
    t0 = time.perf_counter()
    for _ in range(20): syn.step()
    syn.synchronize()
    best = min(best, (time.perf_counter() - t0) / 20)
ctx.call("sdx_profile_enable", 1); ctx.call("sdx_profile_reset")
for _ in range(5): syn.enqueue()
ctx.synchronize()
kern = {}
for k in ("k_dnu_partial", "k_classify", "k_prepass_continuum", "k_line_prepass", "k_hlist", "k_far_ranges", "k_line_all", "k_line_wide", "k_line_narrow",
          "k_line_far", "k_raytrace"):
    cnt, ms = C.c_int64(), C.c_double()
    _lib.check(ctx.lib.sdx_profile_get(ctx.handle, k.encode(), C.byref(cnt), C.byref(ms)))
    if cnt.value: kern[k] = round(ms.value / 5 * 1e3, 1)
print(f"{tag} rank {rank}/{world} shard {shard}: step {best * 1e6:.1f} us  sum {sum(kern.values()):.1f}  {kern}  env {dict((k, v) for k, v in os.environ.items() if k.startswith('SDX_') and k != 'SDX_EXPERIMENT')}", flush=True)
