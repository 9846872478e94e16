"""GPU experiment: the C library's one-process sharded entry point (sdx_synthesize_sharded_f64) on random long- and short-list
configurations with RANDOM shard boundaries — unequal, empty shards at either end or in the middle — 1 to 6 loop-back ranks on one
device (SDX_EXPERIMENT=1 SDX_GROUP_LOOPBACK=1: device copies instead of RCCL), optional planes and evaluation count: the assembled
result against the single-GPU run bit for bit.  SDX_EXPERIMENT=1 SDX_GROUP_LOOPBACK=1 python scripts/fuzz_group_loopback.py FIRST LAST"""
import os, sys, traceback
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, root)
sys.path.insert(0, os.path.join(root, "tests"))
import numpy as np
import test_gpu_long_random as TL
import test_gpu_random as TS
from test_gpu_group import single_gpu
from stardis_amd import _lib
from stardis_amd.group import DeviceGroup

assert os.environ.get("SDX_GROUP_LOOPBACK") == "1" and os.environ.get("SDX_EXPERIMENT") == "1"
ctx = _lib.default_context()
bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    try:
        rng = np.random.default_rng(99000 + seed)
        long_list = bool(rng.random() < 0.6)
        atm, nus, lines, cont, th, w = TL.long_case(seed) if long_list else TS.random_case(seed)
        if th.size > 64 or nus.size < 1:
            continue
        ref = single_gpu(ctx, atm, nus, lines, cont, th, w)
        ranks = int(rng.integers(1, 7))
        cuts = np.sort(rng.integers(0, nus.size + 1, size=ranks - 1)) if ranks > 1 else np.array([], dtype=np.int64)
        if ranks > 2 and rng.random() < 0.3:
            cuts[1] = cuts[0]  # an empty shard in the middle
        if ranks > 1 and rng.random() < 0.2:
            cuts[0] = 0  # an empty first shard (the evaluation count then comes from the first non-empty one)
        edges = np.concatenate([[0], cuts, [nus.size]])
        shards = [(int(edges[r]), int(edges[r + 1] - edges[r])) for r in range(ranks)]
        planes, evals = bool(rng.random() < 0.6), bool(rng.random() < 0.4)
        grp = DeviceGroup(devices=[0] * ranks)
        try:
            out = grp.synthesize(nus, atm["temperatures"], atm["dist"], th, w, lines, cont, shards=shards, want_planes=planes, want_evaluations=evals)
        finally:
            grp.close()
        assert np.array_equal(out["F_nu"], ref["F_nu"]) and np.array_equal(out["emergent_flux"], ref["F_nu"][-1]), "flux"
        if planes:
            assert np.array_equal(out["alpha_line"], ref["alpha_line"]) and np.array_equal(out["total_alphas"], ref["total_alphas"]), "planes"
        if evals:
            assert out["evaluations"] == ref["evaluations"], ("evaluations", out["evaluations"], ref["evaluations"])
        print(f"seed {seed}: ok  {'long' if long_list else 'short'} list, {nus.size} points, {lines['line_nus'].size} lines, shards {shards}{', planes' if planes else ''}{', evaluations' if evals else ''}", flush=True)
    except Exception:
        bad += 1
        print(f"seed {seed}: FAILED", flush=True)
        traceback.print_exc()
print("failures:", bad)
