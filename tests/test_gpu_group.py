"""sdx_synthesize_sharded_f64: the C library's own multi-GPU entry point — one process, one context + stream per device,
ONE RCCL all-gather of the emergent-flux shards inside the library (the reference's frequency loop is a
prange, radiation_field/radiation_field_solvers/base.py:200).  On a one-GPU box the RCCL communicator has one rank (the
collective still runs through RCCL); the sharding logic itself is exercised with the loop-back test hook (several ranks on one
device, plain device copies instead of RCCL — RCCL refuses two ranks on one GPU); with two or more GPUs the real thing runs."""
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import ROOT
from stardis_amd import _lib, parallel, synth
from stardis_amd.engine import SpectralSynthesizer
from stardis_amd.group import DeviceGroup

pytestmark = pytest.mark.gpu


def workload(seed=23, n_lines=900, n_theta=6, lam=(6540.0, 6580.0), step=0.01, mix=(0.8, 0.15, 0.05)):
    atm = synth.solar_atmosphere()
    nus = synth.tracing_grid(*lam, step=step)
    lines = synth.synth_lines(nus, atm, n_lines, seed=seed, mix=mix)
    th, w = synth.thetas_and_weights(n_theta)
    return atm, nus, lines, synth.synth_continuum_state(atm), th, w


def single_gpu(ctx, atm, nus, lines, cont, th, w):
    syn = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, lines, cont, ctx=ctx)
    syn.step()
    out = dict(F_nu=syn.F_nu(), alpha_line=syn.alpha_line(), total_alphas=syn.total_alphas(), evaluations=syn.evaluations())
    syn.close()
    return out


def test_one_rank_group_runs_rccl_and_equals_the_single_gpu_call(ctx):
    """n_gpus = 1: the degenerate group.  Bit for bit sdx_synthesize_f64 / the resident engine, and the gather went through
    RCCL (a one-rank communicator: version reported, 8 * n_nu bytes contributed)."""
    atm, nus, lines, cont, th, w = workload()
    ref = single_gpu(ctx, atm, nus, lines, cont, th, w)
    grp = DeviceGroup(1)
    out = grp.synthesize(nus, atm["temperatures"], atm["dist"], th, w, lines, cont, want_planes=True, want_evaluations=True)
    info = grp.last_gather()
    grp.close()
    assert info["backend"] == "rccl" and info["rccl_version"] > 0 and info["ranks"] == 1 and info["bytes_per_rank"] == 8 * nus.size
    assert np.array_equal(out["F_nu"], ref["F_nu"]) and np.array_equal(out["emergent_flux"], ref["F_nu"][-1])
    assert np.array_equal(out["alpha_line"], ref["alpha_line"]) and np.array_equal(out["total_alphas"], ref["total_alphas"])
    assert out["evaluations"] == ref["evaluations"]


def test_every_visible_gpu_over_rccl(ctx):
    """sdx_device_count() ranks, equal and work-balanced shards: the gathered spectrum and the assembled planes are the
    single-GPU ones bit for bit."""
    n = _lib.load().sdx_device_count()
    if n < 2:
        pytest.skip("one GPU visible: the multi-rank RCCL path needs two")
    atm, nus, lines, cont, th, w = workload(n_lines=3000, lam=(6500.0, 6600.0))
    ref = single_gpu(ctx, atm, nus, lines, cont, th, w)
    grp = DeviceGroup(n)
    for shards in (None, parallel.balanced_shards(parallel.column_cost(nus, lines), n)):
        out = grp.synthesize(nus, atm["temperatures"], atm["dist"], th, w, lines, cont, shards=shards, want_planes=True)
        assert grp.last_gather()["backend"] == "rccl" and grp.last_gather()["ranks"] == n
        assert np.array_equal(out["emergent_flux"], ref["F_nu"][-1]) and np.array_equal(out["F_nu"], ref["F_nu"])
        assert np.array_equal(out["total_alphas"], ref["total_alphas"])
    grp.close()


_LOOPBACK = r"""
import sys, numpy as np
sys.path.insert(0, {root!r})
sys.path.insert(0, {root!r} + "/tests")
from test_gpu_group import workload, single_gpu
from stardis_amd import parallel, _lib
from stardis_amd.group import DeviceGroup
ctx = _lib.default_context()
for kw in (dict(), dict(seed=5, n_lines=9000, lam=(6450.0, 6650.0), step=0.004, mix=(0.9, 0.09, 0.01))):  # short list; long list (indexed + culled pre-pass)
    atm, nus, lines, cont, th, w = workload(**kw)
    ref = single_gpu(ctx, atm, nus, lines, cont, th, w)
    for ranks in (2, 3):
        grp = DeviceGroup(devices=[0] * ranks)
        for shards in (None, parallel.balanced_shards(parallel.column_cost(nus, lines), ranks)):
            out = grp.synthesize(nus, atm["temperatures"], atm["dist"], th, w, lines, cont, shards=shards, want_planes=True)
            info = grp.last_gather()
            assert info["backend"] == "loopback-test-hook" and info["ranks"] == ranks, info
            assert np.array_equal(out["emergent_flux"], ref["F_nu"][-1]), "spectrum differs"
            assert np.array_equal(out["F_nu"], ref["F_nu"]) and np.array_equal(out["alpha_line"], ref["alpha_line"]) and np.array_equal(out["total_alphas"], ref["total_alphas"])
        grp.close()
# more ranks than columns: empty shards take part in the gather with nothing to contribute
atm, nus, lines, cont, th, w = workload(n_lines=30, lam=(6560.0, 6560.05), step=0.01)
assert nus.size < 7
ref = single_gpu(ctx, atm, nus, lines, cont, th, w)
grp = DeviceGroup(devices=[0] * 7)
out = grp.synthesize(nus, atm["temperatures"], atm["dist"], th, w, lines, cont)
assert np.array_equal(out["emergent_flux"], ref["F_nu"][-1]) and np.array_equal(out["F_nu"], ref["F_nu"])
grp.close()
print("IDENTICAL")
"""


def test_sharding_logic_with_several_ranks_on_one_device():
    """2 and 3 ranks on device 0 through the loop-back test hook (SDX_GROUP_LOOPBACK=1): shards of a short and of a long line
    list (culled pre-pass), equal and balanced, padded shards — the assembled result is the single-GPU one bit for bit."""
    env = dict(os.environ, SDX_EXPERIMENT="1", SDX_GROUP_LOOPBACK="1")
    proc = subprocess.run([sys.executable, "-c", _LOOPBACK.format(root=ROOT)], env=env, capture_output=True, text=True, timeout=900)
    assert proc.returncode == 0 and "IDENTICAL" in proc.stdout, proc.stdout[-2000:] + proc.stderr[-3000:]


def test_group_errors(ctx):
    lib = _lib.load()
    with pytest.raises(ValueError, match="listed twice"):
        DeviceGroup(devices=[0, 0])
    with pytest.raises(ValueError, match="not visible"):
        DeviceGroup(devices=[lib.sdx_device_count()])
    atm, nus, lines, cont, th, w = workload(n_lines=50)
    grp = DeviceGroup(1)
    with pytest.raises(ValueError, match="descending"):
        grp.synthesize(nus[::-1], atm["temperatures"], atm["dist"], th, w, lines, cont)
    with pytest.raises(ValueError, match="shard_begin"):
        grp.synthesize(nus, atm["temperatures"], atm["dist"], th, w, lines, cont, shards=[(5, nus.size - 5)])
    grp.close()


def test_rccl_failure_is_sdx_err_comm():
    """A missing RCCL library is a collective error (-3), not a crash and not a silent single-GPU run."""
    code = ("import sys; sys.path.insert(0, %r)\n"
            "from stardis_amd import _lib\nfrom stardis_amd.group import DeviceGroup\n"
            "try:\n    DeviceGroup(1)\nexcept _lib.CommError as e:\n    print('COMM', _lib.load().sdx_last_error_code(), e)\n" % ROOT)
    proc = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SDX_RCCL_LIB="/nonexistent/librccl.so.1"), capture_output=True, text=True,
                          timeout=300)
    assert proc.returncode == 0 and "COMM -3" in proc.stdout and "RCCL not available" in proc.stdout, proc.stdout + proc.stderr


@pytest.mark.parametrize("torch_first", [True, False])
def test_group_works_whichever_rocm_stack_was_loaded_first(torch_first):
    """A PyTorch wheel bundles its own HIP runtime and RCCL; the process runs on whichever HIP runtime was loaded first.  The
    group must open the RCCL that belongs to THAT runtime (an RCCL of the other stack fails in ncclCommInitAll) — both import
    orders, the collective through RCCL each time."""
    first = "import torch\nfrom stardis_amd import _lib\n_lib.load()\n" if torch_first else "from stardis_amd import _lib\n_lib.load()\nimport torch\n"
    code = ("import sys; sys.path.insert(0, %r); sys.path.insert(0, %r + '/tests')\n" % (ROOT, ROOT) + first +
            "import numpy as np\nfrom test_gpu_group import workload, single_gpu\nfrom stardis_amd.group import DeviceGroup\n"
            "atm, nus, lines, cont, th, w = workload(n_lines=200)\n"
            "grp = DeviceGroup(1)\nout = grp.synthesize(nus, atm['temperatures'], atm['dist'], th, w, lines, cont)\n"
            "info = grp.last_gather(); grp.close()\n"
            "ref = single_gpu(_lib.default_context(), atm, nus, lines, cont, th, w)\n"
            "assert info['backend'] == 'rccl' and info['rccl_version'] > 0, info\n"
            "assert np.array_equal(out['F_nu'], ref['F_nu'])\nprint('OK', info['rccl_version'])\n")
    proc = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=600)
    assert proc.returncode == 0 and "OK" in proc.stdout, proc.stdout[-1500:] + proc.stderr[-3000:]
