"""The N > 1 path on CPU: two gloo processes shard the frequency axis, and ONE all-gather reassembles the
emergent flux exactly as a single process holds it (stardis_amd.parallel).  The per-shard numbers come from the
CPU oracle here (no GPU in this container); on the GPU box the same sharding is checked bit-for-bit with the HIP
path in tests/test_gpu_engine.py::test_frequency_shards_reassemble_bit_exactly."""
import os
import socket
import sys

import numpy as np
import pytest

from conftest import ROOT


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_nu, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch

    from stardis_amd import parallel

    r, w, _ = parallel.init_from_env("gloo")
    assert (r, w) == (rank, world)
    begin, count = parallel.shard_bounds(n_nu, world, rank)
    full = np.arange(n_nu, dtype=np.float64) * 1.5 + 3.0  # stands for F_nu[-1] of the whole grid
    local = torch.from_numpy(full[begin : begin + count].copy())
    spectrum = parallel.gather_flux(local, n_nu, world)
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), spectrum.numpy())
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("n_nu", [7634, 11])
def test_two_rank_gather_reassembles_spectrum(tmp_path, n_nu):
    import torch.multiprocessing as mp

    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_nu, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    want = np.arange(n_nu, dtype=np.float64) * 1.5 + 3.0
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / f"rank{r}.npy"), want)


def test_sharded_oracle_equals_global_oracle():
    """Sharding must keep the GLOBAL window rule: computing each shard's columns from the global grid and
    concatenating equals the unsharded result; running the algorithm independently on a sub-grid does not."""
    import oracle
    from stardis_amd import synth
    from stardis_amd.engine import shard_bounds

    atm = synth.solar_atmosphere()
    nus = synth.tracing_grid(6560.0, 6570.0, step=0.05)
    ln = synth.synth_lines(nus, atm, 60, seed=31, mix=(0.6, 0.3, 0.1))
    full = oracle.calc_alan_entries(56, nus, ln["line_nus"], ln["doppler_widths"], ln["gammas"], ln["alphas"])
    b, c = shard_bounds(nus.size, 2, 1)
    sub = nus[b : b + c]
    keep = (ln["line_nus"] >= sub.min()) & (ln["line_nus"] <= sub.max())
    independent = oracle.calc_alan_entries(56, sub, ln["line_nus"][keep], ln["doppler_widths"][keep], ln["gammas"][keep], ln["alphas"][keep])
    assert not np.allclose(independent, full[:, b : b + c], rtol=1e-6, atol=0.0)  # lines outside the sub-grid are dropped (base.py:393-395)


def _worker_uneven(rank, world, port, n_nu, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch

    from stardis_amd import parallel

    parallel.init_from_env("gloo")
    work = np.linspace(3.0, 1.0, n_nu)  # heavier at the blue end, like window counts
    shards = parallel.balanced_shards(work, world)
    begin, count = shards[rank]
    full = np.arange(n_nu, dtype=np.float64) * 1.5 + 3.0
    spectrum = parallel.gather_flux(torch.from_numpy(full[begin : begin + count].copy()), n_nu, world, shards)
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), spectrum.numpy())
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_gather_with_work_balanced_shards(tmp_path):
    import torch.multiprocessing as mp

    n_nu, port = 1001, _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker_uneven, args=(r, 2, port, n_nu, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    want = np.arange(n_nu, dtype=np.float64) * 1.5 + 3.0
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / f"rank{r}.npy"), want)


def test_balanced_shards_partition_and_balance():
    """balanced_shards tiles the grid exactly, and on the window-count estimate of a line list (checked against the
    oracle's own windows) evens out the work that equal-width shards leave 30 % apart."""
    import oracle
    from stardis_amd import parallel, synth

    atm = synth.solar_atmosphere()
    nus = synth.tracing_grid(4000.0, 8000.0, R=2.0e4)
    ln = synth.synth_lines(nus, atm, 1500, seed=5)
    work = parallel.window_work(nus, ln["line_nus"], ln["doppler_widths"], ln["gammas"], ln["alphas"])
    _, evals = oracle.calc_alan_entries(56, nus, ln["line_nus"], ln["doppler_widths"], ln["gammas"], ln["alphas"], return_evals=True)
    assert int(work.sum()) == evals  # the estimate IS the window rule
    for world in (1, 2, 3, 8):
        shards = parallel.balanced_shards(work, world)
        assert shards[0][0] == 0 and sum(c for _, c in shards) == nus.size
        for (b0, c0), (b1, _) in zip(shards, shards[1:]):
            assert b0 + c0 == b1
        per = [work[b : b + c].sum() for b, c in shards]
        assert max(per) <= 1.02 * np.mean(per) + work.max()
    equal = [work[b : b + c].sum() for b, c in (parallel.shard_bounds(nus.size, 8, r) for r in range(8))]
    assert max(equal) > 1.1 * np.mean(equal)


def _worker_overlap(rank, world, port, n_nu, out_dir):
    """The double-buffered loop of bench.py: the gather of step k is in flight while step k + 1 fills the other buffer."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch

    from stardis_amd import parallel

    parallel.init_from_env("gloo")
    work = np.linspace(3.0, 1.0, n_nu)
    shards = parallel.balanced_shards(work, world)  # unequal shards: the shorter ones are padded for the collective
    begin, count = shards[rank]
    lanes = [(torch.zeros(count, dtype=torch.float64), parallel.FluxGatherer(n_nu, world, "cpu", shards=shards)) for _ in range(2)]
    results = []
    for step in range(5):
        buf, gatherer = lanes[step % 2]
        prev = gatherer.finish()  # the gather that last read this buffer (two steps ago)
        if step >= 2:
            results.append(prev.clone())
        buf.copy_(torch.from_numpy((np.arange(n_nu, dtype=np.float64) + 1000.0 * step)[begin : begin + count]))  # "step k" output
        gatherer.start(buf)
    for step in (3, 4):
        results.append(lanes[step % 2][1].finish().clone())
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), torch.stack(results).numpy())
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


def test_two_rank_overlapped_gather_double_buffering(tmp_path):
    import torch.multiprocessing as mp

    n_nu, port = 777, _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker_overlap, args=(r, 2, port, n_nu, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    want = np.stack([np.arange(n_nu, dtype=np.float64) + 1000.0 * k for k in range(5)])
    for r in range(2):
        assert np.array_equal(np.load(tmp_path / f"rank{r}.npy"), want)


def test_bench_starts_its_own_ranks_when_no_rendezvous_is_set(tmp_path):
    """`python bench.py --gpus 2` with no WORLD_SIZE must launch two ranks itself (torch.distributed.run children) instead of
    refusing.  Without a GPU the ranks then fail loudly on the missing device — that failure, coming from BOTH ranks, is what
    shows they were started (on the GPU box tests/test_gpu_multi.py runs the same command to completion)."""
    import subprocess

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["SDX_BENCH_BACKEND"] = "gloo"
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0", "--no-cpu-baseline", "--no-secondary"],
                          env=env, capture_output=True, text=True, timeout=600)
    text = proc.stdout + proc.stderr
    import torch

    if torch.cuda.is_available() and torch.cuda.device_count() >= 1:
        pytest.skip("a GPU is visible: covered by the GPU test")
    assert proc.returncode != 0
    assert "launch with torch.distributed.run" not in text
    assert text.count("no HIP device") >= 1 or "Found no NVIDIA driver" in text or "HIP" in text, text[-2000:]


def _worker_classes(rank, world, port, n_lines, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    import torch

    from stardis_amd import parallel

    parallel.init_from_env("gloo")
    g = parallel.ClassificationGatherer(n_lines, world, rank, "cpu")
    begin, count = g.share
    m_max = np.arange(n_lines, dtype=np.float64) ** 1.5 + 0.25  # stands for the per-line maxima
    g.send[:count] = torch.from_numpy(m_max[begin : begin + count])
    full = g.gather()
    np.save(os.path.join(out_dir, f"classes{rank}.npy"), np.concatenate([[begin, count, g.per], full.numpy()]))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("n_lines", [150000, 9001])
def test_two_ranks_exchange_their_classification_shares(tmp_path, n_lines):
    """The optional second collective of strong-scaled long lists (parallel.ClassificationGatherer): each rank owns an equal share of
    the line list, and after ONE all-gather every rank holds the per-line maxima of the whole list at index = line."""
    import torch.multiprocessing as mp

    port = _free_port()
    ctx = mp.get_context("spawn")
    procs = [ctx.Process(target=_worker_classes, args=(r, 2, port, n_lines, str(tmp_path))) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(120)
        assert p.exitcode == 0
    want = np.arange(n_lines, dtype=np.float64) ** 1.5 + 0.25
    shares = []
    for r in range(2):
        got = np.load(tmp_path / f"classes{r}.npy")
        begin, count, per = (int(x) for x in got[:3])
        assert per == -(-n_lines // 2) and begin == r * per and got[3:].size == 2 * per
        assert np.array_equal(got[3 : 3 + n_lines], want)
        shares.append((begin, count))
    assert shares[0][1] + shares[1][1] == n_lines and shares[1][0] == shares[0][1]
