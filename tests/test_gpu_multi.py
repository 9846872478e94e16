"""The N > 1 path on hardware: ranks launched by torch.distributed.run synthesize frequency shards and ONE all-gather
reassembles the emergent flux, bit for bit the single-GPU spectrum.  On a 1-GPU box both ranks share device 0 and the
collective is gloo (host-staged); with two or more GPUs the same check runs over RCCL, and bench.py --gpus 2 must start
its own ranks and print its line."""
import json
import os
import socket
import subprocess
import sys

import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _torchrun(script_args, extra_env, world=2, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), *script_args]
    return subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout)


@pytest.mark.parametrize("balanced", [False, True])
def test_two_ranks_on_one_device_gather_the_single_gpu_spectrum(balanced):
    args = [os.path.join(ROOT, "scripts", "two_rank_check.py")] + (["--balanced"] if balanced else [])
    proc = _torchrun(args, {"SDX_BENCH_BACKEND": "gloo", "SDX_BENCH_SINGLE_DEVICE": "1"})
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-3000:]
    assert "IDENTICAL" in proc.stdout


def test_two_ranks_over_rccl_gather_the_single_gpu_spectrum():
    import torch

    if torch.cuda.device_count() < 2:
        pytest.skip("needs two GPUs")
    proc = _torchrun([os.path.join(ROOT, "scripts", "two_rank_check.py"), "--balanced"], {"SDX_BENCH_BACKEND": "nccl"})
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-3000:]
    assert "IDENTICAL" in proc.stdout and "backend nccl" in proc.stdout


def test_bench_self_launches_two_ranks():
    """`python bench.py --gpus 2` without a rendezvous in the environment: two ranks on two GPUs over RCCL when the box has
    them, otherwise both on device 0 with the gloo collective."""
    import torch

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    if torch.cuda.device_count() < 2:
        env.update(SDX_BENCH_BACKEND="gloo", SDX_BENCH_SINGLE_DEVICE="1")
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "5", "--warmup", "2", "--no-cpu-baseline"],
                          env=env, capture_output=True, text=True, timeout=1200)
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-3000:]
    line = [ln for ln in proc.stdout.splitlines() if ln.startswith("{")][-1]
    out = json.loads(line)
    # at N > 1 the headline is BASELINE configs[2] (S-c3: the fixed 120 398-point grid) strong-scaled, with the evidence a driver
    # needs to tell what ran: the backend and rank count torch.distributed saw, every rank's shard and kernel times, and the
    # one-GPU step of the same workload measured in the same run
    assert out["n_gpus"] == 2 and out["value"] > 0 and out["scaling"] == "strong"
    assert out["config"]["workload"].startswith("S-c3") and out["config"]["n_nu_global"] == 120398
    assert len(out["config"]["shards"]) == 2 and sum(c for _, c in out["config"]["shards"]) == 120398
    col = out["collective"]
    assert col["world_size_seen_by_dist"] == 2 and col["backend"] == ("nccl" if torch.cuda.device_count() >= 2 else "gloo")
    assert col["bytes_per_rank"] == 8 * max(c for _, c in out["config"]["shards"])
    assert [r["rank"] for r in out["per_rank"]] == [0, 1] and all(r["avg_kernel_ms"]["k_line_all"] > 0 for r in out["per_rank"])
    assert out["n1_same_workload"]["ms_per_step"] > 0
    assert abs(out["speedup_vs_n1"] - out["n1_same_workload"]["ms_per_step"] / out["ms_per_step"]) < 1e-9
    assert out["ms_per_step_cold"] > 0


@pytest.mark.parametrize("mode", ["one collective", "two collectives", "two in flight", "two collectives, two in flight"])
def test_bench_eight_ranks_on_one_device_rehearse_the_scale_run(mode):
    """The closest thing to the driver's 8-GPU SCALE run a one-GPU box allows: `bench.py --gpus 8` (all ranks on device 0, gloo) —
    eight per-rank entries, shards that tile the 120 398-point grid of configs[2], and the gathered spectrum bit-equal to the
    one-GPU run of the same workload in the same process; with two collectives (the default; --one-collective turns it off) every rank
    classifies an eighth of the list; with two syntheses in flight (the default; --in-flight 1) every rank alternates two syntheses on two
    contexts (each gather ordered behind its own stream)."""
    two_collectives = "two collectives" in mode
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(SDX_BENCH_BACKEND="gloo", SDX_BENCH_SINGLE_DEVICE="1")
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--steps", "3", "--warmup", "1", "--no-cpu-baseline"]
    # (the defaults of a strong-scaling run on several GPUs are "two collectives, two in flight": that mode passes no flag at all)
    default = mode == "two collectives, two in flight"
    extra = [] if default else ((["--two-collectives"] if two_collectives else ["--one-collective"]) + ["--in-flight", "2" if "two in flight" in mode else "1"])
    proc = subprocess.run(cmd + extra, env=env, capture_output=True, text=True, timeout=1800)
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-3000:]
    out = json.loads([ln for ln in proc.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 8 and out["scaling"] == "strong" and out["config"]["n_nu_global"] == 120398
    shards = out["config"]["shards"]
    assert len(shards) == 8 and sum(c for _, c in shards) == 120398
    assert all(shards[r + 1][0] == shards[r][0] + shards[r][1] for r in range(7)) and shards[0][0] == 0
    assert [r["rank"] for r in out["per_rank"]] == list(range(8)) and [r["shard"] for r in out["per_rank"]] == shards
    assert all(r["avg_kernel_ms"]["k_line_all"] > 0 and r["avg_kernel_ms"]["k_classify"] > 0 for r in out["per_rank"])  # culled pre-pass on every rank
    assert out["collective"]["world_size_seen_by_dist"] == 8 and out["collective"]["bytes_per_rank"] == 8 * max(c for _, c in shards)
    assert ("second_collective" in out["collective"]) == two_collectives
    if two_collectives:
        assert out["collective"]["second_collective"]["bytes_per_rank"] == 8 * -(-150000 // 8)
    assert out["gathered_spectrum_equals_n1_bit_for_bit"] is True and out["max_rel_dev_vs_n1"] == 0.0
    assert out["n1_same_workload"]["ms_per_step"] > 0 and out["speedup_vs_n1"] > 0
    assert out["config"]["syntheses_in_flight_per_gpu"] == out["n1_same_workload"]["in_flight"] == (2 if "two in flight" in mode else 1)


def test_shards_of_the_million_line_workload_reproduce_the_unsharded_bits():
    """scripts/r4/big_shard_check.py as a test: ranks 0, 3 and 7 of a balanced 8-way split of S-c4m (1e6 lines: the culled pre-pass
    with its classification stream, counter-driven blocks and gather lists at full size) against the unsharded run, bit for bit."""
    proc = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "r4", "big_shard_check.py"), "S-c4m", "8", "0", "3", "7"],
                          capture_output=True, text=True, timeout=1800)
    assert proc.returncode == 0, proc.stdout[-3000:] + proc.stderr[-3000:]
    assert proc.stdout.count("identical to the unsharded run") == 3
