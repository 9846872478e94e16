"""Frequency sharding across the GPUs of one node: one process per GPU, one collective.

The hot path shards along the frequency axis with no data-path exchange (every output column depends
only on its own frequency).  Each rank runs the fused synthesis on a contiguous block of
the GLOBAL frequency index — the window rule keeps using the global grid, so results are identical to
a single-GPU run — and the emergent flux F_nu[-1] is gathered with ONE all-gather (RCCL over xGMI when
the backend is "nccl"; gloo works for CPU tests).  Shards are padded to equal length for the
collective and trimmed afterwards.
"""
import os

import numpy as np

from .engine import shard_bounds  # noqa: F401  (re-exported)


def init_from_env(backend=None):
    """Initialise torch.distributed from RANK / WORLD_SIZE / MASTER_* (torch.distributed.run sets them).
    Returns (rank, world_size, local_rank).  Single-process runs return (0, 1, 0) without touching torch."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch
        import torch.distributed as dist

        if not dist.is_initialized():
            if backend is None:
                backend = "nccl" if torch.cuda.is_available() else "gloo"
            if backend == "nccl":
                torch.cuda.set_device(local)
            dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def padded_count(n_nu, world_size):
    return -(-n_nu // world_size)


class FluxGatherer:
    """Reusable buffers for the per-step all-gather of emergent-flux shards (no allocation inside the timed loop)."""

    def __init__(self, n_nu, world_size, device, dtype=None):
        import torch

        self.n_nu, self.world = n_nu, world_size
        self.per = padded_count(n_nu, world_size)
        dtype = dtype or torch.float64
        self.send = torch.zeros(self.per, dtype=dtype, device=device)
        self.recv = torch.empty(self.per * world_size, dtype=dtype, device=device)
        self.host = None

    def start(self, local_flux):
        """Enqueue the all-gather behind the work already on the current stream and return at once; the kernels of
        the NEXT step (writing a different flux buffer) are not held back by it.  Call finish() before this
        gatherer's buffers — or the flux buffer it read — are reused."""
        import torch.distributed as dist

        self._work = None
        if self.world == 1:
            return
        if dist.get_backend() != "nccl":  # CPU collective (tests): nothing to overlap
            self(local_flux)
            return
        src = local_flux
        if local_flux.numel() != self.per:
            self.send[: local_flux.numel()] = local_flux
            src = self.send
        self._work = dist.all_gather_into_tensor(self.recv, src, async_op=True)

    def finish(self):
        """Make the current stream wait for the gather started by start()."""
        work = getattr(self, "_work", None)
        if work is not None:
            work.wait()
            self._work = None
        return self.recv[: self.n_nu]

    def __call__(self, local_flux):
        import torch
        import torch.distributed as dist

        if self.world == 1:
            return local_flux
        src = local_flux
        if local_flux.numel() != self.per:  # last, shorter shard: pad
            self.send[: local_flux.numel()] = local_flux
            src = self.send
        if dist.get_backend() == "gloo" and src.is_cuda:  # CPU-side collective (tests / no RCCL): stage through the host
            if self.host is None:
                self.host = torch.empty(self.per * self.world, dtype=src.dtype)
            dist.all_gather_into_tensor(self.host, src.cpu())
            self.recv.copy_(self.host)
        else:
            dist.all_gather_into_tensor(self.recv, src)
        return self.recv[: self.n_nu]


def gather_flux(local_flux, n_nu, world_size):
    """All-gather per-rank emergent-flux shards (1-D tensors of this rank's `count` columns, on the device the
    backend wants) into the full (n_nu,) spectrum on every rank."""
    return FluxGatherer(n_nu, world_size, local_flux.device, local_flux.dtype)(local_flux).clone()


def assemble_shards(shards, n_nu, world_size):
    """Host-side twin of gather_flux for tests: concatenate per-rank arrays in rank order."""
    out = np.concatenate([np.asarray(s) for s in shards], axis=-1)
    assert out.shape[-1] == n_nu
    return out
