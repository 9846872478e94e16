"""N ranks synthesize their frequency shards (equal or work-balanced), gather the emergent flux with ONE collective
(stardis_amd.parallel.FluxGatherer: RCCL when the backend is nccl, host-staged gloo otherwise) and rank 0 compares the
gathered spectrum bit for bit with the single-GPU synthesis of the whole grid.
Launched by tests/test_gpu_multi.py through torch.distributed.run; exit code 0 = identical.
env: SDX_BENCH_BACKEND (nccl | gloo), SDX_BENCH_SINGLE_DEVICE=1 (all ranks on device 0: the N > 1 path on a 1-GPU box)"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from stardis_amd import _lib, parallel, synth  # noqa: E402
from stardis_amd.engine import SpectralSynthesizer, shard_bounds  # noqa: E402

balanced = "--balanced" in sys.argv
rank, world, local = parallel.init_from_env(os.environ.get("SDX_BENCH_BACKEND", "nccl"))
if os.environ.get("SDX_BENCH_SINGLE_DEVICE") == "1":
    local = 0
torch.cuda.set_device(local)
stream = torch.cuda.Stream(device=local)
torch.cuda.set_stream(stream)
ctx = _lib.Context(local, stream=stream.cuda_stream)

atm = synth.solar_atmosphere()
nus = synth.tracing_grid(5000.0, 5200.0, R=1.0e5)
lines = synth.synth_lines(nus, atm, 9000, seed=77, mix=(0.85, 0.12, 0.03))  # long enough for the indexed wide path
cont = synth.synth_continuum_state(atm)
th, w = synth.thetas_and_weights(8)
shards = None
if balanced:
    shards = parallel.balanced_shards(parallel.column_cost(nus, lines), world)
begin, count = shards[rank] if shards else shard_bounds(nus.size, world, rank)
lanes = []
for _ in range(2):
    flux = torch.zeros((56, count), dtype=torch.float64, device=f"cuda:{local}")
    syn = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, lines, cont, ctx=ctx, shard=(begin, count), flux_out=flux,
                              track_evaluations=False)
    lanes.append((syn, flux, parallel.FluxGatherer(nus.size, world, flux.device, shards=shards)))
spectra = []
for step in range(4):  # the double-buffered loop of bench.py
    syn, flux, gatherer = lanes[step % 2]
    prev = gatherer.finish()
    if prev is not None and step >= 2:
        spectra.append(prev.clone())
    syn.step()
    gatherer.start(flux[-1])
for syn, flux, gatherer in lanes:
    spectra.append(gatherer.finish().clone())
torch.cuda.synchronize()
ok = True
if rank == 0:
    full = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, lines, cont, ctx=ctx, track_evaluations=False)
    full.step()
    want = full.F_nu()[-1]
    for s in spectra:
        ok = ok and np.array_equal(s.cpu().numpy(), want)
    print(f"two_rank_check: world {world} backend {torch.distributed.get_backend()} balanced {balanced} "
          f"shards {shards or 'equal'}: {'IDENTICAL' if ok else 'DIFFERENT'} ({len(spectra)} gathers, {nus.size} frequencies)", flush=True)
torch.distributed.barrier()
torch.distributed.destroy_process_group()
sys.exit(0 if ok else 1)
