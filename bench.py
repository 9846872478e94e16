#!/usr/bin/env python
"""bench.py — spectral points / s of the STARDIS hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload S-c2]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one full pass of the hot path over inputs already resident in HBM: window pre-pass + line
opacity (Voigt/Faddeeva over the whole line list) + continuum (H- bf table, H I bf/ff, Thomson) + total +
LTE formal solution over N_theta angles -> F_nu (N_d, N_nu).  Metric: spectral points per second =
N_nu * N_depth * steps / time (BASELINE.json).  At N GPUs the frequency axis is sharded in contiguous
blocks of the global index (fixed points per GPU: the window's resolving power grows with N), and every
step ends with ONE all-gather of the emergent flux (RCCL).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X HBM3E, /opt/skills/guides/MI355X_MICROARCH.md
FP64_VECTOR_PEAK_TFLOPS = 78.6


def build_workload(tag, world, inputs="dense", scaling="weak"):
    from stardis_amd import synth

    cfg = synth.WORKLOADS[tag]
    atm = synth.cool_dwarf_atmosphere() if cfg.get("atmosphere") == "cool_dwarf" else synth.solar_atmosphere()
    base = synth.tracing_grid(cfg["lam0"], cfg["lam1"], cfg.get("R"), cfg.get("step"))
    n_per_gpu = base.size
    # weak scaling: same window and line list, N x the resolving power -> N x the grid points
    # strong scaling: the workload's own grid, split N ways
    nus = base if (world == 1 or scaling == "strong") else synth.tracing_grid(cfg["lam0"], cfg["lam1"], cfg.get("R", 1.0), None, n_override=n_per_gpu * world)
    if inputs == "linelist":  # per-line scalars: alpha, gamma and the Doppler width are generated in the pre-pass
        lines = synth.synth_linelist(nus, atm, cfg["n_lines"], synth.SEED)
    else:
        lines = synth.synth_lines(nus, atm, cfg["n_lines"], synth.SEED, cfg["gamma_per_depth"])
    cont = synth.synth_continuum_state(atm)
    thetas, weights = synth.thetas_and_weights(synth.N_THETAS)
    return dict(atm=atm, nus=nus, lines=lines, cont=cont, thetas=thetas, weights=weights, n_per_gpu=n_per_gpu)


def cpu_baseline(w, budget_s=25.0):
    """The reference algorithm restated in C (oracle/, OpenMP over lines / frequencies exactly like the numba
    prange loops, per-thread accumulator slabs included), timed on this box's host cores on the same workload at
    several thread counts; the fastest is reported.  A reported baseline, not the target."""
    import oracle
    from stardis_amd import constants as K

    atm, nus, ln, cont = w["atm"], w["nus"], w["lines"], w["cont"]
    nd = atm["temperatures"].size
    max_threads = oracle.num_threads()
    spec = None if isinstance(ln, dict) else ln

    def dense_tables():
        """what the reference forms on the host before its line kernel (plasma/base.py:200-321, broadening.py:659-732)"""
        args = (spec.atomic_number, spec.ion_number, spec.ionization_energy, spec.upper_energy, spec.lower_energy, spec.A_ul)
        state = (spec.electron_density, spec.temperature, spec.h_density)
        gam = (oracle.calc_vald_gamma(*args, spec.stark, spec.waals, spec.mass, *state, flags=spec.flags) if spec.gamma_mode == 1
               else oracle.calc_gamma(*args, *state, flags=spec.flags))
        return dict(line_nus=spec.nu, gammas=gam, doppler_widths=oracle.doppler_widths(spec.nu, spec.mass, spec.temperature, spec.microturbulence),
                    alphas=oracle.alpha_line_linelist(spec.e_low_ev, spec.g_lo, spec.strength, spec.nu, spec.pop_row, spec.pop, spec.temperature,
                                                      spec.alpha_coefficient))

    def one_pass():
        ln = dense_tables() if spec is not None else w["lines"]
        line = oracle.calc_alan_entries(nd, nus, ln["line_nus"], ln["doppler_widths"], ln["gammas"], ln["alphas"])
        lam = K.nu_to_angstrom(nus)
        cutoff = (cont["ionization_energy"] - cont["level_excitation"]) / K.H_CGS
        total = oracle.alpha_file_1d(lam, cont["hminus_bf_wavelength"], cont["hminus_bf_cross_section"], cont["n_hminus"])
        total = total + oracle.alpha_bf(nus, [0, len(cutoff)], [0], cutoff, cont["level_density"])
        total = total + oracle.alpha_ff(nus, atm["temperatures"], [1], cont["n_e"] * cont["n_h2"])
        total = total + oracle.alpha_electron(nus.size, cont["n_e"])
        total = total + line
        F, _ = oracle.raytrace(nus, atm["temperatures"], atm["dist"], w["thetas"], w["weights"], total)
        return F

    pts = nus.size * nd
    sweep, F = {}, None
    counts = sorted({1, min(16, max_threads), min(64, max_threads), max_threads})
    for n in counts:
        oracle.set_num_threads(n)
        times, spent = [], 0.0
        while len(times) < 2 or (spent < budget_s / len(counts) and len(times) < 6):
            t0 = time.perf_counter()
            F = one_pass()
            times.append(time.perf_counter() - t0)
            spent += times[-1]
        sweep[n] = min(times)
    oracle.set_num_threads(max_threads)
    best = min(sweep, key=sweep.get)
    return dict(
        value=pts / sweep[best], unit="spectral points/s", cores=best, kind="port",
        sample=f"full workload ({nus.size} nu x {nd} depths, {spec.n_lines if spec is not None else ln['line_nus'].size} lines, {len(w['thetas'])} angles), best pass of each thread count; "
               + ", ".join(f"{n} thr: {t * 1e3:.0f} ms" for n, t in sweep.items()),
        host_cores=max_threads,
    ), F


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--workload", default="S-c2")
    ap.add_argument("--inputs", choices=("dense", "linelist"), default="dense",
                    help="line list as the reference's dense (N_l, N_d) tables, or as per-line scalars expanded on the device")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak",
                    help="weak (default): fixed grid points per GPU; strong: the workload's grid split across the GPUs in shards of equal "
                         "estimated work (stardis_amd.parallel.balanced_shards)")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import torch

    from stardis_amd import _lib, parallel
    from stardis_amd.engine import SpectralSynthesizer, shard_bounds

    # test hooks: SDX_BENCH_BACKEND=gloo and SDX_BENCH_SINGLE_DEVICE=1 let the N > 1 path run on a 1-GPU box
    rank, world, local = parallel.init_from_env(os.environ.get("SDX_BENCH_BACKEND", "nccl"))
    if os.environ.get("SDX_BENCH_SINGLE_DEVICE") == "1":
        local = 0
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    torch.cuda.set_device(local)
    import torch.distributed as dist

    w = build_workload(args.workload, world, args.inputs, args.scaling)
    nus, atm = w["nus"], w["atm"]
    nd = atm["temperatures"].size
    shards = None
    if args.scaling == "strong" and world > 1 and isinstance(w["lines"], dict):
        ln_ = w["lines"]
        shards = parallel.balanced_shards(parallel.window_work(nus, ln_["line_nus"], ln_["doppler_widths"], ln_["gammas"], ln_["alphas"]), world)
    begin, count = shards[rank] if shards else shard_bounds(nus.size, world, rank)

    # the library enqueues on torch's current (non-default, capturable) stream, so the RCCL gather is
    # stream-ordered behind the kernels without a host sync
    stream = torch.cuda.Stream(device=local)
    torch.cuda.set_stream(stream)
    ctx = _lib.Context(local, stream=stream.cuda_stream)
    flux = torch.zeros((nd, count), dtype=torch.float64, device=f"cuda:{local}")
    syn = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], w["thetas"], w["weights"], w["lines"], w["cont"],
                              ctx=ctx, shard=(begin, count), flux_out=flux)
    syn.step()
    ctx.synchronize()
    evals = syn.evaluations()
    syn.count_evaluations = False  # known now; the counter costs a memset + a copy per step
    if not args.no_graph:
        syn.capture()

    # N > 1: two flux buffers alternate so that the all-gather of step k (RCCL, its own stream) overlaps the kernels
    # of step k+1; a buffer is reused only after its gather has been waited for.  SDX_BENCH_SYNC_GATHER=1 falls back
    # to a blocking gather per step.
    overlap_gather = world > 1 and os.environ.get("SDX_BENCH_SYNC_GATHER") != "1"
    lanes = [(syn, flux, parallel.FluxGatherer(nus.size, world, flux.device, shards=shards))]
    if overlap_gather:
        flux_b = torch.zeros_like(flux)
        syn_b2 = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], w["thetas"], w["weights"], w["lines"], w["cont"],
                                     ctx=ctx, shard=(begin, count), flux_out=flux_b, track_evaluations=False)
        if not args.no_graph:
            syn_b2.capture()
        lanes.append((syn_b2, flux_b, parallel.FluxGatherer(nus.size, world, flux.device, shards=shards)))
    counter = [0]

    def step():
        s_, f_, g_ = lanes[counter[0] % len(lanes)]
        counter[0] += 1
        if overlap_gather:
            g_.finish()  # the gather that last read this lane's flux buffer
            s_.step()
            g_.start(f_[-1])
            return None
        s_.step()
        return g_(f_[-1])

    def drain():
        if overlap_gather:
            for _, _, g_ in lanes:
                g_.finish()

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    drain()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        spectrum = step()
    drain()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=f"cuda:{local}" if dist.get_backend() == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # per-kernel durations, live, with HIP events on the launch stream (eager launches, separate from the timed region)
    ctx.call("sdx_profile_enable", 1)
    ctx.call("sdx_profile_reset")
    n_prof = 20
    for _ in range(n_prof):
        syn.enqueue()
    ctx.synchronize()
    kern = {}
    import ctypes as C

    for name in ("k_dnu_partial", "k_prepass_continuum", "k_line_prepass", "k_line_all", "k_line_wide", "k_line_narrow", "k_reduce_partials", "k_total_alphas", "k_raytrace"):
        n, ms = C.c_int64(), C.c_double()
        _lib.check(ctx.lib.sdx_profile_get(ctx.handle, name.encode(), C.byref(n), C.byref(ms)))
        if n.value:
            kern[name] = ms.value / n.value
    ctx.call("sdx_profile_enable", 0)
    ctx.call("sdx_profile_reset")

    # secondary figure (not `value`): two independent syntheses in flight on two streams — what a parameter grid
    # of stars would use; each is still a full pass, they only overlap on the device
    pipelined = None
    if world == 1:
        ctx_b = _lib.Context(local)
        syn_b = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], w["thetas"], w["weights"], w["lines"], w["cont"],
                                    ctx=ctx_b, track_evaluations=False)
        if not args.no_graph:
            syn_b.capture()
        for _ in range(max(2, args.warmup // 2)):
            syn.step()
            syn_b.step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(args.steps):
            (syn if k % 2 == 0 else syn_b).step()
        torch.cuda.synchronize()
        pipelined = nus.size * nd * args.steps / (time.perf_counter() - t0)

    if rank == 0:
        pts_total = nus.size * nd
        ms_per_step = elapsed / args.steps * 1e3
        value = pts_total * args.steps / elapsed
        n_l, g_cols = syn.n_lines, syn.gamma_cols
        dom = max(kern, key=kern.get)
        alg_bytes = {
            # SURVEY §8d per-stage figures, for the columns this rank produced
            "k_line_all": 8 * (n_l * (1 + 2 * nd + g_cols) + nus.size + nd * count),
            "k_line_wide": 8 * (n_l * (1 + 2 * nd + g_cols) + nus.size + nd * count),
            "k_raytrace": 16 * nd * count,
        }.get(dom, syn.algorithmic_bytes())
        achieved = alg_bytes / (kern[dom] * 1e-3) / 1e9
        out = {
            "metric": "spectral points/sec (N_nu x N_depth)",
            "value": value,
            "unit": "spectral points/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{args.workload}: {'cool-dwarf' if 'm' == args.workload[-1] else 'solar'} MARCS structure, {synth_desc(args.workload)}, fp64",
                "n_nu_global": int(nus.size),
                "n_nu_per_gpu": int(count),
                "n_depth": int(nd),
                "n_lines": int(n_l),
                "n_theta": int(len(w["thetas"])),
                "voigt_evaluations_global": int(evals),
                "parallelism": f"nu-shard x{world}" + (", 1 all-gather of F_nu[-1] per step" + (" overlapped with the next step" if overlap_gather else "") if world > 1 else ""),
                "hip_graph": not args.no_graph,
                "line_inputs": args.inputs,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": dom,
                "achieved": achieved,
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBPS,
                "traffic": profiled_traffic(dom) if (world == 1 and args.workload == "S-c2") else None,
                "algorithmic_bytes_per_launch": int(alg_bytes),
                "avg_kernel_ms": kern,
                "note": "path is fp64-VALU bound (Faddeeva evaluations), not HBM bound: see DESIGN.md; evaluations/s below",
                "voigt_evaluations_per_s": (evals / world) / ((kern.get("k_line_all", 0.0) + kern.get("k_line_wide", 0.0) + kern.get("k_line_narrow", 0.0)) * 1e-3),
            },
        }
        valu = profiled_valu(kern) if (world == 1 and args.workload == "S-c2") else None
        if valu is not None:
            out["roofline_fp64_valu"] = valu
        if world == 1 and not args.no_cpu_baseline:
            base, F_cpu = cpu_baseline(w)
            out["cpu_baseline"] = base
            F_gpu = flux.cpu().numpy()
            out["parity_vs_cpu_oracle"] = {
                "emergent_flux_max_rel_err": float(np.max(np.abs(F_gpu[-1] - F_cpu[-1]) / np.abs(F_cpu[-1]))),
                "F_nu_max_rel_err": float(np.max(np.abs(F_gpu[1:] - F_cpu[1:]) / np.abs(F_cpu[1:]))),
            }
            out["speedup_vs_cpu_baseline"] = value / base["value"]
        if pipelined is not None:
            out["throughput_two_syntheses_in_flight"] = pipelined
        print(json.dumps(out))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def profiled_traffic(kernel):
    """HBM bytes per launch of `kernel` from the committed rocprofv3 PMC passes (profiles/, S-c2, separate FETCH_SIZE
    and WRITE_SIZE runs).  Raw counters: these kernels issue 8-byte-per-lane accesses, for which the gfx950
    16-byte-stream correction does not apply (profiles/README.md).  None when the files are absent."""
    import csv

    total = 0.0
    for name in ("FETCH_SIZE", "WRITE_SIZE"):
        path = os.path.join(ROOT, "profiles", f"r01_S-c2_pmc_{name}.csv")
        if not os.path.exists(path):
            return None
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(path)) if kernel in r["Kernel_Name"] and r["Counter_Name"] == name]
        if not vals:
            return None
        total += sum(vals) / len(vals) * 1024.0
    return total


FP64_VALU_PEAK = 439.0e9  # wave-level fp64 VALU instructions/s the chip sustains (scripts/fp64_peak.hip: 55 TFLOP/s FMA)


def profiled_valu(kern):
    """The bound that actually limits this path: fp64 VALU issue.  Wave-level VALU instructions per step from the
    committed SQ_INSTS_VALU pass (profiles/, S-c2) over the kernel time measured live in this run."""
    import csv

    path = os.path.join(ROOT, "profiles", "r01_S-c2_pmc_SQ.csv")
    if not os.path.exists(path):
        return None
    rows = [r for r in csv.DictReader(open(path)) if r["Counter_Name"] == "SQ_INSTS_VALU"]
    insts, t_ms = 0.0, 0.0
    for name, ms in kern.items():
        vals = [float(r["Counter_Value"]) for r in rows if name in r["Kernel_Name"]]
        if not vals:
            return None
        insts += sum(vals) / len(vals)
        t_ms += ms
    achieved = insts / (t_ms * 1e-3)
    return {"bound": "fp64-valu-issue", "achieved": achieved / 1e9, "peak": FP64_VALU_PEAK / 1e9, "unit": "G wave-instr/s",
            "frac": achieved / FP64_VALU_PEAK, "valu_wave_instr_per_step": insts, "kernel_ms_per_step": t_ms}


def synth_desc(tag):
    from stardis_amd import synth

    c = synth.WORKLOADS[tag]
    grid = f"R={c['R']:.0f}" if "R" in c else f"step {c['step']} A"
    return f"{c['lam0']:.0f}-{c['lam1']:.0f} A at {grid}"


if __name__ == "__main__":
    main()
