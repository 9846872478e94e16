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


# The library's far-field rule (include/stardis_hip.h: sdx_far_field_rule) as the planner's DEFAULTS: far_field_rule() asks the
# library itself whenever it can be loaded, and tests/test_abi_cpu.py pins these numbers to it.
FAR_FIELD_MIN_POINTS = 32768  # the automatic rule for the "far_field" option
FAR_NEAR_POINTS = 896        # ... with it a window is evaluated point by point only this close to its centre (3.5 tiles of 256)


def far_field_rule():
    """(min_points, near_points) of the far field as libstardis_hip.so applies it (no GPU needed); the module's defaults if the
    library is not built."""
    import ctypes as C

    from . import _lib

    try:
        lib = _lib.load()
    except Exception:  # noqa: BLE001  (a planning estimate: the defaults are the library's constants anyway)
        return FAR_FIELD_MIN_POINTS, FAR_NEAR_POINTS
    mn, tile, near = C.c_int64(), C.c_int(), C.c_int()
    _lib.check(lib.sdx_far_field_rule(C.byref(mn), C.byref(tile), C.byref(near)))
    return int(mn.value), int(near.value)


def far_field_active(n_nu, ctx=None):
    """Whether a synthesis of a global grid of n_nu points runs with the far field: the context's "far_field" option when a
    context is given (sdx_far_field_active), else the automatic rule."""
    if ctx is not None:
        from . import _lib

        rc = ctx.lib.sdx_far_field_active(ctx.handle, int(n_nu))
        if rc < 0:
            _lib.check(rc)
        return bool(rc)
    return int(n_nu) >= far_field_rule()[0]


def window_work(nus, line_nus, doppler_widths, gammas, alphas, core_weight=1.0, far_weight=None, count_huge=None):
    """Voigt evaluations per grid column, sum over (line, depth) of [lo <= i < hi] with the window rule of
    calc_alan_entries (opacities_solvers/base.py:524-575).  A planning estimate on the host (numpy, O(N_l N_d)): it only
    decides where shard boundaries go, never what is computed.
    core_weight > 1 counts an evaluation in a line core — a narrow window (half-width <= 64 points) or the points of a wide
    window within 15 Doppler widths of the centre, Faddeeva regions II-IV — that many times: those cost 100-250
    instructions where a far-wing evaluation costs 13.
    far_weight (None: no far field): what a window point beyond FAR_NEAR_POINTS of its centre costs when the far field of the
    line kernels takes it — 16 node evaluations per 256 points, ~1/12 with the interpolation.
    count_huge: a list that receives the number of lines whose widest window exceeds 4096 points (the kernels' huge-line list, which
    every tile of every depth scans whatever its position)."""
    nus = np.asarray(nus, dtype=np.float64)
    n = nus.size
    if n < 2 or np.asarray(line_nus).size == 0:
        return np.ones(n)
    d_nu = -np.max(np.diff(nus))
    centre = n - np.searchsorted(nus[::-1], np.asarray(line_nus, dtype=np.float64))  # first index with nu < line_nu
    g = np.asarray(gammas, dtype=np.float64).reshape(centre.size, -1)
    dw = np.asarray(doppler_widths, dtype=np.float64)
    pixels = (g + dw) * np.asarray(alphas, dtype=np.float64) / d_nu * 20.0
    hw = np.minimum(np.where(pixels > 10.0, pixels, 10.0), float(n)).astype(np.int64)
    if count_huge is not None:
        count_huge.append(int(np.count_nonzero(hw.max(axis=1) > 4096)))
    lo = np.clip(centre[:, None] - hw, 0, n)
    hi = np.clip(centre[:, None] + hw, 0, n)
    # +1 where a window opens, -1 where it closes (np.bincount: the same sums as np.add.at, ~30 times faster on 1.7e7 entries)
    cover = np.bincount(lo.ravel(), minlength=n + 1).astype(np.float64) - np.bincount(hi.ravel(), minlength=n + 1)
    if far_weight is not None:
        near = np.minimum(hw, far_field_rule()[1])
        nlo = np.clip(centre[:, None] - near, 0, n)
        nhi = np.clip(centre[:, None] + near, 0, n)
        cover_near = np.bincount(nlo.ravel(), minlength=n + 1).astype(np.float64) - np.bincount(nhi.ravel(), minlength=n + 1)
        cover = cover_near + far_weight * (cover - cover_near)
    if core_weight != 1.0:
        y = g / (np.sqrt(np.pi) * np.pi) / dw
        chw = np.minimum(np.where(hw <= 64, hw, np.maximum(15.0 - y, 0.0) * dw / d_nu + 2.0).astype(np.int64), hw)
        clo = np.clip(centre[:, None] - chw, 0, n)
        chi = np.clip(centre[:, None] + chw, 0, n)
        cover += (core_weight - 1.0) * (np.bincount(clo.ravel(), minlength=n + 1) - np.bincount(chi.ravel(), minlength=n + 1))
    return np.cumsum(cover)[:n]


def scan_work(nus, line_nus, half_width=4096):
    """Lines whose centre index lies within `half_width` grid points of each column: what a tile of a long line list
    scans to find the windows that reach it (stardis_amd/csrc: line_wide_walk, candidates by centre range).  On a
    logarithmic grid the line density per grid point follows the frequency, so this cost is several times higher at the
    blue end than at the red end while the evaluation count is not."""
    nus = np.asarray(nus, dtype=np.float64)
    n = nus.size
    ln = np.asarray(line_nus, dtype=np.float64)
    if n == 0 or ln.size == 0:
        return np.zeros(n)
    centre = np.sort(n - np.searchsorted(nus[::-1], ln))  # first index with nu < line_nu, ascending
    i = np.arange(n)
    return (np.searchsorted(centre, i + half_width, side="left") - np.searchsorted(centre, i - half_width, side="right")).astype(np.float64)


def column_cost(nus, lines, indexed_min_lines=8192, scan_weight=0.6, fixed=None, core_weight=14.0, far_field=None, far_weight=1.0 / 12, ctx=None,
                huge_weight=2.3, huge_free=15000):
    """Estimated cost of every grid column in units of one far-wing Voigt evaluation, for balanced_shards: the window
    evaluations (line cores weighted by core_weight), the candidate scan of long lists (scan_weight per line in range) and
    a constant for the continuum and the formal solution — weights measured on MI355X.  A planning estimate on the host: it
    only decides where shard boundaries go.  far_field: whether the library's far field is on (None: ask the library — the
    "far_field" option of `ctx` when given, else its automatic rule)."""
    if far_field is None: far_field = far_field_active(np.asarray(nus).size, ctx)
    # (per column: continuum, formal solution, the far kernel's nodes — 8000 evaluation-equivalents of the direct sum; with the far field the
    # windows weigh less against them: 16000 measured best over S-c3, S-c4m, S-c3 at R = 5e5 and S-big)
    if fixed is None: fixed = 16000.0 if far_field else 8000.0
    huge = []
    cost = window_work(nus, lines["line_nus"], lines["doppler_widths"], lines["gammas"], lines["alphas"], core_weight,
                       far_weight if far_field else None, count_huge=huge) + fixed
    # a per-column cost of the line kernel that does not depend on what the column holds and grows with the number of HUGE lines (widest
    # window > 4096 points): at R = 1e6 with 1e6 lines (42 503 huge) the kernel costs 46.6 ns per column at the blue end and 26.4 at the red
    # end where the variable terms above say 3.5 : 1 — 18 ns per column are constant — and shards balanced without this term ran 5.3 - 8.0 ms
    # (now 6.3 - 6.6).  Up to ~15 000 huge lines nothing of it shows (S-c3: 2 246, S-c3 at R = 5e5: 5 141, S-c4m: 13 939 — a linear term made
    # all three 1 - 3 % worse).  Not the scan of the huge list (chunk summaries that skip most of it changed no kernel time: EXPERIMENTS).
    # A fit on that one workload (round 6).
    if far_field and np.asarray(lines["line_nus"]).size >= indexed_min_lines:
        cost = cost + huge_weight * max(0, huge[0] - huge_free)
    if np.asarray(lines["line_nus"]).size >= indexed_min_lines:
        cost = cost + scan_weight * scan_work(nus, lines["line_nus"])
    return cost


def window_evaluations(nus, line_nus, doppler_widths, gammas, alphas):
    """sum over (line, depth) of hi - lo with the window rule of calc_alan_entries (opacities_solvers/base.py:524-575): the
    number of Voigt evaluations of the whole grid, counted on the host in chunks of lines (the kernels count the same
    quantity in their pre-pass)."""
    nus = np.asarray(nus, dtype=np.float64)
    n = nus.size
    ln = np.asarray(line_nus, dtype=np.float64)
    if n < 2 or ln.size == 0:
        return 0
    d_nu = -np.max(np.diff(nus))
    centre = n - np.searchsorted(nus[::-1], ln)
    g = np.asarray(gammas, dtype=np.float64).reshape(ln.size, -1)
    dw = np.asarray(doppler_widths, dtype=np.float64)
    al = np.asarray(alphas, dtype=np.float64)
    total = 0
    for a in range(0, ln.size, 65536):
        b = min(a + 65536, ln.size)
        pixels = (g[a:b] + dw[a:b]) * al[a:b] / d_nu * 20.0
        hw = np.minimum(np.where(pixels > 10.0, pixels, 10.0), float(n)).astype(np.int64)
        c = centre[a:b, None]
        total += int(np.sum(np.clip(c + hw, 0, n) - np.clip(c - hw, 0, n)))
    return total


def window_evaluations_performed(nus, line_nus, doppler_widths, gammas, alphas, far_field=True):
    """What the line kernels evaluate with the far field on, as a HOST ESTIMATE: a window point is evaluated where it lies unless its
    256-point tile of the global grid lies wholly inside the window and at least `near_points` - 128 grid points (6 tile half-widths)
    from the line's centre — such a (line, depth, tile) triple costs the tile's 16 Chebyshev nodes instead of its 256 points.  The
    kernels apply the distance test in frequency space and also keep tiles that touch a line's core range; on the logarithmic grids
    of the benchmarks the index-space count here is within a few per cent of theirs.  -> (performed, nominal)."""
    nus = np.asarray(nus, dtype=np.float64)
    n = nus.size
    ln = np.asarray(line_nus, dtype=np.float64)
    if n < 2 or ln.size == 0:
        return 0, 0
    _, near = far_field_rule()
    tile, reach = 256, near - 128
    d_nu = -np.max(np.diff(nus))
    centre = n - np.searchsorted(nus[::-1], ln)
    g = np.asarray(gammas, dtype=np.float64).reshape(ln.size, -1)
    dw = np.asarray(doppler_widths, dtype=np.float64)
    al = np.asarray(alphas, dtype=np.float64)
    nominal = performed = 0
    for a in range(0, ln.size, 65536):
        b = min(a + 65536, ln.size)
        pixels = (g[a:b] + dw[a:b]) * al[a:b] / d_nu * 20.0
        hw = np.minimum(np.where(pixels > 10.0, pixels, 10.0), float(n)).astype(np.int64)
        c = centre[a:b, None]
        lo, hi = np.clip(c - hw, 0, n), np.clip(c + hw, 0, n)
        points = int(np.sum(hi - lo))
        nominal += points
        if not far_field:
            performed += points
            continue
        t0, t1 = -(-lo // tile), hi // tile  # tiles wholly inside [lo, hi): t0 <= T < t1
        inside = np.maximum(t1 - t0, 0)
        n0 = (c - reach - tile // 2) // tile + 1  # first tile whose centre lies within `reach` of the line's
        n1 = -(-(c + reach - tile // 2) // tile)   # first tile at or beyond c + reach
        near_inside = np.maximum(np.minimum(n1, t1) - np.maximum(n0, t0), 0)
        far_tiles = int(np.sum(np.maximum(inside - near_inside, 0)))
        performed += points - far_tiles * (tile - 16)
    return performed, nominal


def balanced_shards(work, world_size, fixed_cost=0.0):
    """Contiguous shards of (nearly) equal work: -> list of (begin, count), one per rank.  `work` is a per-column cost
    (e.g. window_work(...) + a constant per column for the continuum and the formal solution); equal-width shards of a
    long spectrum differ by 30 % in work because windows are wider, in grid points, at the blue end."""
    w = np.asarray(work, dtype=np.float64) + float(fixed_cost)
    n = w.size
    cum = np.concatenate([[0.0], np.cumsum(w)])
    cuts = [0]
    for r in range(1, world_size):
        c = int(np.searchsorted(cum, cum[-1] * r / world_size))
        cuts.append(min(max(c, cuts[-1]), n))
    cuts.append(n)
    return [(cuts[r], cuts[r + 1] - cuts[r]) for r in range(world_size)]


class FluxGatherer:
    """Reusable buffers for the per-step all-gather of emergent-flux shards (no allocation inside the timed loop)."""

    def __init__(self, n_nu, world_size, device, dtype=None, shards=None):
        """shards: optional list of (begin, count) per rank for unequal shards (balanced_shards); default equal blocks."""
        import torch

        self.n_nu, self.world = n_nu, world_size
        self.shards = shards
        self.per = padded_count(n_nu, world_size) if shards is None else max(c for _, c in shards)
        dtype = dtype or torch.float64
        self.send = torch.zeros(self.per, dtype=dtype, device=device)
        self.recv = torch.empty(self.per * world_size, dtype=dtype, device=device)
        self.host = None

    def start(self, local_flux):
        """Enqueue the all-gather behind the work already on the current stream and return at once; the kernels of
        the NEXT step (writing a different flux buffer) are not held back by it.  Call finish() before this
        gatherer's buffers — or the flux buffer it read — are reused.  With the gloo backend and device tensors the
        collective runs on host copies (tests / no RCCL); otherwise it is asynchronous on either backend."""
        import torch.distributed as dist

        self._work = None
        self._staged = False
        if self.world == 1:
            self._single = local_flux
            return
        src = local_flux
        if local_flux.numel() != self.per:
            self.send[: local_flux.numel()] = local_flux
            src = self.send
        if dist.get_backend() == "gloo" and src.is_cuda:  # CPU-side collective: stage through the host, then gather asynchronously
            import torch

            if self.host is None:
                self.host = torch.empty(self.per * self.world, dtype=src.dtype)
            self._host_src = src.cpu()
            self._work = dist.all_gather_into_tensor(self.host, self._host_src, async_op=True)
            self._staged = True
            return
        self._work = dist.all_gather_into_tensor(self.recv, src, async_op=True)

    def finish(self):
        """Wait for the gather started by start() (RCCL: the current stream waits; gloo: the host does) and return the
        assembled spectrum; returns the previous result again when nothing is in flight."""
        if self.world == 1:
            return getattr(self, "_single", None)
        work = getattr(self, "_work", None)
        if work is not None:
            work.wait()
            self._work = None
            if getattr(self, "_staged", False):
                self.recv.copy_(self.host)
        return self._assemble()

    def _assemble(self):
        if self.shards is None:
            return self.recv[: self.n_nu]
        import torch

        return torch.cat([self.recv[r * self.per : r * self.per + c] for r, (_, c) in enumerate(self.shards)])

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
        return self._assemble()


class ClassificationGatherer:
    """The optional SECOND collective of a strong-scaled step (include/stardis_hip.h, sdx_synthesize_classify_dev): every rank
    classifies an equal share of the replicated line list — the largest (gamma + doppler_width) alpha of each of its lines — and
    the shares are exchanged by one all-gather of 8 N_l bytes, instead of every rank streaming the whole list (8 N_l (2 N_d + G)
    bytes per rank and step, which does not shrink with the number of ranks).  Buffers are allocated once.

        g = ClassificationGatherer(n_lines, world, rank, device)
        syn = SpectralSynthesizer(..., classify_share=g.share, m_max=g.full, m_share_out=g.send)
        per step:  syn.step_classify(); g.gather(); syn.step()
    """

    def __init__(self, n_lines, world_size, rank, device):
        import torch

        self.n_lines, self.world, self.rank = int(n_lines), int(world_size), int(rank)
        self.per = -(-self.n_lines // self.world)
        begin = min(self.rank * self.per, self.n_lines)
        self.share = (begin, max(0, min(self.per, self.n_lines - begin)))
        self.send = torch.zeros(self.per, dtype=torch.float64, device=device)
        self.full = torch.zeros(self.per * self.world, dtype=torch.float64, device=device)  # rank r's lines at [r per, r per + per)
        self.host = None

    def gather(self):
        """All-gather the shares behind the work already on the current stream; the stream waits for the result (RCCL) — with
        the gloo backend and device tensors the collective runs on host copies (tests, one device)."""
        import torch
        import torch.distributed as dist

        if self.world == 1:
            self.full[: self.per].copy_(self.send)
            return self.full
        if dist.get_backend() == "gloo" and self.send.is_cuda:
            if self.host is None:
                self.host = torch.empty(self.per * self.world, dtype=torch.float64)
            dist.all_gather_into_tensor(self.host, self.send.cpu())
            self.full.copy_(self.host)
        else:
            dist.all_gather_into_tensor(self.full, self.send)
        return self.full


def gather_flux(local_flux, n_nu, world_size, shards=None):
    """All-gather per-rank emergent-flux shards (1-D tensors of this rank's `count` columns, on the device the
    backend wants) into the full (n_nu,) spectrum on every rank.  shards: (begin, count) per rank when unequal."""
    return FluxGatherer(n_nu, world_size, local_flux.device, local_flux.dtype, shards)(local_flux).clone()


def assemble_shards(shards, n_nu, world_size):
    """Host-side twin of gather_flux for tests: concatenate per-rank arrays in rank order."""
    out = np.concatenate([np.asarray(s) for s in shards], axis=-1)
    assert out.shape[-1] == n_nu
    return out
