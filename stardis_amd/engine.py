"""Device-resident spectral synthesis: line opacity + continuum + formal solution for data already in HBM.

This is the path bench.py times and the one a frequency-sharded multi-GPU run uses: inputs are
uploaded once, every step is a handful of kernel launches on one stream (optionally replayed as a
hipGraph), and the only host traffic is the final flux.  Numerically it is the same sequence as
calc_alphas + raytrace in stardis_amd.radiation_field (and hence the reference's
radiation_field/base.py:104-115): total = ((((file + bf) + ff) + rayleigh) + electron) + line.
"""
import ctypes as C

import numpy as np

from . import _lib
from . import constants as K
from . import linelist as LL
from ._lib import Continuum, default_context, ptr_of


def shard_bounds(n_nu, world_size, rank):
    """Contiguous block of the GLOBAL frequency index owned by `rank`."""
    per = -(-n_nu // world_size)
    begin = min(rank * per, n_nu)
    return begin, min(per, n_nu - begin)


def _require_f64_buffer(name, buf, n_min):
    """A DeviceArray or contiguous CUDA tensor of at least n_min float64 values."""
    ptr_of(buf)  # (type, contiguity)
    if isinstance(buf, _lib.DeviceArray):
        dtype_ok, numel = buf.dtype == np.dtype(np.float64), int(np.prod(buf.shape, dtype=np.int64))
    else:
        dtype_ok, numel = str(buf.dtype) == "torch.float64", int(buf.numel())
    if not dtype_ok:
        raise TypeError(f"{name} must hold float64 values")
    if numel < n_min:
        raise ValueError(f"{name} holds {numel} values, {n_min} are needed")


class SpectralSynthesizer:
    def __init__(self, nus, temperatures, dist, thetas, theta_weights, lines, continuum=None, ctx=None, shard=None,
                 flux_out=None, track_evaluations=True, keep_line=True, keep_total=True, classify_share=None, m_max=None, m_share_out=None):
        """nus: global grid (descending).  lines: dict(line_nus, doppler_widths, gammas, alphas) in the
        reference layout (N_l, N_d), or a stardis_amd.linelist.LineList (per-line scalars; the pre-pass generates the
        three values per (line, depth) itself).  continuum: dict as produced by synth.synth_continuum_state or None.
        shard: (begin, count) of the global frequency index computed here (default: everything).
        flux_out: optional contiguous CUDA tensor (N_d, count) to receive F_nu (e.g. for an RCCL gather).
        classify_share, m_max: the two-collective mode of frequency shards of long lists (include/stardis_hip.h,
        sdx_synthesize_classify_dev) — classify_share = (first line, number of lines) this rank classifies, m_max = a device
        buffer of n_lines doubles (DeviceArray or CUDA tensor) that holds every rank's share once the caller has gathered it.
        m_share_out: optional device buffer that receives the share's values from its index 0 (the send buffer of the all-gather;
        default: they are written in place, m_max[first line ...]).
        A step is then enqueue_classify() [-> the caller's all-gather of m_max] -> enqueue()."""
        self.ctx = ctx or default_context()
        c = self.ctx
        nus = np.ascontiguousarray(nus, dtype=np.float64)
        if np.any(np.diff(nus) >= 0):
            raise ValueError("tracing frequencies must be strictly descending (stardis/base.py:34)")
        self.linelist = None
        if isinstance(lines, (LL.LineList, LL.DeviceLineList)):
            (lines.host if isinstance(lines, LL.DeviceLineList) else lines).check_sorted()
            self.linelist = lines if isinstance(lines, LL.DeviceLineList) else lines.upload(c)
        elif np.any(np.asarray(lines["doppler_widths"]) == 0):
            raise ZeroDivisionError("float division by zero")  # voigt.py:148
        self.n_nu = nus.size
        self.begin, self.count = shard if shard is not None else (0, self.n_nu)
        t = np.ascontiguousarray(temperatures, dtype=np.float64).reshape(-1)
        self.n_depth = t.size
        thetas = np.asarray(thetas, dtype=np.float64)
        self.n_theta = thetas.size
        ray = np.asarray(dist, dtype=np.float64).reshape(-1, 1) / np.cos(thetas)  # radiation_field_solvers/base.py:302-305

        self.nus_host = nus
        self.d_nus = c.upload(nus)
        self.d_t = c.upload(t)
        self.d_ray = c.upload(ray)
        self.d_w = c.upload(np.asarray(theta_weights, dtype=np.float64))
        if self.linelist is not None:
            if self.linelist.n_depth != self.n_depth:
                raise ValueError("line list and model disagree on the number of depth points")
            self.n_lines, self.gamma_cols = self.linelist.n_lines, self.linelist.gamma_cols
        else:
            ln = np.ascontiguousarray(lines["line_nus"], dtype=np.float64)
            self.n_lines = ln.size
            g = np.ascontiguousarray(lines["gammas"], dtype=np.float64)
            g = g.reshape(self.n_lines, -1) if self.n_lines else np.zeros((0, 1))
            dw = np.ascontiguousarray(lines["doppler_widths"], dtype=np.float64)
            al = np.ascontiguousarray(lines["alphas"], dtype=np.float64)
            if ln.size > 1 and np.any(ln[1:] < ln[:-1]):
                # the kernels take the list in ascending frequency (what calc_alpha_line_at_nu passes, base.py:392-397); the
                # reference's calc_alan_entries accepts any order, so an unsorted list is sorted (stably) here, not refused
                order = np.argsort(ln, kind="stable")
                ln, g, dw, al = (np.ascontiguousarray(a[order]) for a in (ln, g, dw, al))
            self.gamma_cols = g.shape[1]
            self.d_ln = c.upload(ln)
            self.d_dw = c.upload(dw)
            self.d_g = c.upload(g)
            self.d_a = c.upload(al)

        self._keep = []
        self.cont = self._build_continuum(continuum, nus, t)

        # optional output planes: allocated only when asked for (0.5 GB each at 1.2e6 frequencies)
        self.d_line = c.empty((self.n_depth, self.count)) if keep_line else None
        self.d_total = c.empty((self.n_depth, self.count)) if keep_total else None
        self._flux_tensor = flux_out
        self.d_F = None if flux_out is not None else c.empty((self.n_depth, self.count))
        self.d_evals = c.zeros((1,), np.int64) if track_evaluations else None
        self._keep_line = keep_line  # also write the summed line opacity plane (alpha_line())
        self._keep_total = keep_total  # also write total_alphas (the reference keeps it on Opacities; the flux does not need it in HBM)
        self.count_evaluations = track_evaluations  # sum(hi - lo) per step costs a memset + copy: switch off when timing
        self.graph = None
        self.graph_classify = None
        self.graph_batch, self.batch = None, 1
        self.classify_share, self.m_max, self.m_share_out = classify_share, m_max, m_share_out
        if (classify_share is None) != (m_max is None):
            raise ValueError("classify_share and m_max go together")
        if m_max is not None and (self.linelist is not None or track_evaluations):
            raise ValueError("the two-collective mode takes dense line lists and no evaluation count")
        if m_max is not None:
            # the library writes m_max[first line ...] (or m_share_out[0 ...]) and reads all of m_max: check the buffers here, an
            # undersized one would be an out-of-bounds device access
            b, n = (int(v) for v in classify_share)
            if b < 0 or n < 0 or b + n > self.n_lines:
                raise ValueError("classify_share lies outside the line list")
            _require_f64_buffer("m_max", m_max, self.n_lines)
            if m_share_out is not None:
                _require_f64_buffer("m_share_out", m_share_out, n)
        c.call("sdx_reserve_line_workspace", self.n_depth, self.n_lines)

    # -- set-up ---------------------------------------------------------------------------------
    def _build_continuum(self, cont, nus, t):
        c = self.ctx
        s = Continuum()
        up = lambda a, dt=np.float64: self._hold(c.upload(np.ascontiguousarray(a, dtype=dt), dt))  # noqa: E731
        s.temperature = self.d_t.ptr
        if cont is None:
            return s
        lam = K.nu_to_angstrom(nus)  # tracing_nus.to(u.AA, u.spectral()) (opacities_solvers/base.py:62)
        s.lambdas = up(lam)
        s.n_table = len(cont["hminus_bf_wavelength"])
        s.table_wavelength = up(cont["hminus_bf_wavelength"])
        s.table_sigma = up(cont["hminus_bf_cross_section"])
        s.table_density = up(cont["n_hminus"])
        cutoff = (cont["ionization_energy"] - np.asarray(cont["level_excitation"])) / K.H_CGS
        s.bf_n_species = 1
        s.bf_n_levels = len(cutoff)
        s.bf_species_offsets = up([0, len(cutoff)], np.int32)
        s.bf_species_ion_number = up([0], np.int32)
        s.bf_cutoff = up(cutoff)
        s.bf_level_density = up(cont["level_density"])
        s.ff_n_species = 1
        s.ff_species_ion_number = up([1], np.int32)  # get_number_density("H_I_ff") returns ion_number + 1
        s.ff_number_density = up(np.asarray(cont["n_e"]) * np.asarray(cont["n_h2"]))  # util.py:160-164
        s.rayleigh_enabled = 0
        s.electron_density = up(cont["n_e"])
        return s

    @property
    def keep_line(self):
        return self._keep_line

    @keep_line.setter
    def keep_line(self, on):
        if on and self.d_line is None:
            self.d_line = self.ctx.empty((self.n_depth, self.count))
        self._keep_line = bool(on)

    @property
    def keep_total(self):
        return self._keep_total

    @keep_total.setter
    def keep_total(self, on):
        if on and self.d_total is None:
            self.d_total = self.ctx.empty((self.n_depth, self.count))
        self._keep_total = bool(on)

    def _hold(self, dev):
        self._keep.append(dev)
        return dev.ptr

    @property
    def flux_ptr(self):
        return ptr_of(self._flux_tensor) if self._flux_tensor is not None else self.d_F.ptr

    # -- one step: everything from resident inputs to F_nu -----------------------------------------
    def enqueue_classify(self):
        """Two-collective mode, phase 1: this rank's share of the classification stream (+ grid spacing, line ranges and continuum
        plane of the step) -> m_max[share]; the caller gathers m_max before enqueue()."""
        c = self.ctx
        b, n = self.classify_share
        # (the library writes m_max[l] for the lines l of the share: with a send buffer, entry l lands at index l - b of it)
        base = ptr_of(self.m_max) if self.m_share_out is None else ptr_of(self.m_share_out) - 8 * int(b)
        c.call("sdx_synthesize_classify_dev", self.n_depth, self.n_nu, self.d_nus.ptr, self.begin, self.count, self.n_lines, self.d_ln.ptr,
               self.d_dw.ptr, self.d_g.ptr, self.gamma_cols, self.d_a.ptr, C.byref(self.cont), int(b), int(n), base)

    def enqueue(self):
        """One fused step on the context's stream: sdx_synthesize_dev (pre-pass, line gather, total, raytrace)."""
        c = self.ctx
        if self.m_max is not None:  # phase 2 of the two-collective mode: everything behind the classification launch
            opt = _lib.SynthesisOptions()
            opt.line_m_max = ptr_of(self.m_max)
            c.call("sdx_synthesize_opt_dev", self.n_depth, self.n_nu, self.d_nus.ptr, self.begin, self.count, self.n_lines,
                   self.d_ln.ptr, self.d_dw.ptr, self.d_g.ptr, self.gamma_cols, self.d_a.ptr, C.byref(self.cont), self.n_theta,
                   self.d_t.ptr, self.d_ray.ptr, self.d_w.ptr, self.d_line.ptr if self.keep_line else None,
                   self.d_total.ptr if self.keep_total else None, self.flux_ptr, self.count, C.byref(opt), None)
            return
        if self.linelist is not None:
            c.call("sdx_synthesize_linelist_dev", self.n_depth, self.n_nu, self.d_nus.ptr, self.begin, self.count,
                   self.linelist.byref(), C.byref(self.cont), self.n_theta, self.d_t.ptr, self.d_ray.ptr, self.d_w.ptr,
                   self.d_line.ptr if self.keep_line else None, self.d_total.ptr if self.keep_total else None, self.flux_ptr, self.count,
                   ptr_of(self.d_evals) if self.count_evaluations else None)
            return
        c.call("sdx_synthesize_dev", self.n_depth, self.n_nu, self.d_nus.ptr, self.begin, self.count, self.n_lines,
               self.d_ln.ptr, self.d_dw.ptr, self.d_g.ptr, self.gamma_cols, self.d_a.ptr, C.byref(self.cont), self.n_theta,
               self.d_t.ptr, self.d_ray.ptr, self.d_w.ptr, self.d_line.ptr if self.keep_line else None,
               self.d_total.ptr if self.keep_total else None, self.flux_ptr, self.count, ptr_of(self.d_evals) if self.count_evaluations else None)

    def enqueue_unfused(self):
        """The same step through the individual entry points (what calc_alphas + raytrace issue)."""
        c = self.ctx
        nd, cnt = self.n_depth, self.count
        self.keep_line = self.keep_total = True  # this path materialises both planes
        if self.linelist is not None:
            c.call("sdx_line_opacity_linelist_dev", nd, self.n_nu, self.d_nus.ptr, self.begin, cnt, self.linelist.byref(),
                   self.d_line.ptr, cnt, 0, ptr_of(self.d_evals))
        else:
            c.call("sdx_line_opacity_dev", nd, self.n_nu, self.d_nus.ptr, self.begin, cnt, self.n_lines, self.d_ln.ptr,
                   self.d_dw.ptr, self.d_g.ptr, self.gamma_cols, self.d_a.ptr, self.d_line.ptr, cnt, 0, ptr_of(self.d_evals))
        c.call("sdx_total_alphas_dev", nd, self.n_nu, self.d_nus.ptr, self.begin, cnt, C.byref(self.cont), self.d_line.ptr, cnt,
               self.d_total.ptr, cnt)
        nus_shard = self.d_nus.ptr + 8 * self.begin
        c.call("sdx_raytrace_dev", nd, cnt, self.n_theta, nus_shard, self.d_t.ptr, self.d_ray.ptr, self.d_w.ptr,
               self.d_total.ptr, cnt, self.flux_ptr, cnt, None, 0)

    def capture(self, eager_phase2=True, batch=1):
        """Record one step into a hipGraph (after one eager step has sized the scratch).  batch > 1: ALSO a graph of `batch` consecutive
        steps (step_batch()): successive graph launches are ~8.5 us apart on this runtime whatever they hold — a tenth of a 92 us step —
        so a caller with a queue of syntheses replays several steps per launch.  Two-collective mode: two graphs, the
        classification launch and the rest — the caller's all-gather of m_max goes between step_classify() and step(); m_max must
        hold every rank's share when capture() is called (the eager pass reads it) unless eager_phase2 is False (a re-capture
        after the scratch has moved: it is large enough already, and m_max may not have been gathered yet)."""
        c = self.ctx

        def record(enqueue):
            c.call("sdx_graph_begin")
            try:
                enqueue()
            finally:
                handle = C.c_void_p()
                _lib.check(c.lib.sdx_graph_end(c.handle, C.byref(handle)))
            return handle

        if self.m_max is not None:
            self.enqueue_classify()
            if eager_phase2:
                self.enqueue()
            c.synchronize()
            self.graph_classify = record(self.enqueue_classify)
            self.graph = record(self.enqueue)
            return self
        self.enqueue()
        c.synchronize()
        self.graph = record(self.enqueue)
        if batch > 1:
            def many():
                for _ in range(batch):
                    self.enqueue()
            self.graph_batch, self.batch = record(many), int(batch)
        return self

    def step_classify(self):
        if self.graph_classify is not None:
            try:
                self.ctx.call("sdx_graph_launch", self.graph_classify)
                return
            except _lib.StaleGraphError:
                # (m_max holds the previous step's values or nothing yet: no eager phase 2 on it)
                self.close()
                self.capture(eager_phase2=False)
                self.ctx.call("sdx_graph_launch", self.graph_classify)
                return
        self.enqueue_classify()

    def step(self):
        if self.graph is not None:
            try:
                self.ctx.call("sdx_graph_launch", self.graph)
            except _lib.StaleGraphError:
                # another, larger synthesis on this context made the library reallocate its scratch: the captured pointers
                # are dead.  Capture again (the scratch is large enough for both now) and replay.
                self.close()
                self.capture()
                self.ctx.call("sdx_graph_launch", self.graph)
        else:
            self.enqueue()

    def step_batch(self):
        """`self.batch` consecutive steps as ONE graph launch (capture(batch=n)); -> the number of steps enqueued."""
        if self.graph_batch is None:
            self.step()
            return 1
        try:
            self.ctx.call("sdx_graph_launch", self.graph_batch)
        except _lib.StaleGraphError:
            n = self.batch
            self.close()
            self.capture(batch=n)
            self.ctx.call("sdx_graph_launch", self.graph_batch)
        return self.batch

    def synchronize(self):
        self.ctx.synchronize()

    # -- results ----------------------------------------------------------------------------------
    def F_nu(self):
        if self._flux_tensor is not None:
            self.ctx.synchronize()
            return self._flux_tensor.cpu().numpy()
        return self.d_F.numpy()

    def total_alphas(self):
        if not self.keep_total:
            raise RuntimeError("total_alphas was not kept: construct the synthesizer with keep_total=True")
        return self.d_total.numpy()

    def alpha_line(self):
        if not self.keep_line:
            raise RuntimeError("alpha_line was not kept: construct the synthesizer with keep_line=True")
        return self.d_line.numpy()

    def evaluations(self):
        return int(self.d_evals.numpy()[0]) if self.d_evals is not None else None

    def algorithmic_bytes(self):
        """line list, grid read once; total_alphas and F_nu written once (this shard's columns)."""
        if self.linelist is not None:
            lines = self.n_lines * self.linelist.host.bytes_per_line() + 8 * self.linelist.host.pop.size
        else:
            lines = 8 * self.n_lines * (1 + 2 * self.n_depth + self.gamma_cols)
        return lines + 8 * self.n_nu + 16 * self.n_depth * self.count

    def close(self):
        if self.graph is not None:
            self.ctx.call("sdx_graph_destroy", self.graph)
            self.graph = None
        if self.graph_classify is not None:
            self.ctx.call("sdx_graph_destroy", self.graph_classify)
            self.graph_classify = None
        if self.graph_batch is not None:
            self.ctx.call("sdx_graph_destroy", self.graph_batch)
            self.graph_batch, self.batch = None, 1


class SynthesisPool:
    """Several independent syntheses in flight on one GPU — a grid of stars, abundances or line lists.  Members are
    dealt round-robin onto `n_streams` contexts (each its own HIP stream and scratch), so kernels of different members
    overlap on the device: the formal solution of one fills the issue slots the line kernel of another leaves idle
    (S-c2: about 4.7e9 spectral points/s with two in flight against 3.8e9 one after the other).  Every member computes exactly
    what it would alone."""

    def __init__(self, device=None, n_streams=2):
        import os

        dev = int(os.environ.get("STARDIS_AMD_DEVICE", os.environ.get("LOCAL_RANK", "0"))) if device is None else int(device)
        self.contexts = [_lib.Context(dev) for _ in range(max(1, int(n_streams)))]
        self.members = []

    def add(self, *args, capture=True, **kwargs):
        """SpectralSynthesizer(*args, **kwargs) on the next context; captured into a hipGraph unless capture=False."""
        kwargs["ctx"] = self.contexts[len(self.members) % len(self.contexts)]
        syn = SpectralSynthesizer(*args, **kwargs)
        if capture:
            syn.count_evaluations = False
            syn.capture()
        self.members.append(syn)
        return syn

    def step(self):
        """Enqueue one step of every member (returns at once; members on different contexts run concurrently)."""
        for syn in self.members:
            syn.step()

    def synchronize(self):
        for c in self.contexts:
            c.synchronize()

    def fluxes(self):
        self.synchronize()
        return [syn.F_nu() for syn in self.members]

    def close(self):
        for syn in self.members:
            syn.close()
        self.members = []
