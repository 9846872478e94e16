"""One process, several GPUs: the C library's own multi-device entry point (include/stardis_hip.h, sdx_group_*).

`DeviceGroup` is a thin ctypes wrapper of `sdx_group_create` / `sdx_synthesize_sharded_f64`: host (numpy) arrays in, the
columns of the grid split over the group's devices, the line list replicated, ONE RCCL all-gather of the emergent-flux shards
inside the library (the frequency loop of the reference is a prange,
radiation_field/radiation_field_solvers/base.py:200).  `stardis_amd.parallel` is the other way to shard — one process per
GPU with torch.distributed, which bench.py uses; both run the same kernels on the same (global-grid) window rule.
"""
import ctypes as C

import numpy as np

from . import _lib
from . import constants as K
from ._lib import Continuum


def _group_error():
    lib = _lib.load()
    code, msg = lib.sdx_last_error_code(), lib.sdx_last_error_string().decode()
    try:
        _lib.check(code if code else -2)
    except Exception as exc:  # re-raise with the library's text
        raise type(exc)(msg) from None


class DeviceGroup:
    def __init__(self, n_gpus=None, devices=None):
        """n_gpus devices (default: every visible one), or an explicit device list."""
        self.lib = _lib.load()
        if devices is not None:
            n_gpus = len(devices)
        if n_gpus is None:
            n_gpus = self.lib.sdx_device_count()
        arr = (C.c_int * n_gpus)(*devices) if devices is not None else None
        self.handle = self.lib.sdx_group_create(int(n_gpus), arr)
        if not self.handle:
            _group_error()
        self.n = int(n_gpus)

    def close(self):
        if getattr(self, "handle", None):
            self.lib.sdx_group_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_option(self, name, value):
        for r in range(self.n):
            _lib.check(self.lib.sdx_set_int_option(self.lib.sdx_group_context(self.handle, r), name.encode(), int(value)))

    def last_gather(self):
        ranks, nbytes, ver = C.c_int(), C.c_int64(), C.c_int()
        _lib.check(self.lib.sdx_group_last_gather(self.handle, C.byref(ranks), C.byref(nbytes), C.byref(ver)))
        return {"backend": "rccl" if ver.value else "loopback-test-hook", "ranks": ranks.value, "bytes_per_rank": nbytes.value,
                "rccl_version": ver.value}

    def synthesize(self, nus, temperatures, dist, thetas, theta_weights, lines, continuum, shards=None, want_planes=False,
                   want_evaluations=False):
        """-> dict(emergent_flux (N_nu,), F_nu (N_d, N_nu), [alpha_line, total_alphas], [evaluations]).

        Arguments as stardis_amd.engine.SpectralSynthesizer: `lines` is the reference layout dict (line_nus ascending,
        doppler_widths, gammas, alphas), `continuum` the dict of synth.synth_continuum_state.  shards: optional list of
        (begin, count) per device (stardis_amd.parallel.balanced_shards); default equal blocks."""
        f8 = np.float64
        nus = np.ascontiguousarray(nus, dtype=f8)
        t = np.ascontiguousarray(temperatures, dtype=f8).reshape(-1)
        nd = t.size
        thetas = np.asarray(thetas, dtype=f8)
        ray = np.ascontiguousarray(np.asarray(dist, dtype=f8).reshape(-1, 1) / np.cos(thetas))  # radiation_field_solvers/base.py:302-305
        w = np.ascontiguousarray(theta_weights, dtype=f8)
        ln = np.ascontiguousarray(lines["line_nus"], dtype=f8)
        dw = np.ascontiguousarray(lines["doppler_widths"], dtype=f8)
        al = np.ascontiguousarray(lines["alphas"], dtype=f8)
        g = np.ascontiguousarray(lines["gammas"], dtype=f8).reshape(ln.size, -1) if ln.size else np.zeros((0, 1))
        if ln.size > 1 and np.any(ln[1:] < ln[:-1]):
            # the library takes the list in ascending frequency (what calc_alpha_line_at_nu passes, base.py:392-397) and refuses anything
            # else; like the other Python front-ends this one sorts (stably) instead — calc_alan_entries accepts any order
            order = np.argsort(ln, kind="stable")
            ln, g, dw, al = (np.ascontiguousarray(a[order]) for a in (ln, g, dw, al))
        keep = []
        cont = host_continuum(continuum, nus, t, keep)
        begins = None
        if shards is not None:
            b = np.ascontiguousarray([s[0] for s in shards] + [shards[-1][0] + shards[-1][1]], dtype=np.int64)
            begins = b.ctypes.data
            keep.append(b)
        F = np.empty((nd, nus.size))
        flux = np.empty(nus.size)
        line = np.empty((nd, nus.size)) if want_planes else None
        total = np.empty((nd, nus.size)) if want_planes else None
        ev = C.c_int64(0)
        p = lambda a: None if a is None else a.ctypes.data  # noqa: E731
        _lib.check(self.lib.sdx_synthesize_sharded_f64(
            self.handle, nd, nus.size, p(nus), ln.size, p(ln), p(dw), p(g), g.shape[1], p(al), C.byref(cont), thetas.size, p(t), p(ray), p(w),
            begins, p(line), p(total), p(F), p(flux), C.cast(C.byref(ev), C.c_void_p) if want_evaluations else None))
        out = {"emergent_flux": flux, "F_nu": F}
        if want_planes:
            out.update(alpha_line=line, total_alphas=total)
        if want_evaluations:
            out["evaluations"] = ev.value
        return out


def host_continuum(cont, nus, temperatures, keep):
    """struct sdx_continuum with HOST pointers (what the *_f64 entry points take) from the dict of
    synth.synth_continuum_state; the arrays are appended to `keep` so that they outlive the call."""
    s = Continuum()

    def hold(a, dt=np.float64):
        a = np.ascontiguousarray(a, dtype=dt)
        keep.append(a)
        return a.ctypes.data

    s.temperature = hold(temperatures)
    if cont is None:
        return s
    s.lambdas = hold(K.nu_to_angstrom(nus))  # tracing_nus.to(u.AA, u.spectral()) (opacities_solvers/base.py:62)
    s.n_table = len(cont["hminus_bf_wavelength"])
    s.table_wavelength = hold(cont["hminus_bf_wavelength"])
    s.table_sigma = hold(cont["hminus_bf_cross_section"])
    s.table_density = hold(cont["n_hminus"])
    cutoff = (cont["ionization_energy"] - np.asarray(cont["level_excitation"])) / K.H_CGS
    s.bf_n_species = 1
    s.bf_n_levels = len(cutoff)
    s.bf_species_offsets = hold([0, len(cutoff)], np.int32)
    s.bf_species_ion_number = hold([0], np.int32)
    s.bf_cutoff = hold(cutoff)
    s.bf_level_density = hold(cont["level_density"])
    s.ff_n_species = 1
    s.ff_species_ion_number = hold([1], np.int32)  # get_number_density("H_I_ff") returns ion_number + 1
    s.ff_number_density = hold(np.asarray(cont["n_e"]) * np.asarray(cont["n_h2"]))  # util.py:160-164
    s.rayleigh_enabled = 0
    s.electron_density = hold(cont["n_e"])
    return s
