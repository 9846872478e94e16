"""Line lists as per-line scalars + per-depth state: the input of the on-device line-parameter generation
(include/stardis_hip.h `sdx_linelist`).

The reference tabulates three dense (N_l, N_d) arrays on the host before its line kernel runs — alpha_line
(plasma/base.py:178-455, plasma/molecules.py:192-440), gammas and doppler_widths (opacities_solvers/broadening.py:
659-821, :1009-1085).  A `LineList` carries what those tables are computed FROM (about 80 B per line), and the
pre-pass of the line kernel evaluates the three values per (line, depth) itself.  Nothing here computes an
opacity: the arithmetic is in stardis_amd/csrc/sdx_broadening.h.
"""
import ctypes as C

import numpy as np

from . import constants as K
from ._lib import LineListStruct, default_context, plain

GAMMA_CLASSIC, GAMMA_VALD, GAMMA_RADIATION_ONLY, GAMMA_ZERO = range(4)

_PER_LINE_F8 = ("nu", "e_low_ev", "g_lo", "strength", "mass", "ionization_energy", "upper_energy", "lower_energy", "A_ul", "stark", "waals")
_PER_DEPTH_F8 = ("temperature", "electron_density", "h_density")
_PER_LINE_I4 = ("pop_row", "atomic_number", "ion_number")


def broadening_flags(linear_stark=True, quadratic_stark=True, van_der_waals=True, radiation=True):
    return (1 if linear_stark else 0) | (2 if quadratic_stark else 0) | (4 if van_der_waals else 0) | (8 if radiation else 0)


class LineList:
    """Host-side description.  Lines must be sorted by ascending `nu` and restricted to the tracing grid's range
    (opacities_solvers/base.py:392-395).  `pop` is number density / partition function per species row, (rows, N_d);
    `pop_row[l]` names the row of line l.  `ion_number` is the charge seen by the outer electron (ion_number + 1)."""

    def __init__(self, nu, e_low_ev, strength, pop_row, pop, mass, temperature, g_lo=None, microturbulence=0.0,
                 gamma_mode=GAMMA_ZERO, flags=15, atomic_number=None, ion_number=None, ionization_energy=None,
                 upper_energy=None, lower_energy=None, A_ul=None, stark=None, waals=None, electron_density=None,
                 h_density=None, alpha_coefficient=K.ALPHA_COEFFICIENT):
        f8 = lambda a: None if a is None else np.ascontiguousarray(plain(a), dtype=np.float64).reshape(-1)  # noqa: E731
        i4 = lambda a: None if a is None else np.ascontiguousarray(plain(a), dtype=np.int32).reshape(-1)  # noqa: E731
        self.nu = f8(nu)
        self.n_lines = self.nu.size
        self.e_low_ev, self.strength, self.g_lo, self.mass = f8(e_low_ev), f8(strength), f8(g_lo), f8(mass)
        self.pop_row = i4(pop_row)
        self.temperature = f8(temperature)
        self.n_depth = self.temperature.size
        self.pop = np.ascontiguousarray(plain(pop), dtype=np.float64).reshape(-1, self.n_depth)
        if self.n_lines and (self.pop_row.min() < 0 or self.pop_row.max() >= self.pop.shape[0]):
            raise ValueError("pop_row outside the population table")
        self.microturbulence = float(microturbulence)
        self.gamma_mode, self.flags = int(gamma_mode), int(flags)
        self.atomic_number, self.ion_number = i4(atomic_number), i4(ion_number)
        self.ionization_energy, self.upper_energy, self.lower_energy = f8(ionization_energy), f8(upper_energy), f8(lower_energy)
        self.A_ul, self.stark, self.waals = f8(A_ul), f8(stark), f8(waals)
        self.electron_density, self.h_density = f8(electron_density), f8(h_density)
        self.alpha_coefficient = float(alpha_coefficient)
        for name in _PER_LINE_F8 + _PER_LINE_I4:
            a = getattr(self, name)
            if a is not None and a.size != self.n_lines:
                raise ValueError(f"{name}: expected {self.n_lines} per-line values, got {a.size}")
        for name in _PER_DEPTH_F8:
            a = getattr(self, name)
            if a is not None and a.size != self.n_depth:
                raise ValueError(f"{name}: expected {self.n_depth} per-depth values, got {a.size}")
        if self.n_lines and np.any(self.mass <= 0):
            raise ZeroDivisionError("float division by zero")  # a zero Doppler width (voigt.py:148)

    def check_sorted(self):
        if self.n_lines > 1 and np.any(np.diff(self.nu) < 0):
            raise ValueError("line list must be sorted by ascending nu (opacities_solvers/base.py:392)")

    @property
    def gamma_cols(self):
        return 1 if self.gamma_mode >= GAMMA_RADIATION_ONLY else self.n_depth

    def bytes_per_line(self):
        return sum(getattr(self, n).itemsize for n in _PER_LINE_F8 + _PER_LINE_I4 if getattr(self, n) is not None)

    def upload(self, ctx=None):
        return DeviceLineList(self, ctx or default_context())


class DeviceLineList:
    """The same description resident in HBM, with the C struct that points at it."""

    def __init__(self, host, ctx):
        self.host, self.ctx = host, ctx
        self.n_lines, self.n_depth, self.gamma_cols = host.n_lines, host.n_depth, host.gamma_cols
        self._keep = {}
        s = LineListStruct()
        s.n_lines = host.n_lines
        for name in _PER_LINE_F8 + _PER_DEPTH_F8:
            a = getattr(host, name)
            if a is not None:
                self._keep[name] = ctx.upload(a)
                setattr(s, name, self._keep[name].ptr)
        for name in _PER_LINE_I4:
            a = getattr(host, name)
            if a is not None:
                self._keep[name] = ctx.upload(a, np.int32)
                setattr(s, name, self._keep[name].ptr)
        self._keep["pop"] = ctx.upload(host.pop)
        s.pop = self._keep["pop"].ptr
        s.n_pop_rows = host.pop.shape[0]
        s.alpha_coefficient = host.alpha_coefficient
        s.microturbulence = host.microturbulence
        s.gamma_mode, s.broadening_flags = host.gamma_mode, host.flags
        self.struct = s

    @property
    def nu_ptr(self):
        return self._keep["nu"].ptr

    def byref(self):
        return C.byref(self.struct)


def line_params(linelist, ctx=None, alphas=True, gammas=True, doppler_widths=True):
    """The reference's dense tables -> (alphas (N_l, N_d), gammas (N_l, N_d | 1), doppler_widths (N_l, N_d)) as numpy
    arrays (None for the ones not asked for), computed by sdx_line_params_dev."""
    ctx = ctx or default_context()
    dev = linelist if isinstance(linelist, DeviceLineList) else linelist.upload(ctx)
    nl, nd = dev.n_lines, dev.n_depth
    a = ctx.empty((nl, nd)) if alphas else None
    g = ctx.empty((nl, dev.gamma_cols)) if gammas else None
    d = ctx.empty((nl, nd)) if doppler_widths else None
    ctx.call("sdx_line_params_dev", nd, dev.byref(), a.ptr if a else None, g.ptr if g else None, d.ptr if d else None)
    return tuple(None if x is None else x.numpy() for x in (a, g, d))


def line_opacity(tracing_nus, linelist, ctx=None, return_evaluations=False):
    """calc_alan_entries (opacities_solvers/base.py:487-592) on a line list whose parameters are generated in the
    pre-pass: -> alpha_line_at_nu (N_d, N_nu)."""
    ctx = ctx or default_context()
    nus = np.ascontiguousarray(plain(tracing_nus), dtype=np.float64).reshape(-1)
    if np.any(np.diff(nus) >= 0):
        raise ValueError("tracing frequencies must be strictly descending (stardis/base.py:34)")
    (linelist.host if isinstance(linelist, DeviceLineList) else linelist).check_sorted()
    dev = linelist if isinstance(linelist, DeviceLineList) else linelist.upload(ctx)
    d_nus = ctx.upload(nus)
    out = ctx.empty((dev.n_depth, nus.size))
    ev = ctx.zeros((1,), np.int64)
    ctx.call("sdx_line_opacity_linelist_dev", dev.n_depth, nus.size, d_nus.ptr, 0, nus.size, dev.byref(), out.ptr, nus.size, 0, ev.ptr)
    res = out.numpy()
    return (res, int(ev.numpy()[0])) if return_evaluations else res
