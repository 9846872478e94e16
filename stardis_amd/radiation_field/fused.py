"""create_stellar_radiation_field in ONE device pass: what stardis/radiation_field/base.py:71-117 computes, through
sdx_synthesize_dev instead of one kernel + one (N_d, N_nu) download per opacity source.

The general mirror (calc_alphas, raytrace) forms every entry of `opacities_dict` as a host array because the reference does;
a caller of run_stardis reads `F_nu` and, now and then, one of those entries.  Here the fused step (pre-pass + continuum +
line opacity + formal solution, stardis_amd/csrc) runs on inputs uploaded in ONE staging copy, only `F_nu` comes back, and the
dictionary entries are produced the first time somebody reads them — from the device twins the step left behind
(`alpha_line_at_nu`, `total_alphas`, the broadening tables) or by the per-source entry point the general mirror would have
called (`alpha_bf`, `alpha_ff`, ...: same device functions, bit-identical planes).  Keys, insertion order, shapes, the scalar 0
of disabled sources and `F_nu` accumulation semantics are the reference's (opacities_solvers/base.py:655-738,
radiation_field_solvers/base.py:324-338).

Molecular lines (`include_molecules`, :444-484, :716-736) are a second list whose plane is formed first (sdx_line_opacity_dev /
_linelist_dev, the calls the general path makes) and added by the step after the atomic one; spherical models
(radiation_field_solvers/base.py:141-198, :296-300, :340-344) hand the step the chord table and the inward sweep; line lists
without a dense alpha table go up as per-line scalars and the pre-pass generates alpha, gamma and the Doppler width (f1) — all
through sdx_synthesize_opt_dev, all bit-identical to the general path (tests/test_gpu_round4.py).

`try_fused` returns None for configurations the fused step does not cover (more than four tabulated sources, more than 64
angles, frequencies the Rayleigh cut-off would clip, NaNs in a line table, a zero Doppler width); the caller then takes the
general path.
"""
import ctypes as C
from pathlib import Path

import numpy as np

from stardis_amd import _lib, ops
from stardis_amd import constants as K
from stardis_amd._lib import Continuum, LineListStruct, SynthesisOptions, default_context, plain
from stardis_amd.radiation_field.opacities import Opacities
from stardis_amd.radiation_field.opacities.opacities_solvers import base as B
from stardis_amd.radiation_field.opacities.opacities_solvers.broadening import _microturbulence_cgs, _switches
from stardis_amd.radiation_field.opacities.opacities_solvers.util import get_number_density, read_table, sigma_file_device
from stardis_amd.radiation_field.radiation_field_solvers.base import _source_plane, calculate_spherical_ray

F8 = np.float64
RAYLEIGH_CUTOFF = 2.3e15  # opacities_solvers/base.py:99

# What this module derives from the plasma's pandas objects (sorted line tables, level tables, density vectors) is kept per
# OBJECT and VERIFIED BY CONTENT: an entry is found by the identity of the frames / series it was read from (a strong
# reference is held, so an id cannot be recycled) and served only if a 128-bit digest of the values it was derived from
# (_witness: xxh3 over the objects' value blocks, ~40 us per MB) still matches — a table edited in place between two calls
# is derived again, as the reference (radiation_field/base.py:71-117 recomputes everything per call) would see the edit.
# Cached values are private copies (nothing in them aliases the plasma's memory).  A handful of entries, least recently used
# first out.  CACHE = False derives everything on every call.
CACHE = True
_MEMO = {}
_MEMO_MAX = 16
MEMO_MAX_BYTES = 2 << 30  # ... and at most this much derived data (numpy arrays in the cached values; the newest entry always stays)

try:
    import xxhash as _xx

    def _digest(buf):
        return _xx.xxh3_128_digest(buf)
except ImportError:  # (no xxhash: correct, ~1 GB/s instead of ~20)
    import hashlib as _hl

    def _digest(buf):
        return _hl.blake2b(buf, digest_size=16).digest()


def clear_cache():
    _MEMO.clear()


class _Same:
    """Compares equal only to a wrapper of the very same object: the immutable parts of a pandas object (its Index objects,
    which pandas replaces rather than edits) inside a witness tuple."""

    __slots__ = ("obj",)

    def __init__(self, obj):
        self.obj = obj

    def __eq__(self, other):
        return type(other) is _Same and other.obj is self.obj

    __hash__ = None


class _Never:
    """A witness that matches nothing (an object whose content cannot be read as arrays): derived again on every call."""

    def __eq__(self, other):
        return False

    __hash__ = None


def _hash_array(a):
    if not isinstance(a, np.ndarray):
        a = np.asarray(a)  # (pandas extension arrays)
    if a.dtype.kind == "O":  # strings / mixed objects: pandas' value hash per element (a buffer of pointers says nothing)
        import pandas as pd

        return (a.shape, "O", _digest(np.ascontiguousarray(pd.util.hash_array(a.reshape(-1), categorize=False))))
    if a.flags.c_contiguous:
        d = _digest(a)
    elif a.flags.f_contiguous:  # a DataFrame block made from a row-major table
        d = _digest(a.T)
    else:
        d = _digest(np.ascontiguousarray(a))
    return (a.shape, a.dtype.str, d)


def _witness(obj):
    """What a cached derivation of `obj` is checked against: a digest of every value it holds (+ the identity of its axes).
    ~1 us for a per-depth vector, ~40 us per MB of table."""
    if isinstance(obj, np.ndarray):
        return _hash_array(obj)
    kind = type(obj).__name__
    try:
        if kind in ("DataFrame", "Series"):
            mgr = getattr(obj, "_mgr", None)
            arrays = getattr(mgr, "arrays", None)
            if arrays is None:
                arrays = (obj.to_numpy(),)
            parts = [_Same(obj.index)]
            if kind == "DataFrame":
                parts.append(_Same(obj.columns))
                for name in ("blknos", "blklocs"):  # which column sits where in which block
                    loc = getattr(mgr, name, None)
                    if loc is not None:
                        parts.append(_hash_array(np.asarray(loc)))
            parts.extend(_hash_array(a) for a in arrays)
            return tuple(parts)
        if hasattr(obj, "is_unique") and hasattr(obj, "get_indexer"):  # a pandas Index: immutable
            return (_Same(obj),)
    except Exception:  # noqa: BLE001  (anything exotic: not cacheable)
        pass
    return _Never()


_SEEN = None  # {id(object): witness} for the duration of ONE try_fused call (several derivations read the same tables), else None


def _witness_once(obj):
    if _SEEN is None:
        return _witness(obj)
    w = _SEEN.get(id(obj))
    if w is None:
        w = _SEEN[id(obj)] = _witness(obj)
    return w


def _memo(tag, objects, extra, build, private=()):
    """build() for these source objects, or the value kept from an earlier call if the objects are the same AND hold the same
    values.  `private`: positions in `objects` of values this module made itself (cached derivations handed on): identity only."""
    if not CACHE:
        return build()
    key = (tag, tuple(id(o) for o in objects), extra)
    now = tuple(_Same(o) if i in private else _witness_once(o) for i, o in enumerate(objects))
    hit = _MEMO.get(key)
    if hit is not None and hit[3] == now:
        _MEMO[key] = _MEMO.pop(key)  # most recently used last
        return hit[1]
    _MEMO.pop(key, None)
    value = build()
    _MEMO[key] = (objects, value, _nbytes(value), now)
    while len(_MEMO) > _MEMO_MAX or (len(_MEMO) > 1 and sum(e[2] for e in _MEMO.values()) > MEMO_MAX_BYTES):
        _MEMO.pop(next(iter(_MEMO)))
    return value


def _nbytes(value, depth=0):
    """numpy bytes reachable from a cached value (tuples, lists, dicts, objects with a __dict__), for the cache's byte bound"""
    if isinstance(value, np.ndarray):
        return value.nbytes
    if depth > 4:
        return 0
    if isinstance(value, dict):
        return sum(_nbytes(v, depth + 1) for v in value.values())
    if isinstance(value, (tuple, list)):
        return sum(_nbytes(v, depth + 1) for v in value)
    if hasattr(value, "__dict__"):
        return sum(_nbytes(v, depth + 1) for v in vars(value).values())
    return 0


_QUADRATURE = {}


def _leggauss(n):
    """np.polynomial.legendre.leggauss(n) as RadiationField.__init__ maps it (radiation_field/base.py:61-63), once per n."""
    q = _QUADRATURE.get(n)
    if q is None:
        nodes, weights = np.polynomial.legendre.leggauss(n)
        q = _QUADRATURE[n] = ((nodes / 2) + 0.5 * np.pi / 2, weights * np.pi / 2)
    return q[0].copy(), q[1].copy()


class _Thunk:
    """A dictionary entry that has not been asked for yet."""

    __slots__ = ("make",)

    def __init__(self, make):
        self.make = make


class LazyOpacitiesDict(dict):
    """`opacities_dict` whose array entries materialise on first read.  Every read path goes through __getitem__ (a
    trivially overridden __iter__ keeps dict(d) / d.copy() off CPython's raw-value fast path)."""

    def __getitem__(self, key):
        value = dict.__getitem__(self, key)
        if isinstance(value, _Thunk):
            value = value.make()
            dict.__setitem__(self, key, value)
        return value

    def __iter__(self):
        return dict.__iter__(self)

    def get(self, key, default=None):
        return self[key] if key in self else default

    def _all(self):
        for key in dict.keys(self):
            self[key]

    def items(self):
        self._all()
        return dict.items(self)

    def values(self):
        self._all()
        return dict.values(self)

    def copy(self):
        self._all()
        return dict(dict.items(self))

    def pop(self, key, *default):
        if key in self:
            self[key]
        return dict.pop(self, key, *default)


# Device memory a fused field keeps for its lazy entries (total, line plane, broadening tables, staged inputs, tracked
# intensities: up to ~1 GB at 1e6 lines) is BOUNDED per process: fields are remembered weakly in creation order, and when the
# bytes they hold exceed DEVICE_BUDGET_BYTES the oldest ones are released — their entries materialise on the host first (what
# the reference would hold anyway), then the device twins go back to the context's pool.  A caller that keeps many outputs
# (model grids, fits) therefore runs out of nothing the reference would not run out of.  release_device() does it by hand.
DEVICE_BUDGET_BYTES = 8 << 30
_LIVE = []  # [(weakref to the field, bytes)], oldest first


def _enforce_budget(new_bytes):
    import weakref  # noqa: F401

    alive = [(r, b) for r, b in _LIVE if r() is not None and getattr(r().opacities, "_device_bytes", 0)]
    _LIVE[:] = alive
    total = sum(b for _, b in alive) + new_bytes
    while alive and total > DEVICE_BUDGET_BYTES:
        ref, b = alive.pop(0)
        field = ref()
        if field is not None:
            release_device(field)
        total -= b
    _LIVE[:] = alive


def release_device(field, materialize=True):
    """Drop the device memory a fused RadiationField holds.  materialize=True (default) first forms every lazy entry on the host
    — opacities_dict, total_alphas, I_nus — so that nothing is lost; False discards what has not been read (the entries then
    read as the general path would recompute them is NOT attempted: they raise)."""
    opac = field.opacities
    if not isinstance(opac, FusedOpacities):
        return
    if materialize:
        opac.opacities_dict._all()
        opac.total_alphas  # noqa: B018
        if getattr(field, "_I_dev", None) is not None:
            field.I_nus  # noqa: B018
    else:
        def gone():
            raise RuntimeError("this entry was released with release_device(materialize=False) before it was read")
        for key in list(dict.keys(opac.opacities_dict)):
            if isinstance(dict.__getitem__(opac.opacities_dict, key), _Thunk):
                dict.__setitem__(opac.opacities_dict, key, _Thunk(gone))
        opac._discarded = opac._total_host is None
    opac._total_twin = None
    opac._total_dev = None
    opac._resident = {}
    opac._device_bytes = 0
    if getattr(field, "_I_dev", None) is not None:
        field._I_dev = None
    field._device_blob = None


class FusedOpacities(Opacities):
    """Opacities whose `total_alphas` lives on the device until read (opacities/base.py:4-28 keeps a host array)."""

    def __init__(self, shape):
        self.opacities_dict = LazyOpacitiesDict()
        self._resident = {}
        self._total_dev = None
        self._total_host = None
        self._total_twin = None  # device plane written by the fused step
        self._shape = shape
        self._device_bytes = 0

    @property
    def total_alphas(self):
        if self._total_host is None and getattr(self, "_discarded", False):
            raise RuntimeError("total_alphas was released with release_device(materialize=False) before it was read")
        if self._total_host is None:
            self._total_host = self._total_twin.numpy() if self._total_twin is not None else np.zeros(self._shape)  # (no twin: nothing was computed — np.zeros like the reference's constructor)
            if self._total_twin is not None:
                self._total_dev = (self._total_host.copy(), self._total_twin)
        return self._total_host

    @total_alphas.setter
    def total_alphas(self, value):
        self._total_host = value

    def total_alphas_device(self, ctx):
        if self._total_host is None and self._total_twin is not None:
            return self._total_twin
        return super().total_alphas_device(ctx)


_TRACKED_CLASSES = {}


def _tracked_class(field_cls):
    """field_cls with `I_nus` (radiation_field/base.py:64-68) materialising from the device on first read: the array is
    N_theta times the size of F_nu (68 MB at 7634 frequencies, 20 angles), and most callers never look at it."""
    cls = _TRACKED_CLASSES.get(field_cls)
    if cls is None:
        def get(self):
            if self._I_host is None:
                self._I_host = self._I_dev.numpy()
            return self._I_host

        def put(self, value):
            self._I_host = value

        cls = _TRACKED_CLASSES[field_cls] = type("Fused" + field_cls.__name__, (field_cls,), {"I_nus": property(get, put), "_I_host": None, "_I_dev": None})
    return cls


def _packed_upload(ctx, arrays):
    """One staging copy for many small arrays: -> (DeviceArray holding them all, [device address of each]).  The arrays are
    packed straight into page-locked memory and go up by DMA from there (asynchronous: the call's final download, which
    synchronises, comes before the staging block can be handed out again)."""
    offs, total = [], 0
    for a in arrays:
        offs.append(total)
        total += (a.nbytes + 255) & ~255
    total = max(total, 256)
    blob = ctx.pinned.empty(total, np.uint8)
    pinned = blob is not None
    if not pinned:
        blob = np.empty(total, dtype=np.uint8)
    for a, o in zip(arrays, offs):
        blob[o:o + a.nbytes] = a.reshape(-1).view(np.uint8)
    if pinned:
        dev = ctx.empty(total, np.uint8)
        ctx.call("sdx_memcpy_h2d_pinned", dev.ptr, blob.ctypes.data, total)
        dev._staging = blob  # stays out of the pool while the device copy may still be reading it
    else:
        dev = ctx.upload(blob, np.uint8)
    return dev, [dev.ptr + o for o in offs]


def _sorted_in_grid(nu, lo, hi):
    """Row order of `_in_grid` (opacities_solvers/base.py:392-397): table.sort_values("nu") then nu.between(min, max).
    pandas sorts a single float column with ndarray.argsort(kind="quicksort"); None when NaNs would need its special casing."""
    if np.isnan(nu).any():
        return None
    order = np.argsort(nu, kind="quicksort")
    s = nu[order]
    return order[(s >= lo) & (s <= hi)]


def _mass_of(nuclide_masses, atomic_number):
    idx = nuclide_masses.index.get_indexer(atomic_number)
    if (idx < 0).any():
        raise KeyError(f"no nuclide mass for atomic numbers {sorted(set(np.asarray(atomic_number)[idx < 0].tolist()))}")
    return np.asarray(nuclide_masses.to_numpy(), dtype=F8)[idx]


def _bf_arrays(stellar_plasma, species):
    p = stellar_plasma
    return _memo("bf", (p.levels, p.excitation_energy, p.level_number_density, p.ionization_data, p.ion_number_density, p.electron_densities),
                 tuple(species), lambda: _bf_arrays_build(p, species))


def _bf_arrays_build(stellar_plasma, species):
    """_bf_levels of the general mirror without a pandas look-up per level: levels of each species in plasma order."""
    levels = stellar_plasma.levels
    z_all = np.asarray(levels.get_level_values(0))
    ion_all = np.asarray(levels.get_level_values(1))
    exc, dens = stellar_plasma.excitation_energy, stellar_plasma.level_number_density
    if not (exc.index.equals(levels) and dens.index.equals(levels)):
        return None  # label look-ups needed: general path
    exc_v, dens_v = np.asarray(exc.to_numpy(), dtype=F8), np.asarray(dens.to_numpy(), dtype=F8)
    offsets, ions, cutoffs, densities = [0], [], [], []
    for spec in species:
        _, atomic_number, ion_number = get_number_density(stellar_plasma, spec + "_bf")
        e_ion = float(stellar_plasma.ionization_data.loc[(atomic_number, ion_number + 1)])
        sel = np.flatnonzero((z_all == atomic_number) & (ion_all == ion_number))
        cutoffs.append((e_ion - exc_v[sel]) / K.H_CGS)
        densities.append(dens_v[sel])
        offsets.append(offsets[-1] + sel.size)
        ions.append(ion_number)
    n_depth = dens_v.shape[1] if dens_v.ndim == 2 else 0
    return (np.asarray(offsets, dtype=np.int32), np.asarray(ions, dtype=np.int32),
            np.concatenate(cutoffs) if cutoffs else np.zeros(0), np.vstack(densities) if densities else np.zeros((0, n_depth)))


def _sorted_line_tables(lines, alpha_table, nuclide_masses, with_vald):
    """Both tables in the row order of `_in_grid` (opacities_solvers/base.py:392-397: sort_values("nu"); pandas sorts one float
    column with ndarray.argsort(kind="quicksort")), as plain arrays.  None when NaNs would need pandas' special casing."""
    nu_l = np.asarray(lines["nu"].to_numpy(), dtype=F8)
    nu_a = np.asarray(alpha_table["nu"].to_numpy(), dtype=F8)
    if np.isnan(nu_l).any() or np.isnan(nu_a).any():
        return None
    order_l, order_a = np.argsort(nu_l, kind="quicksort"), np.argsort(nu_a, kind="quicksort")
    col = lambda name, dt=F8: np.ascontiguousarray(np.asarray(lines[name].to_numpy(), dtype=dt)[order_l])  # noqa: E731  (pd.to_numeric of :411)
    out = dict(nu=nu_l[order_l], nu_alpha=nu_a[order_a], z=col("atomic_number", np.int64), ion=col("ion_number", np.int64),
               e_ion=col("ionization_energy"), e_up=col("level_energy_upper"), e_lo=col("level_energy_lower"), a_ul=col("A_ul"))
    if with_vald:
        out["stark"], out["waals"] = col("stark"), col("waals")
    alpha_cols = [c for c in alpha_table.columns if c != "nu"]
    out["alphas"] = np.ascontiguousarray(np.asarray(alpha_table[alpha_cols].to_numpy(), dtype=F8)[order_a])
    out["mass"] = _mass_of(nuclide_masses, out["z"])
    return out


def _line_arrays(stellar_plasma, stellar_model, nus, cfg):
    """calc_alpha_line_at_nu's host preparation (:362-421) as flat arrays: the selected lines in ascending frequency with
    their dense alphas and per-line broadening scalars.  None when this path does not cover the configuration."""
    vald = cfg.vald_linelist
    if vald.use_linelist:
        lines, alpha_table = stellar_plasma.lines_from_linelist, getattr(stellar_plasma, "alpha_line_from_linelist", None)
        if alpha_table is None:
            return None  # parameters generated on the device (f1): general path
    else:
        p = stellar_plasma
        lines = _memo("atomic_line_table", (p.lines, p.ionization_data, p.atomic_data.levels.energy), None, lambda: B._atomic_line_table(p))
        alpha_table = stellar_plasma.alpha_line
    vald_broadening = bool(vald.use_vald_broadening and vald.use_linelist)
    masses = stellar_model.composition.nuclide_masses
    tab = _memo("line_tables", (lines, alpha_table, masses), vald_broadening,
                lambda: _sorted_line_tables(lines, alpha_table, masses, vald_broadening),
                private=() if vald.use_linelist else (0,))  # (the joined TARDIS table is this module's own, verified above)
    if tab is None:
        return None
    # nu.between(min, max) (:393-395) on sorted columns is a slice
    lo, hi = nus.min(), nus.max()
    i0, i1 = np.searchsorted(tab["nu"], lo, "left"), np.searchsorted(tab["nu"], hi, "right")
    j0, j1 = np.searchsorted(tab["nu_alpha"], lo, "left"), np.searchsorted(tab["nu_alpha"], hi, "right")
    if i1 - i0 != j1 - j0:
        return None
    out = {k: v[i0:i1] for k, v in tab.items() if k not in ("alphas", "nu_alpha")}
    alphas = tab["alphas"][j0:j1]
    if not vald.use_vald_broadening:  # auto-ionising lines are dropped unless VALD broadening is used (:413-421)
        keep = ~(out["e_up"] > out["e_ion"])
        if not keep.all():
            out = {k: v[keep] for k, v in out.items()}
            alphas = alphas[keep]
    out["alphas"] = alphas
    out["vald_broadening"] = vald_broadening
    return out


def _depth_vectors(stellar_plasma, opacity, file_source, rayleigh_species):
    """Every per-depth vector the step reads from the plasma, as float64 arrays (get_number_density, util.py:111-166)."""
    p = stellar_plasma
    ff_species = tuple(opacity.ff.keys() if hasattr(opacity.ff, "keys") else opacity.ff)

    def build():
        ions = p.ion_number_density
        own = lambda a: np.array(plain(a), dtype=F8)  # noqa: E731  (a copy: cached values must not alias the plasma's memory)
        out = {"n_e": own(p.electron_densities).reshape(-1)}
        if file_source is not None:
            out["file"] = own(get_number_density(p, file_source)[0])
        ff_ions, ff_dens = [], []
        for spec in ff_species:
            number_density, _, ion_number = get_number_density(p, spec + "_ff")
            ff_ions.append(ion_number), ff_dens.append(own(number_density))
        out["ff_ions"], out["ff_dens"] = ff_ions, ff_dens
        out["n_h"] = own(ions.loc[1, 0])  # neutral hydrogen: van der Waals broadening and Rayleigh scattering
        if "He" in rayleigh_species:
            out["n_he"] = own(ions.loc[2, 0])
        if "H2" in rayleigh_species:
            out["n_h2"] = own(p.h2_density)
        return out

    objs = [p.ion_number_density, p.electron_densities]
    for name in ("h_minus_density", "h2_density", "h2_plus_density"):
        if getattr(p, name, None) is not None:
            objs.append(getattr(p, name))
    return _memo("depth", tuple(objs), (file_source, ff_species, tuple(rayleigh_species)), build)


def _fingerprint(a):
    a = np.ascontiguousarray(a)
    return (a.shape, hash(a.tobytes()))


def _owned(spec):
    """`spec` (a LineList) with every array its own copy: nothing in a cached list may alias the plasma's or the model's memory."""
    for name, a in list(vars(spec).items()):
        if isinstance(a, np.ndarray):  # (owndata says nothing: ascontiguousarray hands a conforming caller's array back as it is)
            setattr(spec, name, a.copy())
    return spec


def _deferred_atomic(stellar_plasma, stellar_model, nus, cfg):
    """The LineList calc_alpha_line_at_nu builds when the plasma carries no dense alpha table (base.py mirror, f1), kept per
    set of plasma objects, grid range, model temperatures and broadening configuration."""
    from stardis_amd.plasma.base import deferred_line_list

    p, vald = stellar_plasma, cfg.vald_linelist
    temps = np.asarray(plain(stellar_model.temperatures), dtype=F8)
    return _memo("deferred_atomic", (p.lines_from_linelist, p.ion_number_density, p.partition_function, p.electron_densities,
                                     stellar_model.composition.nuclide_masses),
                 (float(nus.min()), float(nus.max()), tuple(cfg.broadening), bool(vald.use_vald_broadening), _fingerprint(temps),
                  _microturbulence_cgs(stellar_model)),
                 lambda: _owned(deferred_line_list(p.lines_from_linelist, nus, stellar_model, p, cfg.broadening, vald.use_vald_broadening)))


def _deferred_molecules(stellar_plasma, stellar_model, nus, cfg):
    from stardis_amd.plasma.molecules import deferred_molecule_line_list

    p = stellar_plasma
    temps = np.asarray(plain(stellar_model.temperatures), dtype=F8)
    return _memo("deferred_molecules", (p.molecule_lines_from_linelist, p.molecule_number_density, p.molecule_partition_function, p.molecule_ion_map,
                                        stellar_model.composition.nuclide_masses),
                 (float(nus.min()), float(nus.max()), tuple(cfg.broadening), _fingerprint(temps), _microturbulence_cgs(stellar_model)),
                 lambda: _owned(deferred_molecule_line_list(p.molecule_lines_from_linelist, nus, stellar_model, p, cfg.broadening)))


def _sorted_molecule_tables(lines, alpha_table, ion_map, nuclide_masses):
    """molecule_lines_from_linelist / molecule_alpha_line_from_linelist in the row order of `_in_grid`, as plain arrays, with
    the summed mass of the two constituent nuclides (broadening.py:808-819)."""
    nu_l = np.asarray(lines["nu"].to_numpy(), dtype=F8)
    nu_a = np.asarray(alpha_table["nu"].to_numpy(), dtype=F8)
    if np.isnan(nu_l).any() or np.isnan(nu_a).any():
        return None
    order_l, order_a = np.argsort(nu_l, kind="quicksort"), np.argsort(nu_a, kind="quicksort")
    ions = ion_map.loc[lines["molecule"].to_numpy()[order_l]]
    mass = nuclide_masses.loc[ions.Ion1].values + nuclide_masses.loc[ions.Ion2].values
    alpha_cols = [c for c in alpha_table.columns if c != "nu"]
    return dict(nu=nu_l[order_l], nu_alpha=nu_a[order_a], a_ul=np.ascontiguousarray(np.asarray(lines["A_ul"].to_numpy(), dtype=F8)[order_l]),
                mass=np.asarray(mass, dtype=F8), alphas=np.ascontiguousarray(np.asarray(alpha_table[alpha_cols].to_numpy(), dtype=F8)[order_a]))


def _molecule_arrays(stellar_plasma, stellar_model, nus):
    p = stellar_plasma
    lines, alpha_table = p.molecule_lines_from_linelist, p.molecule_alpha_line_from_linelist
    masses = stellar_model.composition.nuclide_masses
    tab = _memo("molecule_tables", (lines, alpha_table, p.molecule_ion_map, masses), None,
                lambda: _sorted_molecule_tables(lines, alpha_table, p.molecule_ion_map, masses))
    if tab is None:
        return None
    lo, hi = nus.min(), nus.max()
    i0, i1 = np.searchsorted(tab["nu"], lo, "left"), np.searchsorted(tab["nu"], hi, "right")
    j0, j1 = np.searchsorted(tab["nu_alpha"], lo, "left"), np.searchsorted(tab["nu_alpha"], hi, "right")
    if i1 - i0 != j1 - j0:
        return None
    return dict(nu=tab["nu"][i0:i1], a_ul=tab["a_ul"][i0:i1], mass=tab["mass"][i0:i1], alphas=tab["alphas"][j0:j1])


_LL_F8 = ("nu", "e_low_ev", "g_lo", "strength", "mass", "ionization_energy", "upper_energy", "lower_energy", "A_ul", "stark", "waals",
          "temperature", "electron_density", "h_density")
_LL_I4 = ("pop_row", "atomic_number", "ion_number")


def _stage_linelist(spec, tag, add):
    """Queue a LineList's arrays for the staging copy."""
    for name in _LL_F8:
        if getattr(spec, name) is not None:
            add(f"{tag}_{name}", getattr(spec, name))
    for name in _LL_I4:
        if getattr(spec, name) is not None:
            add(f"{tag}_{name}", getattr(spec, name), np.int32)
    add(f"{tag}_pop", spec.pop)


def _linelist_struct(spec, tag, P):
    """struct sdx_linelist over the staged arrays (what linelist.DeviceLineList builds from separate uploads)."""
    s = LineListStruct()
    s.n_lines = spec.n_lines
    for name in _LL_F8 + _LL_I4:
        if getattr(spec, name) is not None:
            setattr(s, name, P(f"{tag}_{name}"))
    s.pop, s.n_pop_rows = P(f"{tag}_pop"), spec.pop.shape[0]
    s.alpha_coefficient, s.microturbulence = spec.alpha_coefficient, spec.microturbulence
    s.gamma_mode, s.broadening_flags = spec.gamma_mode, spec.flags
    return s


def try_fused(field_cls, tracing_nus, stellar_model, stellar_plasma, config, source_function):
    """-> RadiationField computed by one fused device pass, or None when the configuration needs the general path."""
    global _SEEN
    _SEEN = {}  # (witnesses are per call: the tables may be edited before the next one)
    try:
        return _try_fused(field_cls, tracing_nus, stellar_model, stellar_plasma, config, source_function)
    finally:
        _SEEN = None


def _try_fused(field_cls, tracing_nus, stellar_model, stellar_plasma, config, source_function):
    opacity = config.opacity
    spherical = bool(getattr(stellar_model, "spherical", False))
    tracked = bool(config.result_options.return_radiation_field)
    if int(config.no_of_thetas) > 64 or len(opacity.file) > 4:
        return None
    nus = np.ascontiguousarray(plain(tracing_nus), dtype=F8).reshape(-1)
    nd = int(stellar_model.no_of_depth_points)
    if nus.size < 2 or nd < 2 or np.any(np.diff(nus) >= 0):
        return None
    rayleigh_species = list(opacity.rayleigh)
    if rayleigh_species and nus.max() > RAYLEIGH_CUTOFF:
        return None  # the reference clips the caller's frequencies in place there (:99): general path reproduces it
    # tabulated sources (:666-677).  One 1-D table (Hminus_bf): interpolated inside the step.  A two-dimensional table
    # (Hminus_ff, H2plus_bf) or several sources: each becomes a plane on the device first — by the calls the general path makes,
    # so the same bits — and the step adds the planes in the configuration's order (sdx_continuum.file_plane).
    table = None
    tables = [(source, Path(fpath), read_table(Path(fpath), source)) for source, fpath in opacity.file.items()]
    plane_sources = tables if (len(tables) > 1 or any(t[2][0] != "1d" for t in tables)) else []
    if not plane_sources and tables:
        table = tables[0][2]
    bf_species = list(opacity.bf.keys()) if hasattr(opacity.bf, "keys") else list(opacity.bf)
    bf = _bf_arrays(stellar_plasma, bf_species)
    if bf is None or bf[2].size > 4096:
        return None
    line = line_spec = None
    positive_temps = bool(np.all(np.asarray(plain(stellar_model.temperatures), dtype=F8) > 0))
    if not opacity.line.disable:
        if opacity.line.vald_linelist.use_linelist and getattr(stellar_plasma, "alpha_line_from_linelist", None) is None:
            # no dense alpha table on the plasma: per-line scalars go up and the pre-pass generates alpha, gamma and the Doppler
            # width (f1), as calc_alpha_line_at_nu does there (its LineList, its kernel)
            try:
                line_spec = _deferred_atomic(stellar_plasma, stellar_model, nus, opacity.line)
            except (ZeroDivisionError, KeyError, AttributeError):
                return None  # the general path raises what the reference raises
            if line_spec.n_lines and (np.any(np.diff(line_spec.nu) < 0) or np.any(line_spec.nu == 0) or not positive_temps):
                return None
        else:
            line = _line_arrays(stellar_plasma, stellar_model, nus, opacity.line)
            if line is None:
                return None
            if line["nu"].size and np.any(np.diff(line["nu"]) < 0):
                return None
            # a zero Doppler width (nu = 0, or T = 0 without microturbulence) raises in the general path (voigt.py:148): leave it to it
            if np.any(line["nu"] == 0) or np.any(line["mass"] <= 0) or not positive_temps:
                return None
    # molecular lines (:716-736): a second list, its plane added after the atomic one
    mol = mol_spec = None
    molecules = bool(opacity.line.include_molecules) and not opacity.line.disable
    if molecules:
        if getattr(stellar_plasma, "molecule_alpha_line_from_linelist", None) is None:
            try:
                mol_spec = _deferred_molecules(stellar_plasma, stellar_model, nus, opacity.line)
            except (ZeroDivisionError, KeyError, AttributeError):
                return None
            if mol_spec.n_lines and (np.any(np.diff(mol_spec.nu) < 0) or np.any(mol_spec.nu == 0) or not positive_temps):
                return None
        else:
            mol = _molecule_arrays(stellar_plasma, stellar_model, nus)
            if mol is None or (mol["nu"].size and (np.any(np.diff(mol["nu"]) < 0) or np.any(mol["nu"] == 0) or np.any(mol["mass"] <= 0) or not positive_temps)):
                return None
            if mol["alphas"].shape != (mol["nu"].size, nd):
                return None
    ctx = default_context()
    temps = np.ascontiguousarray(plain(stellar_model.temperatures), dtype=F8).reshape(-1)
    file_source = tables[0][0] if table is not None else None
    dv = _depth_vectors(stellar_plasma, opacity, file_source, rayleigh_species)
    n_e = dv["n_e"]

    cls = _tracked_class(field_cls) if tracked else field_cls
    field = cls.__new__(cls)  # the attributes of RadiationField.__init__ (:38-68) without its zero-filled planes
    field.frequencies = tracing_nus
    field.source_function = source_function
    field.thetas, field.I_nus_weights = _leggauss(int(config.no_of_thetas))
    field.track_individual_intensities = tracked
    opac = FusedOpacities((nd, nus.size))
    field.opacities = opac

    # ---- everything the step reads, in one staging copy
    host, slot = [], {}

    def add(name, a, dt=F8):
        slot[name] = len(host)
        host.append(np.ascontiguousarray(a, dtype=dt).reshape(-1))

    correction = 1.0
    if spherical:  # radiation_field_solvers/base.py:296-300, :340-344
        radii = np.asarray(plain(stellar_model.geometry.r), dtype=F8)
        ray_table = calculate_spherical_ray(field.thetas, radii)
        correction = (radii[-1] / float(plain(stellar_model.geometry.reference_r))) ** 2
    else:
        dist = np.asarray(plain(stellar_model.geometry.dist_to_next_depth_point), dtype=F8)
        ray_table = dist.reshape(-1, 1) / np.cos(field.thetas)  # :302-305
    add("nus", nus)
    add("temps", temps)
    add("ray", ray_table)
    add("wts", field.I_nus_weights)
    add("lambdas", K.nu_to_angstrom(nus))
    if table is not None:
        add("tab_x", table[1]), add("tab_y", table[2]), add("tab_n", dv["file"])
    add("bf_off", bf[0], np.int32), add("bf_ion", bf[1], np.int32), add("bf_cut", bf[2]), add("bf_den", bf[3])
    ff_ions, ff_dens = dv["ff_ions"], dv["ff_dens"]
    add("ff_ion", ff_ions, np.int32), add("ff_den", np.vstack(ff_dens) if ff_dens else np.zeros((0, nd)))
    if "H" in rayleigh_species:
        add("ray_h", dv["n_h"])
    if "He" in rayleigh_species:
        add("ray_he", dv["n_he"])
    if "H2" in rayleigh_species:
        add("ray_h2", dv["n_h2"])
    if not opacity.disable_electron_scattering:
        add("n_e", n_e)
    n_lines = 0
    if line is not None:
        n_lines = line["nu"].size
        n_h = dv["n_h"]
        add("l_nu", line["nu"]), add("l_alpha", line["alphas"]), add("l_z", line["z"], np.int32), add("l_ion", line["ion"] + 1, np.int32)
        for k in ("e_ion", "e_up", "e_lo", "a_ul", "mass"):
            add("l_" + k, line[k])
        if line["vald_broadening"]:
            add("l_stark", line["stark"]), add("l_waals", line["waals"])
        add("b_ne", n_e), add("b_nh", n_h)
        if n_lines and np.any(line["alphas"].shape != (n_lines, nd)):
            return None
    elif line_spec is not None:
        n_lines = line_spec.n_lines
        if n_lines:
            _stage_linelist(line_spec, "ls", add)
    n_mol = 0
    if mol is not None:
        n_mol = mol["nu"].size
        add("m_nu", mol["nu"]), add("m_alpha", mol["alphas"]), add("m_mass", mol["mass"])
        radiation = "radiation" in opacity.line.broadening
        # gamma = A_ul as one column, or — without "radiation" — the reference's (N_l, N_d) zeros (broadening.py:799-806)
        add("m_gamma", mol["a_ul"] if radiation else np.zeros((n_mol, nd)))
    elif mol_spec is not None:
        n_mol = mol_spec.n_lines
        if n_mol:
            _stage_linelist(mol_spec, "ms", add)
    # a source function other than the Planck function is evaluated on the host, the way the reference calls it (:133), and goes up
    # with everything else
    try:
        source = _source_plane(source_function, nus, temps)
    except ValueError:
        return None  # a result that does not broadcast to (N_d, N_nu): the general path reports it
    if source is not None:
        add("source", source)
    blob, ptrs = _packed_upload(ctx, host)
    try:
        return _run_step(locals())
    except BaseException:
        # the staging block went up by an asynchronous DMA out of pooled page-locked memory: nothing may hand that block out
        # again (blob._staging returning to the pool when `blob` dies) while the copy could still be reading it
        try:
            ctx.synchronize()
        except Exception:  # noqa: BLE001
            pass
        raise


def _run_step(v):
    """The device part of try_fused (its local variables come in as a dictionary: one function would do, but the guard above
    has to cover everything from the staging upload to the final download)."""
    (ctx, host, slot, blob, ptrs, field, opac, nus, nd, temps, tables, table, plane_sources, bf, dv, n_e, line, line_spec, mol, mol_spec,
     n_lines, n_mol, opacity, config, stellar_plasma, stellar_model, tracked, spherical, correction, source, rayleigh_species, ff_ions) = (
        v[k] for k in ("ctx", "host", "slot", "blob", "ptrs", "field", "opac", "nus", "nd", "temps", "tables", "table", "plane_sources", "bf", "dv",
                       "n_e", "line", "line_spec", "mol", "mol_spec", "n_lines", "n_mol", "opacity", "config", "stellar_plasma", "stellar_model",
                       "tracked", "spherical", "correction", "source", "rayleigh_species", "ff_ions"))
    P = lambda name: ptrs[slot[name]] if name in slot else None  # noqa: E731

    file_planes = []
    for source, fpath, tab in plane_sources:
        density = np.asarray(plain(get_number_density(stellar_plasma, source)[0]), dtype=F8)
        if tab[0] == "1d":
            file_planes.append(ops.alpha_file_1d(K.nu_to_angstrom(nus), tab[1], tab[2], density, ctx=ctx))
        else:
            file_planes.append(ops.alpha_file_2d(sigma_file_device(K.nu_to_angstrom(nus), temps, fpath, source), density, ctx=ctx))

    c = Continuum()
    c.temperature = P("temps")
    c.lambdas = P("lambdas")
    c.n_file_planes = len(file_planes)
    for k, plane in enumerate(file_planes):
        c.file_plane[k] = plane.ptr
    c.file_plane_ld = nus.size
    if table is not None:
        c.n_table, c.table_wavelength, c.table_sigma, c.table_density = int(np.size(table[1])), P("tab_x"), P("tab_y"), P("tab_n")
    if bf[1].size:
        c.bf_n_species, c.bf_n_levels = int(bf[1].size), int(bf[2].size)
        c.bf_species_offsets, c.bf_species_ion_number, c.bf_cutoff, c.bf_level_density = P("bf_off"), P("bf_ion"), P("bf_cut"), P("bf_den")
        if c.bf_n_levels == 0:
            c.bf_n_species = 0
    if ff_ions:
        c.ff_n_species, c.ff_species_ion_number, c.ff_number_density = len(ff_ions), P("ff_ion"), P("ff_den")
    c.ray_n_h, c.ray_n_he, c.ray_n_h2 = P("ray_h"), P("ray_he"), P("ray_h2")
    c.rayleigh_enabled = 1 if rayleigh_species else 0
    c.electron_density = P("n_e")

    d_gamma = d_doppler = None
    if n_lines and line is not None:
        lin, quad, vdw, rad = _switches(opacity.line.broadening)
        flags = (1 if lin else 0) | (2 if quad else 0) | (4 if vdw else 0) | (8 if rad else 0)
        d_gamma, d_doppler = ctx.empty((n_lines, nd)), ctx.empty((n_lines, nd))
        common = (n_lines, nd, P("l_z"), P("l_ion"), P("l_e_ion"), P("l_e_up"), P("l_e_lo"), P("l_a_ul"))
        if line["vald_broadening"]:  # broadening.py:1009-1085
            ctx.call("sdx_calc_vald_gamma_dev", *common, P("l_stark"), P("l_waals"), P("l_mass"), P("b_ne"), P("temps"), P("b_nh"), flags, d_gamma.ptr)
        else:  # broadening.py:550-656 with the argument preparation of :706-721
            ctx.call("sdx_calc_gamma_dev", *common, P("b_ne"), P("temps"), P("b_nh"), flags, d_gamma.ptr)
        ctx.call("sdx_doppler_widths_dev", n_lines, nd, P("l_nu"), P("l_mass"), P("temps"), _microturbulence_cgs(stellar_model), d_doppler.ptr)
    # the molecular plane first (the line workspace of the context is the step's afterwards): the calls the general path makes
    d_mol = d_mol_doppler = mol_struct = None
    if n_mol:
        d_mol = ctx.empty((nd, nus.size))
        if mol is not None:
            d_mol_doppler = ctx.empty((n_mol, nd))
            ctx.call("sdx_doppler_widths_dev", n_mol, nd, P("m_nu"), P("m_mass"), P("temps"), _microturbulence_cgs(stellar_model), d_mol_doppler.ptr)
            ctx.call("sdx_line_opacity_dev", nd, nus.size, P("nus"), 0, nus.size, n_mol, P("m_nu"), d_mol_doppler.ptr, P("m_gamma"),
                     1 if "radiation" in opacity.line.broadening else nd, P("m_alpha"), d_mol.ptr, nus.size, 0, None)
        else:
            mol_struct = _linelist_struct(mol_spec, "ms", P)
            ctx.call("sdx_line_opacity_linelist_dev", nd, nus.size, P("nus"), 0, nus.size, C.byref(mol_struct), d_mol.ptr, nus.size, 0, None)
    d_F, d_total = ctx.empty((nd, nus.size)), ctx.empty((nd, nus.size))
    d_line = ctx.empty((nd, nus.size)) if n_lines else None
    dense = line is not None and n_lines
    step = (nd, nus.size, P("nus"), 0, nus.size, n_lines if dense else 0, P("l_nu"), d_doppler.ptr if dense else None, d_gamma.ptr if dense else None, nd,
            P("l_alpha"), C.byref(c), int(config.no_of_thetas), P("temps"), P("ray"), P("wts"), d_line.ptr if n_lines else None, d_total.ptr,
            d_F.ptr, nus.size)
    if tracked:  # every ray's intensity at every depth point stays on the device until somebody reads field.I_nus
        field._I_dev = ctx.empty((nd, nus.size, int(config.no_of_thetas)))
    line_struct = None
    if spherical or n_mol or (line_spec is not None and n_lines):
        opt = SynthesisOptions()
        opt.source, opt.source_ld = P("source"), nus.size
        opt.I_nus = field._I_dev.ptr if tracked else None
        opt.inward_rays, opt.photospheric_correction = (1 if spherical else 0), float(correction)
        if n_mol:
            opt.n_line_planes, opt.line_plane[0], opt.line_plane_ld = 1, d_mol.ptr, nus.size
        if line_spec is not None and n_lines:
            line_struct = _linelist_struct(line_spec, "ls", P)
            opt.linelist = C.pointer(line_struct)
        ctx.call("sdx_synthesize_opt_dev", *step, C.byref(opt), None)
    elif tracked or source is not None:
        ctx.call("sdx_synthesize_ex_dev", *step, P("source"), nus.size, field._I_dev.ptr if tracked else None, None)
    else:
        ctx.call("sdx_synthesize_dev", *step, None)
    # F_nu lands in page-locked memory by DMA (no bounce buffer, no second copy); the block returns to the context's pool when
    # the last reference to the array is gone
    field.F_nu = ctx.pinned.empty((nd, nus.size))
    if field.F_nu is not None:
        ctx.call("sdx_memcpy_d2h_pinned", field.F_nu.ctypes.data, d_F.ptr, field.F_nu.nbytes)
    else:
        field.F_nu = np.empty((nd, nus.size))
        ctx.call("sdx_memcpy_d2h", field.F_nu.ctypes.data, d_F.ptr, field.F_nu.nbytes)
    blob._staging = None  # (the download above synchronised: the staging block may go back to the pool)
    opac._total_twin = d_total
    field._device_blob = (blob, file_planes)  # keeps the staged inputs alive as long as the lazy entries may need them

    # ---- dictionary entries, the reference's keys in the reference's order (:655-738); planes on first read
    entries = opac.opacities_dict
    put = lambda key, value: dict.__setitem__(entries, key, value)  # noqa: E731

    def twin(dev):
        return _Thunk(lambda: B._download(dev))

    def remembered(key, make):
        def run():
            value = make()
            t = B.device_twin(value)
            if t is not None:
                opac._remember(key, value, t)
            return value
        return _Thunk(run)

    # The continuum entries are formed on first read by the per-source entry points the general path calls (the same device
    # functions: bit-identical planes) — from the arrays THIS call derived and sent up (private copies: `bf`, `dv`, `nus`, `temps`),
    # never from the plasma again: whatever happens to the plasma's tables afterwards, replaced or edited in place, the entries
    # belong to this field's F_nu, as the reference's eagerly computed ones do.
    nus_own, temps_own = nus.copy(), temps.copy()
    ff_plane = np.vstack(dv["ff_dens"]) if dv["ff_dens"] else np.zeros((0, nd))

    def rayleigh_plane():
        picks = {}
        if "H" in rayleigh_species:
            picks["n_h"] = dv["n_h"]
        if "He" in rayleigh_species:
            picks["n_he"] = dv["n_he"]
        if "H2" in rayleigh_species:
            picks["n_h2"] = dv["n_h2"]
        return B._download(ops.alpha_rayleigh(nus_own.copy(), nd, **picks)[0])  # (no frequency above the cut-off here: nothing to clip)

    for k, (source, fpath) in enumerate(opacity.file.items()):
        if file_planes:  # the plane the step added is the entry
            put(f"alpha_file_{source}", twin(file_planes[k]))
        else:  # one 1-D table, interpolated inside the step
            put(f"alpha_file_{source}", remembered(f"alpha_file_{source}", lambda: B._download(
                ops.alpha_file_1d(K.nu_to_angstrom(nus_own), table[1], table[2], dv["file"]))))
    put("alpha_bf", remembered("alpha_bf", lambda: B._download(ops.alpha_bf(nus_own, bf[0], bf[1], bf[2], bf[3], nd))))
    put("alpha_ff", remembered("alpha_ff", lambda: B._download(ops.alpha_ff(nus_own, temps_own, ff_ions, ff_plane))))
    put("alpha_rayleigh", remembered("alpha_rayleigh", rayleigh_plane))
    put("alpha_electron", 0 if opacity.disable_electron_scattering else remembered(
        "alpha_electron", lambda: B._download(ops.alpha_electron(nus_own.size, n_e))))

    def tables_of(spec):
        """gammas / doppler_widths of a deferred list, formed the way the general path forms them (sdx_line_params_dev) on
        first read of either entry."""
        cache = {}

        def get(k):
            if not cache:
                from stardis_amd import linelist as LL

                _, cache["g"], cache["d"] = LL.line_params(spec, ctx, alphas=False) if B.RETURN_BROADENING_TABLES else (None, None, None)
            return cache[k]
        return _Thunk(lambda: get("g")), _Thunk(lambda: get("d"))

    if opacity.line.disable:
        put("alpha_line_at_nu", 0), put("alpha_line_at_nu_gammas", 0), put("alpha_line_at_nu_doppler_widths", 0)
    elif line_spec is not None:
        put("alpha_line_at_nu", twin(d_line) if n_lines else np.zeros((nd, nus.size)))
        g_thunk, d_thunk = tables_of(line_spec)
        put("alpha_line_at_nu_gammas", g_thunk), put("alpha_line_at_nu_doppler_widths", d_thunk)
    elif n_lines:
        put("alpha_line_at_nu", twin(d_line))
        put("alpha_line_at_nu_gammas", _Thunk(d_gamma.numpy))
        put("alpha_line_at_nu_doppler_widths", _Thunk(d_doppler.numpy))
    else:  # no line on the grid: what the general path returns for an empty selection
        put("alpha_line_at_nu", np.zeros((nd, nus.size)))
        put("alpha_line_at_nu_gammas", np.zeros((0, nd))), put("alpha_line_at_nu_doppler_widths", np.zeros((0, nd)))
    if opacity.line.include_molecules:  # (:716-736)
        if opacity.line.disable:
            put("molecule_alpha_line_at_nu", 0), put("molecule_alpha_line_at_nu_gammas", 0), put("molecule_alpha_line_at_nu_doppler_widths", 0)
        elif mol_spec is not None:
            put("molecule_alpha_line_at_nu", twin(d_mol) if n_mol else np.zeros((nd, nus.size)))
            g_thunk, d_thunk = tables_of(mol_spec)
            put("molecule_alpha_line_at_nu_gammas", g_thunk), put("molecule_alpha_line_at_nu_doppler_widths", d_thunk)
        else:
            radiation = "radiation" in opacity.line.broadening
            put("molecule_alpha_line_at_nu", twin(d_mol) if n_mol else np.zeros((nd, nus.size)))
            a_ul = mol["a_ul"]
            put("molecule_alpha_line_at_nu_gammas", a_ul[:, np.newaxis].copy() if radiation else np.zeros((n_mol, nd)))
            put("molecule_alpha_line_at_nu_doppler_widths", _Thunk(d_mol_doppler.numpy) if n_mol else np.zeros((0, nd)))
    # what this field keeps on the device until its entries are read (or it is released): bounded per process
    import weakref

    held = [d_total, d_line, d_gamma, d_doppler, d_mol, d_mol_doppler, blob, getattr(field, "_I_dev", None), *file_planes]
    opac._device_bytes = int(sum(int(np.prod(a.shape)) * a.dtype.itemsize for a in held if a is not None))
    _enforce_budget(opac._device_bytes)
    _LIVE.append((weakref.ref(field), opac._device_bytes))
    return field
