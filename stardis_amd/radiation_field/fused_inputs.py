"""Input packing of the fused call: the plasma's and the model's pandas / astropy objects as the flat arrays the step reads
(calc_alpha_line_at_nu's host preparation, opacities_solvers/base.py:362-421; get_number_density, util.py:111-166; the bf level
walk, :204-226), each derivation kept by `fused_cache._memo`, and the one staging copy that takes them to the device."""
import numpy as np

from stardis_amd import constants as K
from stardis_amd._lib import LineListStruct, plain
from stardis_amd.radiation_field.fused_cache import _memo
from stardis_amd.radiation_field.opacities.opacities_solvers import base as B
from stardis_amd.radiation_field.opacities.opacities_solvers.broadening import _microturbulence_cgs
from stardis_amd.radiation_field.opacities.opacities_solvers.util import get_number_density

F8 = np.float64

_QUADRATURE = {}


def _leggauss(n):
    """np.polynomial.legendre.leggauss(n) as RadiationField.__init__ maps it (radiation_field/base.py:61-63), once per n."""
    q = _QUADRATURE.get(n)
    if q is None:
        nodes, weights = np.polynomial.legendre.leggauss(n)
        q = _QUADRATURE[n] = ((nodes / 2) + 0.5 * np.pi / 2, weights * np.pi / 2)
    return q[0].copy(), q[1].copy()


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
