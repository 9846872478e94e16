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

Layout: this module is the call itself (`try_fused`: eligibility, staging, the step, the dictionary); `fused_inputs` packs the
plasma's tables into flat arrays, `fused_cache` keeps those derivations per plasma object verified by content, `fused_lazy`
holds the lazily materialised outputs and the device-memory budget.  The knobs of all three live here.
"""
import ctypes as C
from pathlib import Path

import numpy as np

from stardis_amd import ops
from stardis_amd import constants as K
from stardis_amd._lib import Continuum, SynthesisOptions, default_context, plain
from stardis_amd.radiation_field.fused_cache import _MEMO, _memo, _witness, clear_cache, one_call  # noqa: F401
from stardis_amd.radiation_field.fused_inputs import (  # noqa: F401
    _bf_arrays, _deferred_atomic, _deferred_molecules, _depth_vectors, _leggauss, _line_arrays, _linelist_struct, _mass_of,
    _molecule_arrays, _packed_upload, _sorted_line_tables, _stage_linelist)
from stardis_amd.radiation_field.fused_lazy import (  # noqa: F401
    _LIVE, FusedOpacities, LazyOpacitiesDict, _enforce_budget, _Thunk, _tracked_class, release_device)
from stardis_amd.radiation_field.opacities.opacities_solvers import base as B
from stardis_amd.radiation_field.opacities.opacities_solvers.broadening import _microturbulence_cgs, _switches
from stardis_amd.radiation_field.opacities.opacities_solvers.util import get_number_density, read_table, sigma_file_device
from stardis_amd.radiation_field.radiation_field_solvers.base import _source_plane, calculate_spherical_ray

F8 = np.float64
RAYLEIGH_CUTOFF = 2.3e15  # opacities_solvers/base.py:99

# knobs (read by fused_cache / fused_lazy on every call)
CACHE = True  # False derives everything from the plasma's tables on every call
_MEMO_MAX = 16  # cached derivations, least recently used first out
MEMO_MAX_BYTES = 2 << 30  # ... and at most this much derived data (numpy arrays in the cached values; the newest entry always stays)
DEVICE_BUDGET_BYTES = 8 << 30  # device memory the lazy entries of live fused fields may hold per process (fused_lazy)


def try_fused(field_cls, tracing_nus, stellar_model, stellar_plasma, config, source_function):
    """-> RadiationField computed by one fused device pass, or None when the configuration needs the general path."""
    with one_call():  # (witnesses are per call and per thread: the tables may be edited before the next one)
        return _try_fused(field_cls, tracing_nus, stellar_model, stellar_plasma, config, source_function)


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
    if nus.max() > RAYLEIGH_CUTOFF:
        # the reference zeroes the caller's frequencies above the cut-off in place (:99) — whatever the species list, an empty one
        # included — and everything after calc_alpha_rayleigh sees the zeros: the general path reproduces it
        return None
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
