"""Array-level Python front-end of the C ABI: numpy (host) in, numpy out, all arithmetic on the GPU.

Each function names the reference callable it stands in for (paths relative to
stardis/radiation_field/).  These are what the reference-named modules under
stardis_amd/radiation_field/ call; they add no numerics of their own.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import DeviceArray, default_context, ptr_of

F8 = np.float64


def _host(a, dtype=F8):
    a = np.asarray(_lib.plain(a), dtype=dtype)
    return a if a.ndim == 0 else np.ascontiguousarray(a)  # ascontiguousarray would turn a scalar into shape (1,)


def _dev(ctx, a, dtype=F8):
    """DeviceArray for a host array (uploads) or pass a DeviceArray / CUDA tensor through."""
    if a is None:
        return None
    if isinstance(a, DeviceArray) or hasattr(a, "data_ptr"):
        return a
    return ctx.upload(_host(a, dtype), dtype)


# ------------------------------------------------------------------------------------------------ voigt.py
def faddeeva(z, ctx=None):
    """opacities/opacities_solvers/voigt.py:89-91"""
    ctx = ctx or default_context()
    zz = np.ascontiguousarray(np.asarray(z), dtype=np.complex128)
    d_z = ctx.upload(zz.view(F8).reshape(-1), F8)
    d_w = ctx.empty(zz.size * 2)
    ctx.call("sdx_faddeeva_dev", zz.size, d_z.ptr, d_w.ptr)
    out = d_w.numpy().view(np.complex128).reshape(zz.shape)
    return out if out.ndim else out[()]


def voigt_profile(delta_nu, doppler_width, gamma, ctx=None):
    """opacities/opacities_solvers/voigt.py:153-155"""
    ctx = ctx or default_context()
    a, b, c = np.broadcast_arrays(_host(delta_nu), _host(doppler_width), _host(gamma))
    if np.any(b == 0):
        raise ZeroDivisionError("float division by zero")  # voigt.py:148; reference test_voigt.py:130-148
    d = [ctx.upload(np.ascontiguousarray(x)) for x in (a, b, c)]
    out = ctx.empty(a.shape)
    ctx.call("sdx_voigt_profile_dev", a.size, d[0].ptr, d[1].ptr, d[2].ptr, out.ptr)
    r = out.numpy()
    return r if r.ndim else r[()]


def voigt_term(delta_nu, doppler_width, gamma, alpha=1.0, ctx=None, fp32=False):
    """alpha * voigt_profile(delta_nu, doppler_width, gamma) through the routine the LINE KERNELS evaluate (real part only,
    FMA arithmetic, one reciprocal per point; sdx_math.h voigt_term) instead of the reference-order element-wise one.
    The derived constants are formed exactly as the pre-pass forms them (voigt.py:148-149, base.py:627).
    fp32=True: the packed-fp32 routine the mixed_precision option uses for narrow windows (sdx_math.h voigt_add32)."""
    ctx = ctx or default_context()
    dnu, dw, g, a = (np.ascontiguousarray(v) for v in np.broadcast_arrays(_host(delta_nu), _host(doppler_width), _host(gamma), _host(alpha)))
    if np.any(dw == 0):
        raise ZeroDivisionError("float division by zero")
    sqrt_pi = np.float64(1.7724538509055159)
    inv = 1.0 / dw
    y = (g / (sqrt_pi * np.float64(np.pi))) / dw
    amp = a / (sqrt_pi * dw)
    d = [ctx.upload(v) for v in (dnu, inv, y, amp)]
    out = ctx.empty(dnu.shape)
    ctx.call("sdx_voigt_term_f32_dev" if fp32 else "sdx_voigt_term_dev", dnu.size, d[0].ptr, d[1].ptr, d[2].ptr, d[3].ptr, out.ptr)
    r = out.numpy()
    return r if r.ndim else r[()]


# ------------------------------------------------------------------------------------------------ line opacity
def _line_inputs(no_of_depth_points, line_nus, doppler_widths, gammas, alphas_array, sort=True):
    """The four line arrays as contiguous float64 in the kernels' shapes.  sort=True: stably sorted by frequency when they
    are not (the line-opacity kernels want an ascending list); sort=False leaves the caller's order alone (per-line outputs)."""
    ln = _host(line_nus).reshape(-1)
    nd = int(no_of_depth_points)
    dw = _host(doppler_widths).reshape(ln.size, nd)  # may arrive F-ordered from DataFrame.to_numpy() (base.py:403-407)
    al = _host(alphas_array).reshape(ln.size, nd)
    g = _host(gammas)
    g = g.reshape(ln.size, -1) if ln.size else g.reshape(0, 1)
    if g.shape[1] not in (1, nd):
        raise ValueError(f"gammas must have shape (n_lines, {nd}) or (n_lines, 1), got {g.shape}")
    if sort and ln.size > 1 and np.any(ln[1:] < ln[:-1]):
        # The reference treats every line independently (base.py:548-590), so any order is legal there; the kernels want
        # ascending frequency (what calc_alpha_line_at_nu hands over, :392-397).  A stable sort changes only the order of
        # the per-point sum, which differs from the reference's per-thread slabs anyway.
        order = np.argsort(ln, kind="stable")
        ln, dw, g, al = (np.ascontiguousarray(a[order]) for a in (ln, dw, g, al))
    return ln, dw, g, al


def calc_alan_entries(no_of_depth_points, tracing_nus_values, line_nus, doppler_widths, gammas, alphas_array,
                      return_evaluations=False, ctx=None):
    """opacities/opacities_solvers/base.py:487-592 through sdx_line_opacity_f64 (host buffers in and out)."""
    ctx = ctx or default_context()
    nus = _host(tracing_nus_values).reshape(-1)
    ln, dw, g, al = _line_inputs(no_of_depth_points, line_nus, doppler_widths, gammas, alphas_array)
    nd = int(no_of_depth_points)
    out = np.empty((nd, nus.size))
    ev = C.c_int64(0)
    _lib.check(
        ctx.lib.sdx_line_opacity_f64(
            ctx.handle, nd, nus.size, nus.ctypes.data, ln.size, ln.ctypes.data, dw.ctypes.data, g.ctypes.data,
            g.shape[1], al.ctypes.data, out.ctypes.data, C.byref(ev),
        )
    )
    return (out, ev.value) if return_evaluations else out


def line_windows(no_of_depth_points, tracing_nus_values, line_nus, doppler_widths, gammas, alphas_array, ctx=None):
    """Window bounds [lower, upper) per (line, depth): opacities/opacities_solvers/base.py:556-575.  Row k belongs to line k
    of the caller's list whatever its order (the window rule is per line; sdx_line_windows_dev does not need a sorted list)."""
    ctx = ctx or default_context()
    nus = _host(tracing_nus_values).reshape(-1)
    ln, dw, g, al = _line_inputs(no_of_depth_points, line_nus, doppler_widths, gammas, alphas_array, sort=False)
    nd = int(no_of_depth_points)
    d = [ctx.upload(x) for x in (nus, ln, dw, g, al)]
    lo = ctx.empty((ln.size, nd), np.int32)
    hi = ctx.empty((ln.size, nd), np.int32)
    ctx.call("sdx_line_windows_dev", nd, nus.size, d[0].ptr, ln.size, d[1].ptr, d[2].ptr, d[3].ptr, g.shape[1], d[4].ptr,
             lo.ptr, hi.ptr)
    return lo.numpy(), hi.numpy()


# ------------------------------------------------------------------------------------------------ broadening
def _flags(linear_stark, quadratic_stark, van_der_waals, radiation):
    return (1 if linear_stark else 0) | (2 if quadratic_stark else 0) | (4 if van_der_waals else 0) | (8 if radiation else 0)


def calc_gamma(atomic_number, ion_number, ionization_energy, upper_level_energy, lower_level_energy, A_ul,
               electron_density, temperature, h_density, linear_stark=True, quadratic_stark=True, van_der_waals=True,
               radiation=True, ctx=None):
    """opacities/opacities_solvers/broadening.py:550-656; per-line inputs (N_l,) or (N_l,1), per-depth (N_d,)"""
    ctx = ctx or default_context()
    z = _host(atomic_number, np.int32).reshape(-1)
    ion = _host(ion_number, np.int32).reshape(-1)
    per_line = [_host(x).reshape(-1) for x in (ionization_energy, upper_level_energy, lower_level_energy, A_ul)]
    per_depth = [_host(x).reshape(-1) for x in (electron_density, temperature, h_density)]
    nd = per_depth[0].size
    d = [ctx.upload(z, np.int32), ctx.upload(ion, np.int32)] + [ctx.upload(x) for x in per_line + per_depth]
    out = ctx.empty((z.size, nd))
    ctx.call("sdx_calc_gamma_dev", z.size, nd, *[x.ptr for x in d],
             _flags(linear_stark, quadratic_stark, van_der_waals, radiation), out.ptr)
    return out.numpy()


def doppler_widths(line_nus, atomic_mass, temperature, microturbulence, ctx=None):
    """opacities/opacities_solvers/broadening.py:69-71 broadcast (N_l,1) x (N_d,) as at :723-730"""
    ctx = ctx or default_context()
    ln = _host(line_nus).reshape(-1)
    m = _host(atomic_mass).reshape(-1)
    t = _host(temperature).reshape(-1)
    d = [ctx.upload(x) for x in (ln, m, t)]
    out = ctx.empty((ln.size, t.size))
    ctx.call("sdx_doppler_widths_dev", ln.size, t.size, d[0].ptr, d[1].ptr, d[2].ptr, float(microturbulence), out.ptr)
    return out.numpy()


def calc_vald_gamma_arrays(atomic_number, ion_number, ionization_energy, upper_level_energy, lower_level_energy, A_ul,
                           stark, waals, atomic_mass, electron_density, temperature, h_density, linear_stark,
                           quadratic_stark, van_der_waals, radiation, ctx=None, halve=True):
    """opacities/opacities_solvers/broadening.py:1009-1085 on plain arrays (halve=False: the sum as :771-799 leaves it)"""
    ctx = ctx or default_context()
    z = _host(atomic_number, np.int32).reshape(-1)
    ion = _host(ion_number, np.int32).reshape(-1)
    per_line = [_host(x).reshape(-1) for x in (ionization_energy, upper_level_energy, lower_level_energy, A_ul, stark, waals, atomic_mass)]
    per_depth = [_host(x).reshape(-1) for x in (electron_density, temperature, h_density)]
    nd = per_depth[0].size
    d = [ctx.upload(z, np.int32), ctx.upload(ion, np.int32)] + [ctx.upload(x) for x in per_line + per_depth]
    out = ctx.empty((z.size, nd))
    ctx.call("sdx_calc_vald_gamma_dev", z.size, nd, *[x.ptr for x in d],
             _flags(linear_stark, quadratic_stark, van_der_waals, radiation) | (0 if halve else 16), out.ptr)
    return out.numpy()


def broadening_scalar(op, *operands, ctx=None):
    """The reference's element-wise broadening ufuncs (broadening.py:69-71,140-146,232-234,346-360,476-490)."""
    ctx = ctx or default_context()
    arrs = np.broadcast_arrays(*[_host(x) for x in operands])
    shape = arrs[0].shape
    d = [ctx.upload(np.ascontiguousarray(a)) for a in arrs]
    while len(d) < 5:
        d.append(None)
    out = ctx.empty(shape)
    ctx.call("sdx_broadening_scalar_dev", int(op), int(np.prod(shape, dtype=np.int64)), *[ptr_of(x) for x in d], out.ptr)
    r = out.numpy()
    return r if r.ndim else r[()]


# ------------------------------------------------------------------------------------------------ continuum
def alpha_file_1d(lambdas, table_wavelength, table_sigma, density, ctx=None):
    """opacities/opacities_solvers/base.py:40-70 with util.py:94-103 (np.interp table, e.g. Hminus_bf)"""
    ctx = ctx or default_context()
    lam, x, y, n = (_host(a).reshape(-1) for a in (lambdas, table_wavelength, table_sigma, density))
    d = [ctx.upload(a) for a in (lam, x, y, n)]
    out = ctx.empty((n.size, lam.size))
    ctx.call("sdx_alpha_file_1d_dev", n.size, lam.size, d[0].ptr, x.size, d[1].ptr, d[2].ptr, d[3].ptr, out.ptr, lam.size)
    return out


def alpha_file_2d(sigma, density, ctx=None):
    """opacities/opacities_solvers/base.py:70 with a (N_d, N_nu) sigma (util.py:35-91); sigma may already be on the device"""
    ctx = ctx or default_context()
    d_s = sigma if isinstance(sigma, DeviceArray) else ctx.upload(_host(sigma))
    n = _host(density).reshape(-1)
    d_n = ctx.upload(n)
    out = ctx.empty(d_s.shape)
    ctx.call("sdx_alpha_file_2d_dev", d_s.shape[0], d_s.shape[1], d_s.ptr, d_s.shape[1], d_n.ptr, out.ptr, d_s.shape[1])
    return out


def upload_sigma_table(wave, axis2, cell_simplices, transform, simplex_values, ctx=None):
    """The table side of sigma_table_2d on the device, for callers that evaluate the same table again and again."""
    ctx = ctx or default_context()
    return tuple(ctx.upload(_host(x)) for x in (np.reshape(_host(wave), -1), np.reshape(_host(axis2), -1), transform, simplex_values)) + (
        ctx.upload(_host(cell_simplices, np.int32), np.int32),)


def sigma_table_2d(wave, axis2, cell_simplices, transform, simplex_values, lambdas, second, scale_kind=0, temperatures=None, ctx=None,
                   table_dev=None):
    """opacities/opacities_solvers/util.py:35-91: LinearNDInterpolator on the table's triangulation at the mesh
    (lambdas, second) -> (sigma DeviceArray (len(second), len(lambdas)), rows that contain an exact zero).
    table_dev: what upload_sigma_table returned for this table (skips five uploads)."""
    ctx = ctx or default_context()
    wave, axis2 = _host(wave).reshape(-1), _host(axis2).reshape(-1)
    lam, sec = _host(lambdas).reshape(-1), _host(second).reshape(-1)
    if table_dev is None:
        table_dev = upload_sigma_table(wave, axis2, cell_simplices, transform, simplex_values, ctx)
    d = list(table_dev[:4]) + [ctx.upload(lam), ctx.upload(sec)]
    d_cells = table_dev[4]
    d_t = ctx.upload(_host(temperatures).reshape(-1)) if temperatures is not None else None
    out = ctx.empty((sec.size, lam.size))
    zero = ctx.empty((sec.size,), np.int32)
    ctx.call("sdx_sigma_table_2d_dev", wave.size, d[0].ptr, axis2.size, d[1].ptr, d_cells.ptr, d[2].ptr, d[3].ptr, sec.size, lam.size,
             d[4].ptr, d[5].ptr, int(scale_kind), d_t.ptr if d_t is not None else None, out.ptr, lam.size, zero.ptr)
    return out, np.flatnonzero(zero.numpy())


def alpha_bf(tracing_nus, species_offsets, species_ion_number, cutoff_frequency, level_number_density, n_depth, ctx=None):
    """opacities/opacities_solvers/base.py:178-271"""
    ctx = ctx or default_context()
    nus = _host(tracing_nus).reshape(-1)
    off = _host(species_offsets, np.int32).reshape(-1)
    ion = _host(species_ion_number, np.int32).reshape(-1)
    cut = _host(cutoff_frequency).reshape(-1)
    ld = _host(level_number_density).reshape(cut.size, n_depth)
    d_nus = ctx.upload(nus)
    d = [ctx.upload(off, np.int32), ctx.upload(ion, np.int32), ctx.upload(cut), ctx.upload(ld)]
    out = ctx.empty((n_depth, nus.size))
    ctx.call("sdx_alpha_bf_dev", n_depth, nus.size, d_nus.ptr, ion.size, *[x.ptr for x in d], out.ptr, nus.size)
    return out


def alpha_ff(tracing_nus, temperature, species_ion_number, number_density, ctx=None):
    """opacities/opacities_solvers/base.py:274-317"""
    ctx = ctx or default_context()
    nus = _host(tracing_nus).reshape(-1)
    t = _host(temperature).reshape(-1)
    ion = _host(species_ion_number, np.int32).reshape(-1)
    n = _host(number_density).reshape(ion.size, t.size) if ion.size else np.zeros((0, t.size))
    d = [ctx.upload(nus), ctx.upload(t), ctx.upload(ion, np.int32), ctx.upload(n)]
    out = ctx.empty((t.size, nus.size))
    ctx.call("sdx_alpha_ff_dev", t.size, nus.size, d[0].ptr, d[1].ptr, ion.size, d[2].ptr, d[3].ptr, out.ptr, nus.size)
    return out


def alpha_rayleigh(tracing_nus_inout, n_depth, n_h=None, n_he=None, n_h2=None, ctx=None):
    """opacities/opacities_solvers/base.py:74-135; returns (alpha_dev, clipped_nus) — the caller writes the
    clipped frequencies back into its own array to reproduce the in-place mutation at :99"""
    ctx = ctx or default_context()
    nus = _host(tracing_nus_inout).reshape(-1)
    d_nus = ctx.upload(nus)
    dens = [None if a is None else ctx.upload(_host(a).reshape(-1)) for a in (n_h, n_he, n_h2)]
    out = ctx.empty((n_depth, nus.size))
    ctx.call("sdx_alpha_rayleigh_dev", n_depth, nus.size, d_nus.ptr, *[ptr_of(x) for x in dens], out.ptr, nus.size)
    return out, d_nus.numpy()


def alpha_electron(n_nu, electron_density, ctx=None):
    """opacities/opacities_solvers/base.py:139-174"""
    ctx = ctx or default_context()
    ne = _host(electron_density).reshape(-1)
    d = ctx.upload(ne)
    out = ctx.empty((ne.size, int(n_nu)))
    ctx.call("sdx_alpha_electron_dev", ne.size, int(n_nu), d.ptr, out.ptr, int(n_nu))
    return out


def accumulate(total_dev, src_dev, ctx=None):
    """opacities/base.py:24-28: total += src, both (N_d, N_nu) device arrays"""
    ctx = ctx or default_context()
    nd, nn = total_dev.shape
    ctx.call("sdx_accumulate_dev", nd, nn, total_dev.ptr, nn, src_dev.ptr, nn)


# ------------------------------------------------------------------------------------------------ formal solution
def blackbody_flux_at_nu(tracing_nus, temps, ctx=None):
    """source_functions/blackbody.py:10-35; temps (N_d,1) or (N_d,)"""
    ctx = ctx or default_context()
    nus = _host(tracing_nus).reshape(-1)
    t = _host(temps).reshape(-1)
    d_n, d_t = ctx.upload(nus), ctx.upload(t)
    out = ctx.empty((t.size, nus.size))
    ctx.call("sdx_blackbody_dev", t.size, nus.size, d_n.ptr, d_t.ptr, out.ptr, nus.size)
    return out.numpy()


def calc_weights_parallel(delta_tau, ctx=None):
    """radiation_field_solvers/base.py:6-47"""
    ctx = ctx or default_context()
    tau = _host(delta_tau)
    d = ctx.upload(tau)
    w = [ctx.empty(tau.shape) for _ in range(3)]
    ctx.call("sdx_calc_weights_dev", tau.size, d.ptr, w[0].ptr, w[1].ptr, w[2].ptr)
    return tuple(x.numpy() for x in w)


def raytrace_arrays(tracing_nus, temperatures, ray_distances, theta_weights, total_alphas, F_nu=None, track=False, ctx=None,
                    inward_rays=False, photospheric_correction=1.0, want_flux=True, source=None):
    """radiation_field_solvers/base.py:271-346 on arrays.

    ray_distances is the (N_d-1, N_theta) table of :302-305 (plane-parallel) or of calculate_spherical_ray
    (:296-300; pass inward_rays=True and the photospheric correction of :340-344).  total_alphas may be a host
    array or a device array.  source: optional (N_d, N_nu) source-function plane (a foreign RadiationField.source_function
    evaluated by the caller, :133); default: the Planck function in the kernel.
    Returns (F_nu host array — accumulated into when given —, I_nus or None)."""
    ctx = ctx or default_context()
    nus = _host(tracing_nus).reshape(-1)
    t = _host(temperatures).reshape(-1)
    rd = _host(ray_distances).reshape(t.size - 1, -1)
    w = _host(theta_weights).reshape(-1)
    n_theta = w.size
    d_alpha = _dev(ctx, total_alphas)
    d = [ctx.upload(nus), ctx.upload(t), ctx.upload(rd), ctx.upload(w)]
    d_F = None
    if want_flux:
        d_F = ctx.zeros((t.size, nus.size)) if F_nu is None else ctx.upload(_host(F_nu))
    d_I = ctx.empty((t.size, nus.size, n_theta)) if track else None
    # accumulate only into a flux the caller handed over (base.py:336 adds to F_nu); a fresh one is written — the entry point
    # then picks its kernel freely (the segmented formal solution needs nothing to add to)
    args = (t.size, nus.size, n_theta, d[0].ptr, d[1].ptr, d[2].ptr, d[3].ptr, ptr_of(d_alpha), nus.size, ptr_of(d_F), nus.size,
            ptr_of(d_I), 0 if F_nu is None else 1)
    if source is not None:
        src = _host(source)
        if src.shape != (t.size, nus.size):
            raise ValueError(f"source function must return shape {(t.size, nus.size)}, got {src.shape}")
        d_S = ctx.upload(src)
        ctx.call("sdx_raytrace_source_dev", *args[:9], d_S.ptr, nus.size, *args[9:], 1 if inward_rays else 0, float(photospheric_correction))
    elif inward_rays:
        ctx.call("sdx_raytrace_spherical_dev", *args, float(photospheric_correction))
    else:
        ctx.call("sdx_raytrace_dev", *args)
    return (d_F.numpy() if want_flux else None), (d_I.numpy() if track else None)
