"""ctypes front-end of the CPU oracle (oracle/stardis_oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import this
package; the product (stardis_amd/) never does.  Every function cites the reference
lines its C counterpart restates.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_build", "libstardis_oracle.so")
_lib = None

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)
_lp = C.POINTER(C.c_int64)


def build(force=False):
    """make decides: the library is rebuilt when the source is newer or when it was built (-march=native) on another CPU
    than this one (oracle/Makefile keys the build on the host CPU)."""
    import shutil

    if force and os.path.exists(_SO):
        os.remove(_SO)
    if shutil.which("make") is not None:
        subprocess.run(["make", "-C", _HERE], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    elif not os.path.exists(_SO):
        raise RuntimeError("oracle/_build/libstardis_oracle.so is missing and `make` is not available")
    return _SO


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(_SO)
        _lib.orc_voigt_profile.restype = C.c_double
        _lib.orc_voigt_profile.argtypes = [C.c_double] * 3
        _lib.orc_d_nu.restype = C.c_double
        _lib.orc_interp1.restype = C.c_double
        for n in (
            "orc_doppler_width",
            "orc_n_effective",
            "orc_gamma_linear_stark",
            "orc_gamma_quadratic_stark",
            "orc_gamma_van_der_waals",
            "orc_vald_stark",
            "orc_vald_vdw",
        ):
            getattr(_lib, n).restype = C.c_double
        _lib.orc_doppler_width.argtypes = [C.c_double] * 4
        _lib.orc_n_effective.argtypes = [C.c_int, C.c_double, C.c_double]
        _lib.orc_gamma_linear_stark.argtypes = [C.c_double] * 3
        _lib.orc_gamma_quadratic_stark.argtypes = [C.c_int] + [C.c_double] * 4
        _lib.orc_gamma_van_der_waals.argtypes = [C.c_int] + [C.c_double] * 4
    return _lib


def _d(a):
    a = np.ascontiguousarray(a, dtype=np.float64)
    return a, a.ctypes.data_as(_dp)


def _i(a):
    a = np.ascontiguousarray(a, dtype=np.int32)
    return a, a.ctypes.data_as(_ip)


def num_threads():
    return lib().orc_num_threads()


def set_num_threads(n):
    lib().orc_set_num_threads(int(n))


def faddeeva(z):
    """voigt.py:17-91"""
    z = np.ascontiguousarray(z, dtype=np.complex128)
    w = np.empty_like(z)
    lib().orc_faddeeva_array(C.c_int64(z.size), z.ctypes.data_as(_dp), w.ctypes.data_as(_dp))
    return w


def voigt_profile(delta_nu, doppler_width, gamma):
    """voigt.py:113-155"""
    a, b, c = np.broadcast_arrays(
        np.asarray(delta_nu, dtype=np.float64), np.asarray(doppler_width, dtype=np.float64), np.asarray(gamma, dtype=np.float64)
    )
    (a, pa), (b, pb), (c, pc) = _d(a), _d(b), _d(c)
    if np.any(b == 0):
        raise ZeroDivisionError("doppler_width == 0")  # reference test_voigt.py:130-148
    out = np.empty(a.shape)
    lib().orc_voigt_profile_array(C.c_int64(a.size), pa, pb, pc, out.ctypes.data_as(_dp))
    return out


def window(nus, line_nu, gamma, doppler_width, alpha):
    """opacities_solvers/base.py:556-575 -> (lower, upper)"""
    nus, pn = _d(nus)
    lo, hi = C.c_int64(), C.c_int64()
    L = lib()
    d_nu = L.orc_d_nu(C.c_int64(nus.size), pn)
    L.orc_window(
        C.c_int64(nus.size), pn, C.c_double(d_nu), C.c_double(line_nu), C.c_double(gamma), C.c_double(doppler_width),
        C.c_double(alpha), C.byref(lo), C.byref(hi),
    )
    return lo.value, hi.value


def calc_alan_entries(no_of_depth_points, tracing_nus_values, line_nus, doppler_widths, gammas, alphas_array, return_evals=False):
    """opacities_solvers/base.py:487-592"""
    nus, pn = _d(tracing_nus_values)
    ln, pl = _d(line_nus)
    nd = int(no_of_depth_points)
    dw, pdw = _d(np.asarray(doppler_widths).reshape(ln.size, nd))
    g, pg = _d(np.asarray(gammas).reshape(ln.size, -1) if ln.size else np.zeros((0, 1)))
    a, pa = _d(np.asarray(alphas_array).reshape(ln.size, nd))
    out = np.empty((int(no_of_depth_points), nus.size))
    ne = C.c_int64()
    rc = lib().orc_calc_alan_entries(
        C.c_int(int(no_of_depth_points)), C.c_int64(nus.size), pn, C.c_int64(ln.size), pl, pdw, pg,
        C.c_int(g.shape[1] if ln.size else 1), pa, out.ctypes.data_as(_dp), C.byref(ne),
    )
    if rc:
        raise MemoryError("oracle allocation failed")
    return (out, ne.value) if return_evals else out


def count_region_flips(no_of_depth_points, tracing_nus_values, line_nus, doppler_widths, gammas, alphas_array, tile=256, near=1e-12):
    """Every evaluation of calc_alan_entries (opacities_solvers/base.py:487-592) with the Humlicek region (voigt.py:31-44) decided
    from the reference's x = delta_nu / doppler_width and from the two forms the HIP kernels use -> dict(evaluations,
    flips_product_form, flips_tile_form, within_near_of_a_boundary)."""
    nus, pn = _d(tracing_nus_values)
    ln, pl = _d(line_nus)
    nd = int(no_of_depth_points)
    dw, pdw = _d(np.asarray(doppler_widths).reshape(ln.size, nd))
    g, pg = _d(np.asarray(gammas).reshape(ln.size, -1) if ln.size else np.zeros((0, 1)))
    a, pa = _d(np.asarray(alphas_array).reshape(ln.size, nd))
    out = (C.c_int64 * 4)()
    lib().orc_count_region_flips(C.c_int(nd), C.c_int64(nus.size), pn, C.c_int64(ln.size), pl, pdw, pg, C.c_int(g.shape[1] if ln.size else 1), pa,
                                 C.c_int(int(tile)), C.c_double(float(near)), out)
    return dict(evaluations=out[0], flips_product_form=out[1], flips_tile_form=out[2], within_near_of_a_boundary=out[3])


def calc_alan_entries_columns(cols, no_of_depth_points, tracing_nus_values, line_nus, doppler_widths, gammas, alphas_array, return_evals=False):
    """opacities_solvers/base.py:487-592 evaluated at the listed grid columns only (ascending global indices), with the window
    rule of the WHOLE grid (global d_nu, global centre, global clamp) -> (N_d, len(cols))"""
    nus, pn = _d(tracing_nus_values)
    ln, pl = _d(line_nus)
    nd = int(no_of_depth_points)
    cols = np.ascontiguousarray(cols, dtype=np.int64)
    if cols.size and (np.any(np.diff(cols) <= 0) or cols[0] < 0 or cols[-1] >= nus.size):
        raise ValueError("cols must be strictly ascending grid indices")
    dw, pdw = _d(np.asarray(doppler_widths).reshape(ln.size, nd))
    g, pg = _d(np.asarray(gammas).reshape(ln.size, -1) if ln.size else np.zeros((0, 1)))
    a, pa = _d(np.asarray(alphas_array).reshape(ln.size, nd))
    out = np.empty((nd, cols.size))
    ne = C.c_int64()
    rc = lib().orc_calc_alan_entries_columns(
        C.c_int(nd), C.c_int64(nus.size), pn, C.c_int64(cols.size), cols.ctypes.data_as(C.POINTER(C.c_int64)), C.c_int64(ln.size), pl, pdw, pg,
        C.c_int(g.shape[1] if ln.size else 1), pa, out.ctypes.data_as(_dp), C.byref(ne),
    )
    if rc:
        raise MemoryError("oracle allocation failed")
    return (out, ne.value) if return_evals else out


def blackbody_flux_at_nu(tracing_nus, temps):
    """blackbody.py:10-35; temps (N_d,1) or (N_d,)"""
    nus, pn = _d(tracing_nus)
    t, pt = _d(np.asarray(temps).reshape(-1))
    out = np.empty((t.size, nus.size))
    lib().orc_blackbody(C.c_int(t.size), C.c_int64(nus.size), pn, pt, out.ctypes.data_as(_dp))
    return out


def calc_weights_parallel(delta_tau):
    """radiation_field_solvers/base.py:6-47"""
    tau, pt = _d(delta_tau)
    w0, w1, w2 = np.empty_like(tau), np.empty_like(tau), np.empty_like(tau)
    lib().orc_calc_weights(C.c_int64(tau.size), pt, w0.ctypes.data_as(_dp), w1.ctypes.data_as(_dp), w2.ctypes.data_as(_dp))
    return w0, w1, w2


def single_theta_trace_parallel(ray_dist, temps, alphas, tracing_nus, inward_rays=False):
    """radiation_field_solvers/base.py:85-268"""
    rd, prd = _d(ray_dist)
    t, pt = _d(np.asarray(temps).reshape(-1))
    a, pa = _d(alphas)
    nus, pn = _d(tracing_nus)
    out = np.empty((t.size, nus.size))
    rc = lib().orc_single_theta_trace(C.c_int(t.size), C.c_int64(nus.size), prd, pt, pa, pn, out.ctypes.data_as(_dp), C.c_int(int(inward_rays)))
    if rc:
        raise MemoryError
    return out


def calculate_spherical_ray(thetas, radii):
    """radiation_field_solvers/base.py:349-381"""
    thetas = np.asarray(thetas, dtype=np.float64)
    r = np.asarray(radii, dtype=np.float64)
    out = np.zeros((len(r) - 1, len(thetas)))
    for k, theta in enumerate(thetas):
        b = r[-1] * np.sin(theta)
        with np.errstate(invalid="ignore"):
            dz = np.diff(np.sqrt(r**2 - b**2))
        out[~np.isnan(dz), k] = dz[~np.isnan(dz)]
    return out


def raytrace(tracing_nus, temps, dist, thetas, weights, total_alphas, F_nu=None, track=False, spherical_r=None, reference_r=None):
    """radiation_field_solvers/base.py:271-346; plane-parallel from `dist`, or spherical when spherical_r (radii) and
    reference_r are given (:296-300, :340-344).  Returns (F_nu, I_nus or None)"""
    nus, pn = _d(tracing_nus)
    t, pt = _d(np.asarray(temps).reshape(-1))
    th = np.asarray(thetas, dtype=np.float64)
    if spherical_r is not None:
        rdist, prd = _d(calculate_spherical_ray(th, spherical_r))
    else:
        rdist, prd = _d(np.asarray(dist, dtype=np.float64).reshape(-1, 1) / np.cos(th))
    w, pw = _d(weights)
    a, pa = _d(total_alphas)
    F = np.zeros((t.size, nus.size)) if F_nu is None else F_nu
    I_nus = np.zeros((t.size, nus.size, th.size)) if track else None
    rc = lib().orc_raytrace(
        C.c_int(t.size), C.c_int64(nus.size), C.c_int(th.size), pn, pt, prd, pw, pa, F.ctypes.data_as(_dp),
        I_nus.ctypes.data_as(_dp) if track else None, C.c_int(1 if spherical_r is not None else 0),
    )
    if rc:
        raise MemoryError
    if spherical_r is not None:
        F *= (np.asarray(spherical_r, dtype=np.float64)[-1] / reference_r) ** 2  # :340-344
    return F, I_nus


def alpha_file_1d(lambdas, tab_x, tab_y, density):
    """opacities_solvers/base.py:40-70 with util.py:94-103 (Hminus_bf)"""
    lam, pl = _d(lambdas)
    x, px = _d(tab_x)
    y, py = _d(tab_y)
    n, pn = _d(density)
    out = np.empty((n.size, lam.size))
    lib().orc_alpha_file_1d(C.c_int(n.size), C.c_int64(lam.size), pl, C.c_int(x.size), px, py, pn, out.ctypes.data_as(_dp))
    return out


def alpha_file_2d(sigma, density):
    s, ps = _d(sigma)
    n, pn = _d(density)
    out = np.empty_like(s)
    lib().orc_alpha_file_2d(C.c_int(s.shape[0]), C.c_int64(s.shape[1]), ps, pn, out.ctypes.data_as(_dp))
    return out


def alpha_bf(tracing_nus, species_offsets, species_ion_number, cutoff, level_density):
    """opacities_solvers/base.py:178-271"""
    nus, pn = _d(tracing_nus)
    off, po = _i(species_offsets)
    ion, pi = _i(species_ion_number)
    cut, pc = _d(cutoff)
    ld, pld = _d(level_density)
    nd = ld.shape[1]
    out = np.empty((nd, nus.size))
    lib().orc_alpha_bf(C.c_int(nd), C.c_int64(nus.size), pn, C.c_int(ion.size), po, pi, pc, pld, out.ctypes.data_as(_dp))
    return out


def alpha_ff(tracing_nus, temps, species_ion_number, number_density):
    """opacities_solvers/base.py:274-317"""
    nus, pn = _d(tracing_nus)
    t, pt = _d(temps)
    ion, pi = _i(species_ion_number)
    n, pnd = _d(np.asarray(number_density).reshape(ion.size, -1))
    out = np.empty((t.size, nus.size))
    lib().orc_alpha_ff(C.c_int(t.size), C.c_int64(nus.size), pn, pt, C.c_int(ion.size), pi, pnd, out.ctypes.data_as(_dp))
    return out


def alpha_rayleigh(tracing_nus, n_h=None, n_he=None, n_h2=None):
    """opacities_solvers/base.py:74-135; tracing_nus (float64, contiguous) is modified in place like the reference"""
    assert tracing_nus.dtype == np.float64 and tracing_nus.flags.c_contiguous
    arrs = [None if a is None else _d(a) for a in (n_h, n_he, n_h2)]
    nd = next(a[0].size for a in arrs if a is not None)
    out = np.empty((nd, tracing_nus.size))
    lib().orc_alpha_rayleigh(
        C.c_int(nd), C.c_int64(tracing_nus.size), tracing_nus.ctypes.data_as(_dp),
        *[None if a is None else a[1] for a in arrs], out.ctypes.data_as(_dp),
    )
    return out


def alpha_electron(n_nu, n_e):
    """opacities_solvers/base.py:139-174"""
    ne, pne = _d(n_e)
    out = np.empty((ne.size, int(n_nu)))
    lib().orc_alpha_electron(C.c_int(ne.size), C.c_int64(int(n_nu)), pne, out.ctypes.data_as(_dp))
    return out


def calc_gamma(atomic_number, ion_number, ionization_energy, upper_energy, lower_energy, A_ul, n_e, temps, n_h, flags=15):
    """broadening.py:550-656; ion_number already incremented as at :708-709"""
    z, pz = _i(atomic_number)
    ion, pi = _i(ion_number)
    args = [_d(x) for x in (ionization_energy, upper_energy, lower_energy, A_ul, n_e, temps, n_h)]
    nd = args[4][0].size
    out = np.empty((z.size, nd))
    lib().orc_calc_gamma(C.c_int64(z.size), C.c_int(nd), pz, pi, *[a[1] for a in args], C.c_int(flags), out.ctypes.data_as(_dp))
    return out


def doppler_widths(line_nus, mass, temps, microturbulence):
    """broadening.py:32-66 over (N_l, N_d) as at :723-730"""
    ln, pl = _d(line_nus)
    m, pm = _d(mass)
    t, pt = _d(temps)
    out = np.empty((ln.size, t.size))
    lib().orc_doppler_widths(C.c_int64(ln.size), C.c_int(t.size), pl, pm, pt, C.c_double(microturbulence), out.ctypes.data_as(_dp))
    return out


def calc_vald_gamma(atomic_number, ion_number, ionization_energy, upper_energy, lower_energy, A_ul, stark, waals, mass, n_e, temps, n_h, flags=15):
    """broadening.py:1009-1085"""
    z, pz = _i(atomic_number)
    ion, pi = _i(ion_number)
    args = [_d(x) for x in (ionization_energy, upper_energy, lower_energy, A_ul, stark, waals, mass, n_e, temps, n_h)]
    nd = args[7][0].size
    out = np.empty((z.size, nd))
    lib().orc_calc_vald_gamma(C.c_int64(z.size), C.c_int(nd), pz, pi, *[a[1] for a in args], C.c_int(flags), out.ctypes.data_as(_dp))
    return out


def alpha_line_linelist(e_low_ev, g_lo, strength, line_nus, pop_row, pop, temps, alpha_coefficient):
    """plasma/base.py:200-321, :348-455; plasma/molecules.py:214-320, :345-440 — (N_l, N_d) alpha_line table.
    g_lo None = short list (strength = 10**log_gf), else strength = f_lu."""
    e, pe = _d(e_low_ev)
    s, ps = _d(strength)
    ln, pl = _d(line_nus)
    r, pr = _i(pop_row)
    pp, ppop = _d(pop)
    t, pt = _d(temps)
    pg = None
    if g_lo is not None:
        g, pg = _d(g_lo)
    out = np.empty((e.size, t.size))
    lib().orc_alpha_line_linelist(C.c_int64(e.size), C.c_int(t.size), pe, pg, ps, pl, pr, ppop, pt, C.c_double(alpha_coefficient),
                                  out.ctypes.data_as(_dp))
    return out


def alpha_line_levels(level_density, lower_index, stim, f_lu, alpha_coefficient):
    """plasma/base.py:146-175 — (N_l, N_d)"""
    ld, pld = _d(level_density)
    li, pli = _i(lower_index)
    st, pst = _d(stim)
    f, pf = _d(f_lu)
    nd = ld.shape[1]
    out = np.empty((li.size, nd))
    lib().orc_alpha_line_levels(C.c_int64(li.size), C.c_int(nd), pld, pli, pst, pf, C.c_double(alpha_coefficient), out.ctypes.data_as(_dp))
    return out


def interp_triangulated(x_axis, y_axis, cell_simplices, transform, simplex_values, qx, qy):
    """util.py:47-56, :75-86 — LinearNDInterpolator(points, values, fill_value=0) at the mesh (qx, qy) -> (len(qy), len(qx))."""
    xa, pxa = _d(x_axis)
    ya, pya = _d(y_axis)
    cs, pcs = _i(cell_simplices)
    tr, ptr = _d(transform)
    sv, psv = _d(simplex_values)
    x, px = _d(qx)
    y, py = _d(qy)
    out = np.empty((y.size, x.size))
    lib().orc_interp_triangulated(C.c_int(xa.size), pxa, C.c_int(ya.size), pya, pcs, ptr, psv, C.c_int(y.size), C.c_int64(x.size), px, py,
                                  out.ctypes.data_as(_dp))
    return out


def rotation_broadening(flux, velocity_per_pix, v_rot, limb_darkening=0.6):
    """broadening.py:824-877 (flux only; wavelengths pass through)"""
    f, pf = _d(flux)
    out = np.empty_like(f)
    rc = lib().orc_rotation_broadening(
        C.c_int64(f.size), pf, C.c_double(velocity_per_pix), C.c_double(v_rot), C.c_double(limb_darkening), out.ctypes.data_as(_dp)
    )
    if rc:
        raise MemoryError
    return out


def gaussian_filter1d(values, sigma, truncate=4.0):
    """scipy.ndimage.gaussian_filter1d(values, sigma) (mode='reflect'): docs/rotation_broadening cell 11"""
    v, pv = _d(values)
    out = np.empty_like(v)
    if lib().orc_gaussian_filter1d(C.c_int64(v.size), pv, C.c_double(sigma), C.c_double(truncate), out.ctypes.data_as(_dp)):
        raise MemoryError
    return out
