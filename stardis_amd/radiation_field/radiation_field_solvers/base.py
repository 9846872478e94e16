"""LTE formal solution on MI355X, call-compatible with
stardis/radiation_field/radiation_field_solvers/base.py (van Noort 2002 eq. 14 short characteristics).

The per-angle, per-frequency depth recurrence runs in one HIP kernel (k_raytrace in
stardis_amd/csrc/sdx_kernels.h); this module builds the ray-length table and moves arrays."""
import numpy as np

from stardis_amd import ops
from stardis_amd._lib import default_context, plain


def calc_weights_parallel(delta_tau):
    """w0, w1, w2 of eq. 14 for an array of optical depths.  Reference :6-47."""
    return ops.calc_weights_parallel(delta_tau)


calc_weights = calc_weights_parallel  # the reference keeps an unused numpy twin (:50-82) with the same contract


def _source_plane(source_function, tracing_nus, temps):
    """None for the Planck function (evaluated inside the kernel), else the caller's source function evaluated on the host
    the way the reference calls it — source_function(tracing_nus, temps) with temps shaped (N_d, 1), :133 — as a plain
    (N_d, N_nu) array for the kernel to read."""
    if source_function is None or getattr(source_function, "__name__", "") == "blackbody_flux_at_nu":
        return None
    nus = np.asarray(plain(tracing_nus), dtype=np.float64).reshape(-1)
    t = np.asarray(plain(temps), dtype=np.float64).reshape(-1, 1)
    plane = np.asarray(plain(source_function(nus, t)), dtype=np.float64)
    return np.ascontiguousarray(np.broadcast_to(plane, (t.size, nus.size)))


def single_theta_trace_parallel(ray_dist_to_next_depth_point, temps, alphas, tracing_nus, source_function=None,
                                inward_rays=False):
    """Specific intensity (N_d, N_nu) along one ray direction.  Reference :85-268; inward_rays adds the
    surface-to-centre sweep of spherical geometry (:141-198) before the outward pass."""
    rd = np.asarray(plain(ray_dist_to_next_depth_point), dtype=np.float64).reshape(-1, 1)
    _, I = ops.raytrace_arrays(tracing_nus, temps, rd, np.ones(1), alphas, track=True, inward_rays=bool(inward_rays), want_flux=False,
                               source=_source_plane(source_function, tracing_nus, temps))
    return I[:, :, 0]


def calculate_spherical_ray(thetas, depth_points_radii):
    """Chord length of each ray through each shell (N_d-1, N_theta).  Reference :349-381.  Geometry set-up,
    a (N_d x N_theta) table built once on the host."""
    thetas = np.asarray(thetas, dtype=np.float64)
    r = np.asarray(plain(depth_points_radii), dtype=np.float64)
    out = np.zeros((len(r) - 1, len(thetas)))
    for k, theta in enumerate(thetas):
        b = r[-1] * np.sin(theta)
        with np.errstate(invalid="ignore"):
            z = np.sqrt(r**2 - b**2)
        dz = np.diff(z)
        ok = ~np.isnan(dz)
        out[ok, k] = dz[ok]
    return out


def raytrace(stellar_model, stellar_radiation_field):
    """Trace every angle and accumulate the Gauss-Legendre flux sum into stellar_radiation_field.F_nu
    (in place, like the reference :324-338).  Fills I_nus when track_individual_intensities is set."""
    field = stellar_radiation_field
    thetas = np.asarray(field.thetas, dtype=np.float64)
    correction = 1.0
    if stellar_model.spherical:  # :296-300, :340-344
        radii = np.asarray(plain(stellar_model.geometry.r), dtype=np.float64)
        ray_distances = calculate_spherical_ray(thetas, radii)
        correction = (radii[-1] / float(plain(stellar_model.geometry.reference_r))) ** 2
    else:
        dist = np.asarray(plain(stellar_model.geometry.dist_to_next_depth_point), dtype=np.float64)
        ray_distances = dist.reshape(-1, 1) / np.cos(thetas)  # :302-305
    ctx = default_context()
    opac = field.opacities
    alphas = opac.total_alphas_device(ctx) if hasattr(opac, "total_alphas_device") else opac.total_alphas
    track = bool(getattr(field, "track_individual_intensities", False))
    # F_nu is accumulated into (:336); a freshly created field holds zeros, which need no upload
    f_in = field.F_nu if np.any(field.F_nu) else None
    F, I = ops.raytrace_arrays(
        field.frequencies, plain(stellar_model.temperatures), ray_distances, field.I_nus_weights, alphas, F_nu=f_in,
        track=track, ctx=ctx, inward_rays=bool(stellar_model.spherical), photospheric_correction=correction,
        source=_source_plane(getattr(field, "source_function", None), field.frequencies, stellar_model.temperatures),
    )
    field.F_nu[...] = F
    if track:
        field.I_nus[...] = I
    return field.F_nu
