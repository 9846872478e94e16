"""Spectrum post-processing on the GPU: rotational broadening, call-compatible with
stardis/radiation_field/opacities/opacities_solvers/broadening.py:824-877.

The Gray rotation profile is a few hundred numbers built on the host exactly as the reference builds it; the
convolution with the whole spectrum (scipy.ndimage.convolve1d, mode='reflect') runs in k_convolve1d_reflect with
scipy's summation order, so results agree with the reference bit for bit on the golden vectors."""
import numpy as np

from . import constants as K
from ._lib import default_context


def _kms(x):
    if hasattr(x, "to"):
        try:
            from astropy import units as u

            return float(x.to(u.km / u.s).value)
        except Exception:  # pragma: no cover
            pass
    return float(getattr(x, "value", x))


def rotation_profile(velocity_per_pix, v_rot, limb_darkening=0.6):
    """Normalised rotational kernel (:853-864, :870)."""
    v_rot_by_c = np.maximum(1e-5, np.abs(v_rot)) / K.C_KMS
    half_width_pix = np.round(v_rot / velocity_per_pix).astype(int)
    profile_velocity = np.linspace(-half_width_pix, half_width_pix, 2 * half_width_pix + 1) * velocity_per_pix
    profile = np.maximum(0.0, 1.0 - (profile_velocity / v_rot) ** 2)
    kernel = (2 * (1 - limb_darkening) * profile**0.5 + 0.5 * K.PI * limb_darkening * profile) / (
        K.PI * v_rot_by_c * (1 - limb_darkening / 3)
    )
    return kernel / kernel.sum()


def convolve1d_reflect(values, weights, ctx=None):
    """scipy.ndimage.convolve1d(values, weights) (mode='reflect') on the GPU."""
    ctx = ctx or default_context()
    v = np.ascontiguousarray(values, dtype=np.float64).reshape(-1)
    w = np.ascontiguousarray(weights, dtype=np.float64).reshape(-1)
    if w.size % 2 == 0:
        raise ValueError("only odd kernel lengths are supported")
    h = w.size // 2
    symmetric = bool(np.all(np.abs(w[h + 1 :] - w[:h][::-1]) <= np.finfo(np.float64).eps))  # scipy's test
    d_v, d_w = ctx.upload(v), ctx.upload(w)
    out = ctx.empty(v.shape)
    ctx.call("sdx_convolve1d_reflect_dev", v.size, d_v.ptr, w.size, d_w.ptr, int(symmetric), out.ptr)
    return out.numpy()


def gaussian_kernel(sigma, truncate=4.0):
    """The weights scipy.ndimage.gaussian_filter1d builds (order 0): radius int(truncate * sigma + 0.5)."""
    sigma = float(sigma)
    radius = int(truncate * sigma + 0.5)
    x = np.arange(-radius, radius + 1)
    phi = np.exp(-0.5 / (sigma * sigma) * x**2)
    return phi / phi.sum()


def gaussian_filter1d(flux, sigma, truncate=4.0, ctx=None):
    """scipy.ndimage.gaussian_filter1d(flux, sigma) (mode='reflect') on the GPU: the instrumental line-spread function the
    reference's rotation-broadening walk-through applies before the rotational kernel (docs/rotation_broadening, cell 11;
    sigma in pixels = lambda / R / dispersion / 2.355)."""
    values = np.asarray(getattr(flux, "value", flux), dtype=np.float64)
    return convolve1d_reflect(values, gaussian_kernel(sigma, truncate), ctx)


def rotation_broadening(velocity_per_pix, wavelength, flux, v_rot=0.0, limb_darkening=0.6):
    """-> (wavelength, broadened flux).  Inputs may be astropy quantities (km/s, Angstrom, flux density) or plain
    numbers in those units; like the reference the flux comes back untouched when |v_rot| < 1e-5 km/s (:866-867)."""
    vpp, v = _kms(velocity_per_pix), _kms(v_rot)
    if np.abs(v) < 1e-5:
        return wavelength, flux
    weights = rotation_profile(vpp, v, limb_darkening)
    values = np.asarray(getattr(flux, "value", flux), dtype=np.float64)
    return wavelength, convolve1d_reflect(values, weights)
