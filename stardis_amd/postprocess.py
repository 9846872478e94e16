"""Spectrum post-processing on the GPU: rotational broadening, call-compatible with
stardis/radiation_field/opacities/opacities_solvers/broadening.py:824-877.

The Gray rotation profile is a few hundred numbers built on the host exactly as the reference builds it; the
convolution with the whole spectrum (scipy.ndimage.convolve1d, mode='reflect') runs in k_convolve1d_reflect with
scipy's summation order, so results agree with the reference bit for bit on the golden vectors."""
import numpy as np

from . import constants as K
from ._lib import default_context, ptr_of


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


def convolve1d_reflect_device(d_values, n, weights, ctx=None):
    """scipy.ndimage.convolve1d(values, weights) (mode='reflect') for n values already in HBM (a DeviceArray, a CUDA
    tensor or a raw device address) -> DeviceArray.  Nothing but the few-hundred-number kernel crosses PCIe."""
    ctx = ctx or default_context()
    w = np.ascontiguousarray(weights, dtype=np.float64).reshape(-1)
    if w.size % 2 == 0:
        raise ValueError("only odd kernel lengths are supported")
    h = w.size // 2
    symmetric = bool(np.all(np.abs(w[h + 1 :] - w[:h][::-1]) <= np.finfo(np.float64).eps))  # scipy's test
    d_w = ctx.upload(w)
    out = ctx.empty((int(n),))
    ctx.call("sdx_convolve1d_reflect_dev", int(n), d_values if isinstance(d_values, int) else ptr_of(d_values), w.size, d_w.ptr,
             int(symmetric), out.ptr)
    return out


def convolve1d_reflect(values, weights, ctx=None):
    """scipy.ndimage.convolve1d(values, weights) (mode='reflect') on the GPU."""
    ctx = ctx or default_context()
    v = np.ascontiguousarray(values, dtype=np.float64).reshape(-1)
    return convolve1d_reflect_device(ctx.upload(v), v.size, weights, ctx).numpy()


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


class DeviceSpectrum:
    """The emergent spectrum kept in HBM from the formal solution to the broadened result (BASELINE configs[4]): what the
    reference's walk-through does on the host with sim.spectrum_lambda (docs/rotation_broadening cells 7-19) —
    F_lambda = F_nu[-1] nu / lambda (stardis/base.py:137-141), scipy.ndimage.gaussian_filter1d for the instrumental
    line-spread function, rotation_broadening (broadening.py:824-877) — as three small kernels behind the synthesis."""

    def __init__(self, synthesizer):
        """synthesizer: a stardis_amd.engine.SpectralSynthesizer that owns the whole grid (shard = everything)."""
        syn = synthesizer
        if syn.begin != 0 or syn.count != syn.n_nu:
            raise ValueError("post-processing needs the whole spectrum: gather the shards first")
        self.syn, self.ctx, self.n = syn, syn.ctx, syn.n_nu
        self.lambdas = K.nu_to_angstrom(syn.nus_host)
        self.d_lambdas = self.ctx.upload(self.lambdas)
        self.d_spectrum = self.ctx.empty((self.n,))

    def spectrum_lambda(self):
        """-> DeviceArray (N_nu,): F_lambda of the outermost depth point."""
        syn = self.syn
        row = syn.flux_ptr + 8 * (syn.n_depth - 1) * syn.count  # F_nu[-1]
        self.ctx.call("sdx_flux_nu_to_lambda_dev", self.n, row, syn.d_nus.ptr, self.d_lambdas.ptr, self.d_spectrum.ptr)
        return self.d_spectrum

    def broadened(self, sigma_pix=None, velocity_per_pix=None, v_rot=0.0, limb_darkening=0.6):
        """spectrum_lambda -> gaussian_filter1d(sigma_pix) (the instrumental line-spread function, cells 7-11: sigma =
        lambda / R / dispersion / 2.355 pixels) -> rotation_broadening(velocity_per_pix, ..., v_rot) (cells 15-19).
        Either stage is skipped when its parameter is None / |v_rot| < 1e-5 km/s (:866-867).  Returns the DeviceArray."""
        d = self.spectrum_lambda()
        if sigma_pix:
            d = convolve1d_reflect_device(d, self.n, gaussian_kernel(float(sigma_pix)), self.ctx)
        if velocity_per_pix is not None and abs(_kms(v_rot)) >= 1e-5:
            d = convolve1d_reflect_device(d, self.n, rotation_profile(_kms(velocity_per_pix), _kms(v_rot), limb_darkening), self.ctx)
        return d
