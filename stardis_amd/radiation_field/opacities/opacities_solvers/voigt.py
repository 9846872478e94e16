"""Faddeeva / Voigt element-wise functions on the GPU, call-compatible with
stardis/radiation_field/opacities/opacities_solvers/voigt.py (faddeeva :89-91, voigt_profile :153-155).

The approximation is the reference's Humlicek-W4 four-region rational form — not an exact
Faddeeva function — because flux parity is defined against it."""
from stardis_amd import ops


def faddeeva(z):
    return ops.faddeeva(z)


def voigt_profile(delta_nu, doppler_width, gamma):
    return ops.voigt_profile(delta_nu, doppler_width, gamma)
