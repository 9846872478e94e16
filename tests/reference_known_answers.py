"""The known-answer cases of the reference's own unit tests for this path, restated as data
(opacities_solvers/tests/test_broadening.py:40-75, :146-180, :251-285, :356-407, :494-536; tests/test_voigt.py:22-37,
:111-181).  Each entry: (function name, arguments, expected) with the reference's np.allclose criterion."""
import numpy as np

from stardis_amd import constants as K

_c4_prefactor = (K.E_ESU**2 * K.BOHR_RADIUS**3) / (36.0 * K.H_CGS * K.VACUUM_ELECTRIC_PERMITTIVITY)  # test_broadening.py:356-358
_ne_quadratic = 1.0e-19 / K.K_B_CGS * (36 * _c4_prefactor) ** (-2.0 / 3.0)
_t_vdw = np.pi / 8 / K.K_B_CGS / 17 ** (1.0 / 0.3)
_nh_vdw = (3.0 * 6.46e-34) ** (-0.4)


def _twice(*scalars):
    return tuple(np.array(2 * [s]) for s in scalars)


BROADENING = [
    ("doppler_width", (K.C_CGS, 0.5, K.K_B_CGS, 0.0), 1.0),
    ("doppler_width", _twice(K.C_CGS, 0.5, K.K_B_CGS, 0.0), np.array([1.0, 1.0])),
    ("n_effective", (1.0, K.RYDBERG_ENERGY, 0.0), 1.0),
    ("n_effective", _twice(1, K.RYDBERG_ENERGY, 0.0), np.array([1.0, 1.0])),
    ("gamma_linear_stark", (1.0, 0.0, (0.60 * 0.642) ** (-3 / 2)), 1.0),
    ("gamma_linear_stark", _twice(1.0, 0.0, (0.60 * 0.642) ** (-3 / 2)), np.array([1.0, 1.0])),
    ("gamma_quadratic_stark", (1, 1.0, 0.0, _ne_quadratic, 1.0), 1.0),
    ("gamma_quadratic_stark", _twice(1, 1.0, 0.0, _ne_quadratic, 1.0), np.array([1.0, 1.0])),
    ("gamma_van_der_waals", (1, 1.0, 0.0, _t_vdw, _nh_vdw), 13582529.79905836),
    ("gamma_van_der_waals", _twice(1, 1.0, 0.0, _t_vdw, _nh_vdw), np.array(2 * [13582529.79905836])),
]

FADDEEVA = [(0, 1 + 0j), (0.0, 1.0 + 0.0j), (np.array([0.0]), np.array([1.0 + 0.0j])), (np.array([0, 0]), np.array([1 + 0j, 1 + 0j]))]

VOIGT = [
    ((0, 1, 0), 1 / np.sqrt(np.pi)),
    ((0, 2, 0), 1 / (np.sqrt(np.pi) * 2)),
    ((np.array([0, 0]), np.array([1, 2]), np.array([0, 0])), np.array([1 / np.sqrt(np.pi), 1 / (np.sqrt(np.pi) * 2)])),
]

VOIGT_DIVISION_BY_ZERO = [-100, -5, -1, 0, 0.0, 1.2, 3, 100, np.array([0, -1.0, 1])]  # delta_nu and gamma values, doppler width 0
