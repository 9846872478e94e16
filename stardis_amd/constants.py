"""Physical constants exactly as the reference evaluates them (astropy CODATA 2018, cgs).

The reference builds these at import time from ``astropy.constants``
(opacities_solvers/base.py:20-34, broadening.py:16-26, blackbody.py:5-7,
voigt.py:12-14).  astropy is not a dependency of this package, so the float64
values are pinned here; ``tests/golden/make_golden.py`` re-derives every one of
them from the reference modules and ``tests/test_constants.py`` checks the
committed dump bit-for-bit.
"""
import math

H_CGS = 6.62607015e-27
C_CGS = 29979245800.0
K_B_CGS = 1.380649e-16
M_E_CGS = 9.1093837015e-28
M_P_CGS = 1.67262192369e-24  # broadening.py:24 "H_MASS" is the proton mass
AMU_CGS = 1.6605390666e-24
E_ESU = 4.803204712570263e-10
BOHR_RADIUS = 5.2917721090299995e-09
SIGMA_T = 6.6524587321000005e-25
RYDBERG_FREQUENCY = 3289841960250881.0  # opacities_solvers/base.py:34
RYDBERG_ENERGY = 2.1798723611035848e-11  # broadening.py:20
BF_CONSTANT = 2.815403624709817e29  # opacities_solvers/base.py:21-27
FF_CONSTANT = 369234910.67735106  # opacities_solvers/base.py:28-33
VACUUM_ELECTRIC_PERMITTIVITY = 1.0 / (4.0 * math.pi)  # broadening.py:23
C_KMS = 299792.458
PI = math.pi
SQRT_PI = math.sqrt(math.pi)  # 1.7724538509055159, voigt.py:12
ALPHA_COEFFICIENT = 0.026540088545744744  # pi e^2 / (m_e c), plasma/base.py:35
EV_CGS = 1.602176634e-12
EV_TO_ERG_ASTROPY = 1.6021766340000001e-12  # the scale astropy applies in (e * u.eV).cgs (plasma/base.py:308-309); pinned by g11
C_SI = 299792458.0
# astropy's metre -> Angstrom scale, 1 / (0.1 * 1e-9): tracing_nus.to(u.AA, u.spectral()) evaluates
# (c / nu) * this (opacities_solvers/base.py:62); the golden vectors pin it bit-for-bit.
M_TO_ANGSTROM = 1.0 / (0.1 * 1e-9)
# astropy's nm -> Angstrom scale is not 10: (x * u.nm).to(u.AA) multiplies by this (util.py:43).  It matters: the H2+ bf
# table's nodes move by an ulp, and with them Qhull's choice of cell diagonals (2 % on individual cross-sections).
NM_TO_ANGSTROM = 9.999999999999998


def nu_to_angstrom(nus):
    return (C_SI / nus) * M_TO_ANGSTROM

ALL = {
    k: v
    for k, v in dict(globals()).items()
    if k.isupper() and isinstance(v, float)
}
