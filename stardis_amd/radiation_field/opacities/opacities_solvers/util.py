"""Cross-section tables and density look-ups feeding calc_alpha_file; host-side I/O that mirrors
stardis/radiation_field/opacities/opacities_solvers/util.py (sigma_file :14-108, get_number_density :111-166).

The tables are a few hundred numbers read once per run.  The 2-D tables (H- ff, H2+ bf) are interpolated
with the same scipy LinearNDInterpolator call the reference makes, because that interpolant's Delaunay
diagonals are implementation-defined (SURVEY hazard 4); the 1-D H- bf table is returned raw and
interpolated on the GPU (np.interp semantics)."""
import logging

import numpy as np

from stardis_amd import constants as K
from stardis_amd.util import species_string_to_tuple

logger = logging.getLogger(__name__)


def read_table(fpath, opacity_source):
    """-> ("1d", wavelength_AA, sigma) or ("2d", wavelength_AA, second_axis, values, scale_kind)"""
    import pandas as pd

    if opacity_source == "Hminus_bf":
        tab = pd.read_csv(fpath, header=None, comment="#", names=["wavelength", "cross_section"])
        return "1d", tab.wavelength.to_numpy(dtype=float), tab.cross_section.to_numpy(dtype=float)
    if opacity_source == "Hminus_ff":
        tab = pd.read_csv(fpath, delimiter=r"\s+", comment="#")
        tab.columns = tab.columns.str.strip(",")
        wave = tab[tab.columns[0]].to_numpy(dtype=float)
        theta = tab.columns[1:].astype(float).to_numpy()
        return "2d", wave, theta, tab.to_numpy()[:, 1:].astype(float)
    if opacity_source == "H2plus_bf":
        tab = pd.read_csv(fpath, delimiter=r"\s+", index_col=0, comment="#")
        tab = tab.replace({"-": "e-"}, regex=True).astype(float)
        wave = tab.index.to_numpy(dtype=float) * 10.0  # nm -> Angstrom
        temps = tab.columns.to_numpy().astype(int)
        return "2d", wave, temps, tab.to_numpy()
    raise ValueError(f"Unknown opacity_source: {opacity_source}")


def _interp2d(wave, axis2, values, lambdas, second):
    from scipy.interpolate import LinearNDInterpolator

    w_mesh, a_mesh = np.meshgrid(wave, axis2, indexing="ij")
    f = LinearNDInterpolator(np.vstack([w_mesh.ravel(), a_mesh.ravel()]).T, values.flatten(), fill_value=0)
    lam, sec = np.meshgrid(lambdas, second)
    return f(lam, sec)


def sigma_file(tracing_lambdas, temperatures, fpath, opacity_source=None):
    """Cross-sections (N_T, N_lambda) for the 2-D tables, (N_lambda,) for Hminus_bf — as util.py:14-108."""
    tracing_lambdas = np.asarray(tracing_lambdas, dtype=float)
    temperatures = np.asarray(temperatures, dtype=float)
    table = read_table(fpath, opacity_source)
    if table[0] == "1d":
        return np.interp(tracing_lambdas, table[1], table[2])
    _, wave, axis2, values = table
    if opacity_source == "Hminus_ff":
        sig = _interp2d(wave, axis2, values, tracing_lambdas, 5040 / temperatures)
        sig = sig * 1e-26 * K.K_B_CGS * temperatures[:, np.newaxis]
        what = "H- FF"
    else:
        sig = _interp2d(wave, axis2, values, tracing_lambdas, temperatures) * 1e-18
        what = "H2+ BF"
    if np.any(sig == 0):
        logger.warning(
            "Outside of interpolation range for %s cross-sections at depth points %s. Assuming 0 opacity there.",
            what, np.unique(np.where(sig == 0)[0]),
        )
    return sig


def get_number_density(stellar_plasma, opacity_source):
    """(number_density, atomic_number, ion_number) for an opacity-source string — util.py:111-166."""
    ions = stellar_plasma.ion_number_density
    n_e = stellar_plasma.electron_densities
    fixed = {
        "Hminus_bf": lambda: stellar_plasma.h_minus_density,
        "Hminus_ff": lambda: ions.loc[1, 0] * n_e,
        "Heminus_ff": lambda: ions.loc[2, 0] * n_e,
        "H2minus_ff": lambda: stellar_plasma.h2_density * n_e,
        "H2plus_ff": lambda: ions.loc[1, 0] * ions.loc[1, 1],
        "H2plus_bf": lambda: stellar_plasma.h2_plus_density,
    }
    if opacity_source in fixed:
        return fixed[opacity_source](), None, None
    species, kind = opacity_source[:-3], opacity_source[-2:]
    atomic_number, ion_number = species_string_to_tuple(species.replace("_", " "))
    density = 1
    if kind == "ff":
        ion_number += 1
        density = density * n_e
    density = density * ions.loc[atomic_number, ion_number]
    return density, atomic_number, ion_number
