"""alpha_line of molecular VALD lines on the GPU, call-compatible with stardis/plasma/molecules.py:192-440
(AlphaLineValdMolecule, AlphaLineShortlistValdMolecule).  Same structure as stardis_amd.plasma.base: O(N_l) host
bookkeeping, the (N_l, N_d) table from sdx_line_params_dev, and `deferred_molecule_line_list` for the fused path."""
import numpy as np

from stardis_amd import constants as K
from stardis_amd import linelist as LL
from stardis_amd._lib import plain
from stardis_amd.plasma.base import _dense_alphas, population_table, upper_energy_from_wavelength, wavelength_to_nu

ALPHA_COEFFICIENT = K.ALPHA_COEFFICIENT

_FULL_COLUMNS = ["molecule", "wavelength", "log_gf", "e_low", "e_up", "j_lo", "j_up", "rad", "stark", "waals"]
_SHORT_COLUMNS = ["molecule", "wavelength", "log_gf", "e_low", "rad", "stark", "waals"]


def _molecule_lines(atomic_data, short):
    ll = atomic_data.linelist_molecules[_SHORT_COLUMNS if short else _FULL_COLUMNS].copy()
    if short:
        ll["e_up"] = upper_energy_from_wavelength(ll.e_low.values, ll.wavelength.values)  # :373-380
    else:
        ll["g_lo"] = ll.j_lo * 2 + 1
        ll["g_up"] = ll.j_up * 2 + 1
        ll["f_lu"] = 10**ll.log_gf / ll.g_lo  # :271-273
    ll["nu"] = wavelength_to_nu(ll.wavelength.values)
    ll["level_energy_lower"] = ll["e_low"].values * K.EV_TO_ERG_ASTROPY
    ll["level_energy_upper"] = ll["e_up"].values * K.EV_TO_ERG_ASTROPY
    ll["A_ul"] = 10 ** ll["rad"]
    return ll


def _alpha_inputs(ll, density, partition, short):
    keys = list(dict.fromkeys(ll.molecule.values.tolist()))
    pop, index = population_table(keys, density, partition)
    row = np.array([index[m] for m in ll.molecule.values], dtype=np.int32) if len(ll) else np.zeros(0, np.int32)
    return pop, row, ((10 ** ll.log_gf.values) if short else ll.f_lu.values), (None if short else ll.g_lo.values)


class AlphaLineValdMolecule:
    """plasma/molecules.py:192-320."""

    outputs = ("molecule_alpha_line_from_linelist", "molecule_lines_from_linelist")
    short = False

    def calculate(self, atomic_data, molecule_number_density, t_electrons, molecule_partition_function):
        ll = _molecule_lines(atomic_data, self.short)
        pop, row, strength, g_lo = _alpha_inputs(ll, molecule_number_density, molecule_partition_function, self.short)
        return _dense_alphas(ll, pop, row, strength, g_lo, t_electrons), ll


class AlphaLineShortlistValdMolecule(AlphaLineValdMolecule):
    """plasma/molecules.py:322-440."""

    short = True


def deferred_molecule_line_list(lines, tracing_nus, stellar_model, stellar_plasma, broadening_config, density=None, partition=None):
    """LineList of the molecular lines on the grid (calc_molecular_alpha_line_at_nu, opacities_solvers/base.py:444-484;
    calculate_molecule_broadening, broadening.py:735-821: gamma = A_ul as one column, or zero)."""
    nus = np.asarray(plain(tracing_nus), dtype=np.float64)
    sel = lines.sort_values("nu")
    sel = sel[sel.nu.between(nus.min(), nus.max())]
    short = "f_lu" not in sel.columns
    density = stellar_plasma.molecule_number_density if density is None else density
    partition = stellar_plasma.molecule_partition_function if partition is None else partition
    pop, row, strength, g_lo = _alpha_inputs(sel, density, partition, short)
    ions = stellar_plasma.molecule_ion_map.loc[sel.molecule]
    masses = stellar_model.composition.nuclide_masses
    xi = stellar_model.microturbulence
    return LL.LineList(
        sel.nu.values, sel.e_low.values, strength, row, pop, masses.loc[ions.Ion1].values + masses.loc[ions.Ion2].values,
        plain(stellar_model.temperatures), g_lo=g_lo, microturbulence=float(xi.cgs.value) if hasattr(xi, "cgs") else float(xi),
        gamma_mode=LL.GAMMA_RADIATION_ONLY if "radiation" in broadening_config else LL.GAMMA_ZERO, A_ul=sel.A_ul.values,
    )
