"""alpha_line from VALD line lists on the GPU, call-compatible with stardis/plasma/base.py:178-455.

`AlphaLineVald.calculate` / `AlphaLineShortlistVald.calculate` take the reference's arguments and return the
reference's two DataFrames; the (N_l, N_d) table is evaluated by sdx_line_params_dev.  The host part is the
reference's per-line bookkeeping (column selection, degeneracies, unit conversions, the ionisation-energy merge,
the auto-ionisation filter), all O(N_l).

`deferred_line_list` is the other half of f1: it turns the SAME per-line table into a `stardis_amd.linelist.LineList`
so that the dense table is never formed and the line kernel's pre-pass generates alpha, gamma and the Doppler width
itself.
"""
import numpy as np

from stardis_amd import constants as K
from stardis_amd import linelist as LL
from stardis_amd._lib import plain

ALPHA_COEFFICIENT = K.ALPHA_COEFFICIENT  # plasma/base.py:35

_FULL_COLUMNS = ["atomic_number", "ion_number", "wavelength", "log_gf", "e_low", "e_up", "j_lo", "j_up", "rad", "stark", "waals"]
_SHORT_COLUMNS = ["atomic_number", "ion_number", "wavelength", "log_gf", "e_low", "rad", "stark", "waals"]


def wavelength_to_nu(wavelength_angstrom):
    """(wavelength * u.AA).to(u.Hz, equivalencies=u.spectral()) (:272-274): astropy divides by its m -> AA scale, then c / x."""
    return K.C_SI / (np.asarray(wavelength_angstrom, dtype=np.float64) / K.M_TO_ANGSTROM)


def upper_energy_from_wavelength(e_low_ev, wavelength_angstrom):
    """Short lists carry no upper level: e_up = e_low + h c / lambda in eV (:383-390)."""
    h_si, ev_j = 6.62607015e-34, 1.602176634e-19
    return np.asarray(e_low_ev, dtype=np.float64) + (h_si * K.C_SI / np.asarray(wavelength_angstrom, dtype=np.float64)) / (ev_j * 1e-10)


def population_table(keys, density, partition):
    """N / U per species: rows of (density / partition) in the order of `keys` (the left-merge at :254-258).
    -> (pop (n_keys, N_d), {key: row}).  A species absent from the plasma gives NaN rows, as the merge does."""
    ratio = density / partition
    rows, index = [], {}
    n_depth = ratio.shape[1]
    for k in keys:
        index[k] = len(rows)
        try:
            rows.append(np.asarray(ratio.loc[k], dtype=np.float64).reshape(n_depth))
        except KeyError:
            rows.append(np.full(n_depth, np.nan))
    return (np.array(rows) if rows else np.zeros((0, n_depth))), index


def _atom_lines(atomic_data, ionization_data, short):
    import pandas as pd

    cols = _SHORT_COLUMNS if short else _FULL_COLUMNS
    ll = atomic_data.linelist_atoms.rename(columns={"ion_charge": "ion_number"})[cols]
    ll = ll[ll.atomic_number <= atomic_data.selected_atomic_numbers.max()].copy()  # :238-241
    if short:
        ll["e_up"] = upper_energy_from_wavelength(ll.e_low.values, ll.wavelength.values)
    else:
        ll["g_lo"] = ll.j_lo * 2 + 1
        ll["g_up"] = ll.j_up * 2 + 1
        ll["f_lu"] = 10**ll.log_gf / ll.g_lo  # :268-270
    ll["nu"] = wavelength_to_nu(ll.wavelength.values)
    ion = ionization_data.reset_index()
    ion["ion_number"] -= 1  # :304-305 charge convention
    ll = pd.merge(ll, ion, how="left", on=["atomic_number", "ion_number"])
    ll["level_energy_lower"] = ll["e_low"].values * K.EV_TO_ERG_ASTROPY
    ll["level_energy_upper"] = ll["e_up"].values * K.EV_TO_ERG_ASTROPY
    ll["A_ul"] = 10 ** ll["rad"]  # :316-318
    return ll


def _alpha_inputs(ll, pop_keys, density, partition, short):
    pop, index = population_table(pop_keys, density, partition)
    row = np.array([index[k] for k in zip(*(ll[c].values for c in ("atomic_number", "ion_number")))], dtype=np.int32) if len(ll) else np.zeros(0, np.int32)
    strength = (10 ** ll.log_gf.values) if short else ll.f_lu.values
    g_lo = None if short else ll.g_lo.values
    return pop, row, strength, g_lo


def _dense_alphas(ll, pop, row, strength, g_lo, t_electrons):
    import pandas as pd

    t = np.asarray(plain(t_electrons), dtype=np.float64).reshape(-1)
    spec = LL.LineList(ll.nu.values, ll.e_low.values, strength, row, pop, np.ones(len(ll)), t, g_lo=g_lo)
    alphas, _, _ = LL.line_params(spec, gammas=False, doppler_widths=False)
    if np.any(np.isnan(alphas)) or np.any(np.isinf(np.abs(alphas))):
        raise ValueError("Some alpha_line from vald are nan, inf, -inf " " Something went wrong!")  # :298-301
    df = pd.DataFrame(alphas)
    df["nu"] = ll.nu.values
    return df


def _species_keys(ll):
    seen = dict.fromkeys(zip(ll.atomic_number.values.tolist(), ll.ion_number.values.tolist()))
    return list(seen)


class AlphaLine:
    """plasma/base.py:130-175: lines of the TARDIS atomic data.  The level populations and the stimulated-emission factor
    are TARDIS plasma properties (dense inputs, as in the reference); the gather and the products run in
    sdx_alpha_line_levels_dev."""

    outputs = ("alpha_line",)

    def calculate(self, lines, level_number_density, lines_lower_level_index, stimulated_emission_factor, f_lu):
        import pandas as pd

        from stardis_amd._lib import default_context

        ctx = default_context()
        levels = np.ascontiguousarray(plain(level_number_density), dtype=np.float64)
        index = np.asarray(plain(lines_lower_level_index))
        if index.size and (index.min() < -levels.shape[0] or index.max() >= levels.shape[0]):
            raise IndexError("index out of range")  # numpy take(..., mode="raise") (:157-159)
        index = np.where(index < 0, index + levels.shape[0], index).astype(np.int32)
        stim = np.ascontiguousarray(plain(stimulated_emission_factor), dtype=np.float64).reshape(index.size, levels.shape[1])
        d = [ctx.upload(levels), ctx.upload(index, np.int32), ctx.upload(stim), ctx.upload(np.ascontiguousarray(plain(f_lu), dtype=np.float64))]
        out = ctx.empty(stim.shape)
        ctx.call("sdx_alpha_line_levels_dev", index.size, levels.shape[1], levels.shape[0], d[0].ptr, d[1].ptr, d[2].ptr, d[3].ptr,
                 ALPHA_COEFFICIENT, out.ptr)
        alpha = out.numpy()
        if np.any(np.isnan(alpha)) or np.any(np.isinf(np.abs(alpha))):
            raise ValueError("Some alpha_line are nan, inf, -inf " " Something went wrong!")  # :162-165
        df = pd.DataFrame(alpha, index=lines.index, columns=np.array(level_number_density.columns))
        df["nu"] = lines.nu
        return df


class AlphaLineVald:
    """plasma/base.py:178-321."""

    outputs = ("alpha_line_from_linelist", "lines_from_linelist")
    short = False

    def calculate(self, atomic_data, ion_number_density, t_electrons, ionization_data, partition_function):
        ll = _atom_lines(atomic_data, ionization_data, self.short)
        pop, row, strength, g_lo = _alpha_inputs(ll, _species_keys(ll), ion_number_density, partition_function, self.short)
        alphas = _dense_alphas(ll, pop, row, strength, g_lo, t_electrons)
        if self.short:
            return alphas, ll
        valid = ll.level_energy_upper < ll.ionization_energy  # auto-ionising lines cannot be broadened (:320-321)
        return alphas[valid], ll[valid]


class AlphaLineShortlistVald(AlphaLineVald):
    """plasma/base.py:324-455: no upper level or degeneracies in the list; g cancels between n_lower and f_lu."""

    short = True


def deferred_line_list(lines, tracing_nus, stellar_model, stellar_plasma, broadening_config, use_vald_broadening,
                       density=None, partition=None):
    """A LineList for the lines of `lines` (a lines_from_linelist-style table) that fall on the tracing grid, sorted
    by frequency, with the auto-ionisation filter of calc_alpha_line_at_nu (opacities_solvers/base.py:392-421) and the
    broadening description of calculate_broadening (broadening.py:659-732)."""
    nus = np.asarray(plain(tracing_nus), dtype=np.float64)
    sel = lines.sort_values("nu")
    sel = sel[sel.nu.between(nus.min(), nus.max())]
    if not use_vald_broadening:
        sel = sel[~(sel.level_energy_upper > sel.ionization_energy).values]
    short = "f_lu" not in sel.columns
    density = stellar_plasma.ion_number_density if density is None else density
    partition = stellar_plasma.partition_function if partition is None else partition
    pop, row, strength, g_lo = _alpha_inputs(sel, _species_keys(sel), density, partition, short)
    cfg = broadening_config
    flags = LL.broadening_flags("linear_stark" in cfg, "quadratic_stark" in cfg, "van_der_waals" in cfg, "radiation" in cfg)
    xi = stellar_model.microturbulence
    return LL.LineList(
        sel.nu.values, sel.e_low.values, strength, row, pop,
        stellar_model.composition.nuclide_masses.loc[sel.atomic_number].values, plain(stellar_model.temperatures), g_lo=g_lo,
        microturbulence=float(xi.cgs.value) if hasattr(xi, "cgs") else float(xi),
        gamma_mode=LL.GAMMA_VALD if use_vald_broadening else LL.GAMMA_CLASSIC, flags=flags,
        atomic_number=sel.atomic_number.values, ion_number=sel.ion_number.values + 1,
        ionization_energy=sel.ionization_energy.values, upper_energy=sel.level_energy_upper.values,
        lower_energy=sel.level_energy_lower.values, A_ul=sel.A_ul.values,
        stark=sel.stark.values if "stark" in sel.columns else None, waals=sel.waals.values if "waals" in sel.columns else None,
        electron_density=plain(stellar_plasma.electron_densities), h_density=plain(stellar_plasma.ion_number_density.loc[1, 0]),
    )
