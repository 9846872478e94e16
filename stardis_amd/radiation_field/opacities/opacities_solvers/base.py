"""Opacity assembly on MI355X, call-compatible with
stardis/radiation_field/opacities/opacities_solvers/base.py.

Host code here only unpacks plasma / model / config objects into flat arrays; every opacity value is
computed by a HIP kernel behind include/stardis_hip.h.  Returned arrays are numpy (host) arrays like
the reference's; their device twins are kept so that the total and the raytrace do not re-upload them.
"""
import logging
from pathlib import Path

import numpy as np

from stardis_amd import constants as K
from stardis_amd import ops
from stardis_amd._lib import DeviceArray, default_context, plain
from stardis_amd.radiation_field.opacities.opacities_solvers.broadening import (
    calculate_broadening,
    calculate_molecule_broadening,
)
from stardis_amd.radiation_field.opacities.opacities_solvers.util import (
    get_number_density,
    read_table,
    sigma_file,
    sigma_file_device,
)

logger = logging.getLogger(__name__)

VACUUM_ELECTRIC_PERMITTIVITY = K.VACUUM_ELECTRIC_PERMITTIVITY
BF_CONSTANT = K.BF_CONSTANT
FF_CONSTANT = K.FF_CONSTANT
RYDBERG_FREQUENCY = K.RYDBERG_FREQUENCY


def _nus_of(tracing_nus):
    return np.ascontiguousarray(plain(tracing_nus), dtype=np.float64).reshape(-1)


def _lambdas_angstrom(nus):
    """tracing_nus.to(u.AA, u.spectral()).value (:62): astropy divides c [m/s] by nu, then scales m -> AA."""
    return K.nu_to_angstrom(nus)


def _n_depth(stellar_model):
    return int(stellar_model.no_of_depth_points)


class _Result(np.ndarray):
    """ndarray that remembers the device buffer it was downloaded from (consumed by Opacities)."""

    _device = None


def _download(dev):
    host = dev.numpy().view(_Result)
    host._device = dev
    return host


def device_twin(array):
    return getattr(array, "_device", None)


# ------------------------------------------------------------------------------------------------ continuum
def calc_alpha_file(stellar_plasma, stellar_model, tracing_nus, opacity_source, fpath):
    """Tabulated cross-section x number density -> (N_d, N_nu).  Reference :40-70."""
    nus = _nus_of(tracing_nus)
    lambdas = _lambdas_angstrom(nus)
    density, _, _ = get_number_density(stellar_plasma, opacity_source)
    density = plain(density).astype(np.float64)
    table = read_table(Path(fpath), opacity_source)
    if table[0] == "1d":
        return _download(ops.alpha_file_1d(lambdas, table[1], table[2], density))
    sigmas = sigma_file_device(lambdas, plain(stellar_model.temperatures), Path(fpath), opacity_source)
    return _download(ops.alpha_file_2d(sigmas, density))


def calc_alpha_rayleigh(stellar_plasma, stellar_model, tracing_nus, species):
    """Rayleigh scattering by H, He, H2 -> (N_d, N_nu).  Reference :74-135, including its side effect:
    frequencies above 2.3e15 Hz are set to 0 in the caller's array (:99)."""
    picks = {}
    if "H" in species:
        picks["n_h"] = plain(stellar_plasma.ion_number_density.loc[1, 0])
    if "He" in species:
        picks["n_he"] = plain(stellar_plasma.ion_number_density.loc[2, 0])
    if "H2" in species:
        picks["n_h2"] = plain(stellar_plasma.h2_density)
    dev, clipped = ops.alpha_rayleigh(_nus_of(tracing_nus), _n_depth(stellar_model), **picks)
    target = tracing_nus.value if hasattr(tracing_nus, "unit") else tracing_nus
    if isinstance(target, np.ndarray) and target.flags.writeable and target.shape == clipped.shape:
        target[...] = clipped
    return _download(dev)


def calc_alpha_electron(stellar_plasma, stellar_model, tracing_nus, disable_electron_scattering=False):
    """Thomson scattering -> (N_d, N_nu), or the scalar 0 when disabled.  Reference :139-174."""
    if disable_electron_scattering:
        return 0
    return _download(ops.alpha_electron(len(tracing_nus), plain(stellar_plasma.electron_densities.values)))


def _bf_levels(stellar_plasma, species):
    """Flatten the plasma tables into per-level arrays, species-major, levels in plasma order (:204-226)."""
    offsets, ions, cutoffs, densities = [0], [], [], []
    for spec in species:
        _, atomic_number, ion_number = get_number_density(stellar_plasma, spec + "_bf")
        ionization_energy = float(stellar_plasma.ionization_data.loc[(atomic_number, ion_number + 1)])
        for level in stellar_plasma.levels:
            if level[0] == atomic_number and level[1] == ion_number:
                cutoffs.append((ionization_energy - float(stellar_plasma.excitation_energy.loc[level])) / K.H_CGS)
                densities.append(np.asarray(stellar_plasma.level_number_density.loc[level], dtype=np.float64))
        offsets.append(len(cutoffs))
        ions.append(ion_number)
    return offsets, ions, cutoffs, densities


def calc_alpha_bf(stellar_plasma, stellar_model, tracing_nus, species):
    """Hydrogenic bound-free opacity -> (N_d, N_nu).  Reference :178-271 (its per-frequency Python loop at
    :227-231 becomes one kernel)."""
    n_depth = _n_depth(stellar_model)
    offsets, ions, cutoffs, densities = _bf_levels(stellar_plasma, list(species.keys()) if hasattr(species, "keys") else list(species))
    level_density = np.vstack(densities) if densities else np.zeros((0, n_depth))
    return _download(ops.alpha_bf(_nus_of(tracing_nus), offsets, ions, cutoffs, level_density, n_depth))


def calc_alpha_ff(stellar_plasma, stellar_model, tracing_nus, species):
    """Hydrogenic free-free opacity -> (N_d, N_nu).  Reference :274-317."""
    ions, dens = [], []
    for spec in (species.keys() if hasattr(species, "keys") else species):
        number_density, _, ion_number = get_number_density(stellar_plasma, spec + "_ff")
        ions.append(ion_number)
        dens.append(plain(number_density).astype(np.float64))
    temps = plain(stellar_model.temperatures)
    nd = np.vstack(dens) if dens else np.zeros((0, temps.size))
    return _download(ops.alpha_ff(_nus_of(tracing_nus), temps, ions, nd))


def gaunt_times_departure(tracing_nus, temperatures, gaunt_fpath, departure_fpath):
    """Placeholder in the reference as well (:320-324)."""
    return None


# ------------------------------------------------------------------------------------------------ lines
def calc_alan_entries(no_of_depth_points, tracing_nus_values, line_nus, doppler_widths, gammas, alphas_array):
    """Line opacity at every (depth, frequency): the reference's hot loop (:487-592) as a HIP gather kernel."""
    return ops.calc_alan_entries(no_of_depth_points, tracing_nus_values, line_nus, doppler_widths, gammas, alphas_array)


def _in_grid(table, nus):
    ordered = table.sort_values("nu")
    return ordered[ordered.nu.between(nus.min(), nus.max())]


def _atomic_line_table(stellar_plasma):
    """TARDIS `lines` joined with ionisation and level energies (:366-390)."""
    import pandas as pd

    lines = stellar_plasma.lines.reset_index()
    ionization = stellar_plasma.ionization_data.reset_index()
    ionization["ion_number"] -= 1
    lines = pd.merge(lines, ionization, how="left", on=["atomic_number", "ion_number"])
    energies = stellar_plasma.atomic_data.levels.energy
    for side in ("lower", "upper"):
        lines = pd.merge(
            lines, energies, how="left",
            left_on=["atomic_number", "ion_number", f"level_number_{side}"],
            right_on=["atomic_number", "ion_number", "level_number"],
        ).rename(columns={"energy": f"level_energy_{side}"})
    return lines


RETURN_BROADENING_TABLES = True  # the reference returns gammas and doppler_widths (N_l, N_d); set False to skip forming them


def _line_opacity_from_list(spec, nus):
    """(alpha_line_at_nu, gammas, doppler_widths) for a stardis_amd.linelist.LineList: one upload of per-line scalars,
    parameters generated on the device (sdx_line_opacity_linelist_dev)."""
    from stardis_amd import linelist as LL

    ctx = default_context()
    dev = spec.upload(ctx)
    d_nus = ctx.upload(nus)
    out = ctx.empty((dev.n_depth, nus.size))
    ctx.call("sdx_line_opacity_linelist_dev", dev.n_depth, nus.size, d_nus.ptr, 0, nus.size, dev.byref(), out.ptr, nus.size, 0, None)
    gammas = doppler = None
    if RETURN_BROADENING_TABLES:
        _, gammas, doppler = LL.line_params(dev, ctx, alphas=False)
    return _download(out), gammas, doppler


def calc_alpha_line_at_nu(stellar_plasma, stellar_model, tracing_nus, line_opacity_config):
    """-> (alpha_line_at_nu (N_d, N_nu), gammas (N_l, N_d), doppler_widths (N_l, N_d)); (0, 0, 0) if disabled.
    Reference :328-441."""
    import pandas as pd

    if line_opacity_config.disable:
        return 0, 0, 0
    nus = _nus_of(tracing_nus)
    vald = line_opacity_config.vald_linelist
    if vald.use_linelist and getattr(stellar_plasma, "alpha_line_from_linelist", None) is None:
        # no dense alpha table on the plasma: generate alpha, gamma and the Doppler width in the kernel's pre-pass (f1)
        from stardis_amd.plasma.base import deferred_line_list

        spec = deferred_line_list(stellar_plasma.lines_from_linelist, nus, stellar_model, stellar_plasma,
                                  line_opacity_config.broadening, vald.use_vald_broadening)
        return _line_opacity_from_list(spec, nus)
    if vald.use_linelist:
        lines, alpha_table = stellar_plasma.lines_from_linelist, stellar_plasma.alpha_line_from_linelist
    else:
        lines, alpha_table = _atomic_line_table(stellar_plasma), stellar_plasma.alpha_line
    selected = _in_grid(lines, nus)
    line_nus = selected.nu.to_numpy()
    alphas_array = _in_grid(alpha_table, nus).drop(labels="nu", axis=1).to_numpy()
    selected = selected.apply(pd.to_numeric)
    if not vald.use_vald_broadening:  # auto-ionising lines are dropped unless VALD broadening is used (:413-421)
        keep = ~(selected.level_energy_upper > selected.ionization_energy).values
        selected, alphas_array, line_nus = selected[keep].copy(), alphas_array[keep].copy(), line_nus[keep].copy()
    gammas, doppler_widths = calculate_broadening(
        selected, stellar_model, stellar_plasma, line_opacity_config.broadening,
        use_vald_broadening=vald.use_vald_broadening and vald.use_linelist,
    )
    logger.info("Calculating line opacities at spectral points.")
    alpha = calc_alan_entries(_n_depth(stellar_model), nus, line_nus, doppler_widths, gammas, alphas_array)
    return alpha, gammas, doppler_widths


def calc_molecular_alpha_line_at_nu(stellar_plasma, stellar_model, tracing_nus, line_opacity_config):
    """Molecular counterpart of calc_alpha_line_at_nu.  Reference :444-484."""
    if line_opacity_config.disable:
        return 0, 0, 0
    nus = _nus_of(tracing_nus)
    if getattr(stellar_plasma, "molecule_alpha_line_from_linelist", None) is None:
        from stardis_amd.plasma.molecules import deferred_molecule_line_list

        spec = deferred_molecule_line_list(stellar_plasma.molecule_lines_from_linelist, nus, stellar_model, stellar_plasma,
                                           line_opacity_config.broadening)
        return _line_opacity_from_list(spec, nus)
    selected = _in_grid(stellar_plasma.molecule_lines_from_linelist, nus)
    alphas_array = _in_grid(stellar_plasma.molecule_alpha_line_from_linelist, nus).drop(labels="nu", axis=1).to_numpy()
    gammas, doppler_widths = calculate_molecule_broadening(selected, stellar_model, stellar_plasma, line_opacity_config.broadening)
    alpha = calc_alan_entries(_n_depth(stellar_model), nus, selected.nu.to_numpy(), doppler_widths, gammas, alphas_array)
    return alpha, gammas, doppler_widths


# ------------------------------------------------------------------------------------------------ driver
def calc_alphas(stellar_plasma, stellar_model, stellar_radiation_field, opacity_config):
    """Fill stellar_radiation_field.opacities.opacities_dict (same keys, same insertion order as the reference
    :655-736) and return the total (N_d, N_nu)."""
    opac = stellar_radiation_field.opacities
    entries = opac.opacities_dict
    nus = stellar_radiation_field.frequencies

    def put(key, value):
        entries[key] = value
        twin = device_twin(value)
        if twin is not None and hasattr(opac, "_remember"):
            opac._remember(key, value, twin)

    for source, fpath in opacity_config.file.items():
        put(f"alpha_file_{source}", calc_alpha_file(stellar_plasma, stellar_model, nus, source, fpath))
    put("alpha_bf", calc_alpha_bf(stellar_plasma, stellar_model, nus, opacity_config.bf))
    put("alpha_ff", calc_alpha_ff(stellar_plasma, stellar_model, nus, opacity_config.ff))
    put("alpha_rayleigh", calc_alpha_rayleigh(stellar_plasma, stellar_model, nus, opacity_config.rayleigh))
    put("alpha_electron", calc_alpha_electron(stellar_plasma, stellar_model, nus, opacity_config.disable_electron_scattering))
    alpha, gammas, doppler = calc_alpha_line_at_nu(stellar_plasma, stellar_model, nus, opacity_config.line)
    put("alpha_line_at_nu", alpha)
    put("alpha_line_at_nu_gammas", gammas)
    put("alpha_line_at_nu_doppler_widths", doppler)
    if opacity_config.line.include_molecules:
        alpha, gammas, doppler = calc_molecular_alpha_line_at_nu(stellar_plasma, stellar_model, nus, opacity_config.line)
        put("molecule_alpha_line_at_nu", alpha)
        put("molecule_alpha_line_at_nu_gammas", gammas)
        put("molecule_alpha_line_at_nu_doppler_widths", doppler)
    return opac.calc_total_alphas()
