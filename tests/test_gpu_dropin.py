"""Drop-in surface: stardis_amd.radiation_field.{calc_alphas, raytrace, RadiationField, Opacities} driven with a
pandas stand-in for the TARDIS plasma, against what the reference's own calc_alphas + raytrace returned for the
same objects (tests/golden/g9_end_to_end_*.npz).  Pins the host-side logic: line selection and sort, the
auto-ionisation filter, VALD / non-VALD broadening, dictionary keys and order, scalar-0 entries, totals, F_nu."""
import json
import os
import types

import numpy as np
import pandas as pd
import pytest

from conftest import ROOT, load_golden, rel_err

pytestmark = pytest.mark.gpu
NS = types.SimpleNamespace


def rebuild(tag, tmp_path):
    g = load_golden("g9_end_to_end_" + tag)
    nd = g["temperatures"].size
    cols = np.arange(nd)
    lines = pd.DataFrame({k[len("lines_"):]: g[k] for k in g.files if k.startswith("lines_")})
    alpha_line = pd.DataFrame(g["alpha_line_table"], columns=cols)
    alpha_line["nu"] = g["alpha_line_nu"]
    masses = pd.Series(g["line_mass_by_z_vals"], index=pd.Index(g["line_mass_by_z_keys"], name="atomic_number"))
    ind = pd.MultiIndex.from_tuples([(1, 0), (1, 1), (2, 0), (2, 1)], names=["atomic_number", "ion_number"])
    n_lev = g["level_density"].shape[0]
    lev_index = pd.MultiIndex.from_tuples([(1, 0, k) for k in range(n_lev)] + [(2, 0, 0)], names=["atomic_number", "ion_number", "level_number"])
    ionization_data = pd.Series(
        np.array([13.598434, 24.587, 54.418]) * 1.602176634e-12,
        index=pd.MultiIndex.from_tuples([(1, 1), (2, 1), (2, 2)], names=["atomic_number", "ion_number"]), name="ionization_energy",
    )
    plasma = NS(
        ion_number_density=pd.DataFrame(np.vstack([g["n_h1"], g["n_h2"], g["n_he1"], 1e-6 * g["n_he1"]]), index=ind, columns=cols),
        electron_densities=pd.Series(g["n_e"], index=cols),
        levels=lev_index,
        excitation_energy=pd.Series(np.append(g["level_excitation"], 0.0), index=lev_index),
        level_number_density=pd.DataFrame(np.vstack([g["level_density"], g["n_he1"][None, :]]), index=lev_index, columns=cols),
        ionization_data=ionization_data,
        h_minus_density=pd.Series(g["n_hminus"], index=cols),
        h2_density=pd.Series(g["h2_density"], index=cols),
    )
    vald = tag != "tardis"
    if vald:
        plasma.lines_from_linelist = lines
        plasma.alpha_line_from_linelist = alpha_line
    else:
        base_cols = ["atomic_number", "ion_number", "level_number_lower", "level_number_upper", "nu", "A_ul"]
        plasma.lines = lines[base_cols].copy()
        plasma.lines.index.name = "line_id"
        keys = lines[["atomic_number", "ion_number", "ionization_energy"]].drop_duplicates(["atomic_number", "ion_number"])
        extra = pd.Series(
            keys.ionization_energy.values,
            index=pd.MultiIndex.from_arrays([keys.atomic_number.values, keys.ion_number.values + 1], names=["atomic_number", "ion_number"]),
            name="ionization_energy",
        )
        plasma.ionization_data = pd.concat([ionization_data[~ionization_data.index.isin(extra.index)], extra]).sort_index()
        lev_idx = pd.MultiIndex.from_arrays(
            [np.concatenate([lines.atomic_number.values] * 2), np.concatenate([lines.ion_number.values] * 2),
             np.concatenate([lines.level_number_lower.values, lines.level_number_upper.values])],
            names=["atomic_number", "ion_number", "level_number"],
        )
        plasma.atomic_data = NS(levels=NS(energy=pd.Series(
            np.concatenate([lines.level_energy_lower.values, lines.level_energy_upper.values]), index=lev_idx, name="energy")))
        plasma.alpha_line = alpha_line
    r = np.concatenate([[0.0], np.cumsum(g["dist"])])
    model = NS(
        temperatures=g["temperatures"], no_of_depth_points=nd, spherical=False,
        geometry=NS(dist_to_next_depth_point=g["dist"], r=r, reference_r=None),
        composition=NS(nuclide_masses=masses), microturbulence=1.0e5,
    )
    with open(os.path.join(ROOT, "stardis_amd", "data", "hminus_bf_wishart1979.json")) as fh:
        tab = json.load(fh)
    table_path = tmp_path / "h_minus_bf.dat"
    table_path.write_text("\n".join(f"{x!r},{y!r}" for x, y in zip(tab["wavelength"], tab["cross_section"])) + "\n")
    cfg = NS(
        file={"Hminus_bf": str(table_path)}, bf={"H_I": {}}, ff={"H_I": {}}, rayleigh=["H", "He", "H2"],
        disable_electron_scattering=False,
        line=NS(disable=False, broadening=["linear_stark", "quadratic_stark", "van_der_waals", "radiation"],
                vald_linelist=NS(use_linelist=vald, use_vald_broadening=(tag == "vald")), include_molecules=False),
    )
    return g, plasma, model, cfg


@pytest.mark.parametrize("tag", ["tardis", "vald", "vald_nb"])
def test_calc_alphas_and_raytrace_match_reference(ctx, tag, tmp_path):
    from stardis_amd.radiation_field import RadiationField
    from stardis_amd.radiation_field.opacities.opacities_solvers import calc_alphas
    from stardis_amd.radiation_field.radiation_field_solvers import raytrace
    from stardis_amd.radiation_field.source_functions.blackbody import blackbody_flux_at_nu

    g, plasma, model, cfg = rebuild(tag, tmp_path)
    nus = g["nus"].copy()
    field = RadiationField(nus, blackbody_flux_at_nu, model, 6)
    assert np.array_equal(field.thetas, g["thetas"]) and np.array_equal(field.I_nus_weights, g["weights"])
    total = calc_alphas(plasma, model, field, cfg)
    assert list(field.opacities.opacities_dict.keys()) == [str(k) for k in g["dict_keys"]]
    for key, value in field.opacities.opacities_dict.items():
        ref = g["od_" + key]
        tol = 1e-12 if "alpha_line" in key else 1e-13
        assert np.shape(value) == ref.shape, key
        assert rel_err(np.asarray(value, dtype=float), ref) < tol, key
    assert total is field.opacities.total_alphas
    assert rel_err(total, g["total_alphas"]) < 1e-12
    F = raytrace(model, field)
    assert F is field.F_nu
    assert rel_err(F, g["F_nu"]) < 1e-10


def test_disabled_sources_return_scalar_zero(ctx, tmp_path):
    from stardis_amd.radiation_field import RadiationField
    from stardis_amd.radiation_field.opacities.opacities_solvers import base as B
    from stardis_amd.radiation_field.source_functions.blackbody import blackbody_flux_at_nu

    g, plasma, model, cfg = rebuild("vald", tmp_path)
    cfg.line.disable = True
    cfg.disable_electron_scattering = True
    cfg.rayleigh = []
    field = RadiationField(g["nus"].copy(), blackbody_flux_at_nu, model, 4, track_individual_intensities=True)
    total = B.calc_alphas(plasma, model, field, cfg)
    od = field.opacities.opacities_dict
    assert od["alpha_electron"] == 0 and od["alpha_line_at_nu"] == 0 and od["alpha_line_at_nu_gammas"] == 0  # base.py:164-165, :359-360
    assert not np.asarray(od["alpha_rayleigh"]).any()
    want = g["od_alpha_file_Hminus_bf"] + g["od_alpha_bf"] + g["od_alpha_ff"]
    assert rel_err(total, want) < 1e-13
    from stardis_amd.radiation_field.radiation_field_solvers import raytrace

    raytrace(model, field)
    assert field.I_nus.shape == (56, g["nus"].size, 4)
    assert rel_err(field.F_nu, np.tensordot(field.I_nus, field.I_nus_weights, axes=([2], [0]))) < 1e-14


def test_species_keys_with_the_stage_in_digits(ctx, tmp_path):
    """`H_1` is neutral hydrogen, like `H_I`: get_number_density hands "H 1" to tardis.util.base.species_string_to_tuple
    (util.py:154-156), for which digits are the spectroscopic stage.  The reference run with that key (G5's *_digit_key
    arrays, equal to its `H_I` arrays) pins the meaning; here both spellings must give the reference's G9 planes, through
    the source-by-source functions and through the fused call."""
    from stardis_amd.radiation_field import base as RB
    from stardis_amd.radiation_field.opacities.opacities_solvers import base as B

    g5 = load_golden("g5_continuum")
    for tag in ("opt", "wide"):
        assert np.array_equal(g5[tag + "_alpha_bf_digit_key"], g5[tag + "_alpha_bf"]) and np.array_equal(g5[tag + "_alpha_ff_digit_key"], g5[tag + "_alpha_ff"])
    g, plasma, model, cfg = rebuild("vald", tmp_path)
    nus = g["nus"].copy()
    assert rel_err(B.calc_alpha_bf(plasma, model, nus, {"H_1": {}}), g["od_alpha_bf"]) < 1e-13
    assert rel_err(B.calc_alpha_ff(plasma, model, nus, {"H_1": {}}), g["od_alpha_ff"]) < 1e-13
    cfg.bf, cfg.ff = {"H_1": {}}, {"H_1": {}}
    config = NS(opacity=cfg, no_of_thetas=6, result_options=NS(return_radiation_field=False))
    field = RB.create_stellar_radiation_field(nus, model, plasma, config)
    assert rel_err(field.F_nu, g["F_nu"]) < 1e-10
    assert rel_err(field.opacities.opacities_dict["alpha_bf"], g["od_alpha_bf"]) < 1e-13
