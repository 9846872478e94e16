"""Bound-free / free-free with several species and Z > 1 against what the reference returned (G14,
tests/golden/make_golden.py::g14_continuum_species): `(ion_number+1)**4`, `((ion_number+1)*sqrt(nu_R/nu_c))**5`
(opacities_solvers/base.py:262-263), `ion_number**2` (:312), the species loop's summation order (:204, :304) and the level
filter (:215-219) on a plasma whose level index interleaves H I, He I and He II.  Three ways in: the C-ABI operators, the
drop-in calc_alpha_bf / calc_alpha_ff on a pandas stand-in plasma, and the fused create_stellar_radiation_field."""
import types

import numpy as np
import pandas as pd
import pytest

import oracle
from conftest import load_golden, rel_err, species_arrays_from_g14
from stardis_amd import ops

pytestmark = pytest.mark.gpu
NS = types.SimpleNamespace
TOL = 1e-14


def cases(g):
    return [(str(c), str(s).split(",")) for c, s in zip(g["case_names"], g["case_species"])]


def test_operators_match_reference(ctx):
    g = load_golden("g14_continuum_species")
    nd = g["temperatures"].size
    for case, species in cases(g):
        off, bf_ion, cut, ld, ff_ion, ff_n = species_arrays_from_g14(g, species)
        for tag in ("wide", "uv"):
            nus = g[tag + "_nus"]
            bf = ops.alpha_bf(nus, off, bf_ion, cut, ld, nd).numpy()
            ff = ops.alpha_ff(nus, g["temperatures"], ff_ion, ff_n).numpy()
            assert rel_err(bf, g[f"{tag}_alpha_bf_{case}"]) < TOL, (case, tag)
            assert rel_err(ff, g[f"{tag}_alpha_ff_{case}"]) < TOL, (case, tag)
            # zeros below every edge stay exact zeros (`0 * number_density`, :266)
            assert np.array_equal(bf == 0, g[f"{tag}_alpha_bf_{case}"] == 0)
            assert rel_err(bf, oracle.alpha_bf(nus, off, bf_ion, cut, ld)) < TOL


def plasma_of(g):
    nd = g["temperatures"].size
    cols = np.arange(nd)
    ind = pd.MultiIndex.from_tuples([tuple(k) for k in g["ion_index"]], names=["atomic_number", "ion_number"])
    lev = pd.MultiIndex.from_tuples([tuple(k) for k in g["level_index"]], names=["atomic_number", "ion_number", "level_number"])
    chi = pd.Series(g["ionization_energy"], name="ionization_energy",
                    index=pd.MultiIndex.from_tuples([tuple(k) for k in g["ionization_index"]], names=["atomic_number", "ion_number"]))
    plasma = NS(
        ion_number_density=pd.DataFrame(g["ion_number_density"], index=ind, columns=cols),
        electron_densities=pd.Series(g["n_e"], index=cols),
        levels=lev,
        excitation_energy=pd.Series(g["level_excitation"], index=lev),
        level_number_density=pd.DataFrame(g["level_density"], index=lev, columns=cols),
        ionization_data=chi,
    )
    dist = np.full(nd - 1, 1.0e6)
    model = NS(temperatures=g["temperatures"], no_of_depth_points=nd, spherical=False,
               geometry=NS(dist_to_next_depth_point=dist, r=np.concatenate([[0.0], np.cumsum(dist)]), reference_r=None),
               composition=NS(nuclide_masses=None), microturbulence=1.0e5)
    return plasma, model


def test_dropin_functions_match_reference(ctx):
    from stardis_amd.radiation_field.opacities.opacities_solvers import base as B

    g = load_golden("g14_continuum_species")
    plasma, model = plasma_of(g)
    for case, species in cases(g):
        for tag in ("wide", "uv"):
            nus = g[tag + "_nus"].copy()
            assert rel_err(B.calc_alpha_bf(plasma, model, nus, {k: {} for k in species}), g[f"{tag}_alpha_bf_{case}"]) < TOL, (case, tag)
            assert rel_err(B.calc_alpha_ff(plasma, model, nus, {k: {} for k in species}), g[f"{tag}_alpha_ff_{case}"]) < TOL, (case, tag)


@pytest.mark.parametrize("case", ["he2_h_he1", "h_he2"])
def test_fused_call_matches_reference(ctx, case):
    """create_stellar_radiation_field (one fused device pass) with the species lists of G14: the dictionary's alpha_bf and
    alpha_ff entries are the reference's arrays, the total is their sum plus Thomson, and the general path gives the same bits."""
    from stardis_amd.radiation_field import base as RB

    g = load_golden("g14_continuum_species")
    species = dict(cases(g))[case]
    plasma, model = plasma_of(g)
    cfg = NS(file={}, bf={k: {} for k in species}, ff={k: {} for k in species}, rayleigh=[], disable_electron_scattering=False,
             line=NS(disable=True, broadening=[], vald_linelist=NS(use_linelist=False, use_vald_broadening=False), include_molecules=False))
    config = NS(opacity=cfg, no_of_thetas=4, result_options=NS(return_radiation_field=False))
    for tag in ("wide", "uv"):
        nus = g[tag + "_nus"].copy()
        first = nus.copy()
        field = RB.create_stellar_radiation_field(nus, model, plasma, config)
        od = field.opacities.opacities_dict
        # frequencies above 2.3e15 Hz are zeroed in the caller's array by calc_alpha_rayleigh (:99) even with no Rayleigh species:
        # such a grid takes the general path (the fused pass does not reproduce the mutation), a grid below the cut-off the fused one
        clipped = g[tag + "_nus"] > 2.3e15
        assert np.array_equal(nus == 0, clipped)
        assert type(field.opacities).__name__ == ("Opacities" if clipped.any() else "FusedOpacities")
        assert rel_err(od["alpha_bf"], g[f"{tag}_alpha_bf_{case}"]) < TOL
        assert rel_err(od["alpha_ff"], g[f"{tag}_alpha_ff_{case}"]) < TOL
        want = g[f"{tag}_alpha_bf_{case}"] + g[f"{tag}_alpha_ff_{case}"] + np.asarray(od["alpha_electron"])
        assert rel_err(field.opacities.total_alphas, want) < 1e-13
        saved, RB.FUSED = RB.FUSED, False
        try:
            general = RB.create_stellar_radiation_field(first, model, plasma, config)
        finally:
            RB.FUSED = saved
        assert np.array_equal(np.asarray(general.opacities.opacities_dict["alpha_bf"]), np.asarray(od["alpha_bf"]))
        assert np.array_equal(np.asarray(general.opacities.opacities_dict["alpha_ff"]), np.asarray(od["alpha_ff"]))
        assert np.array_equal(general.F_nu, field.F_nu, equal_nan=True)
