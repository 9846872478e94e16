"""f1 (line parameters from a line list) without a GPU: the oracle's alpha_line restatement against what the
reference's AlphaLine* classes returned (tests/golden/g11_linelist.npz), and the O(N_l) host bookkeeping of
stardis_amd.plasma against the reference's second output (the per-line table)."""
import numpy as np
import pytest

import oracle
from conftest import rel_err
from linelist_fixture import rebuild
from stardis_amd.plasma import base as pb
from stardis_amd.plasma import molecules as pm


@pytest.fixture(scope="module")
def fx():
    return rebuild()


def _atom_rows(fx, ll):
    pop, index = pb.population_table(pb._species_keys(ll), fx.ion_density, fx.partition)
    return pop, np.array([index[k] for k in zip(ll.atomic_number.values, ll.ion_number.values)], dtype=np.int32)


@pytest.mark.parametrize("short", [False, True])
def test_oracle_alpha_line_atoms(fx, short):
    g, tag = fx.g, ("short" if short else "full")
    ll = pb._atom_lines(fx.atomic_data, fx.ionization_data, short)
    pop, row = _atom_rows(fx, ll)
    a = oracle.alpha_line_linelist(ll.e_low.values, None if short else ll.g_lo.values, 10 ** ll.log_gf.values if short else ll.f_lu.values,
                                   ll.nu.values, row, pop, fx.t, float(g["alpha_coefficient"]))
    a = a[g[tag + "_index"]]  # the reference returns alphas[valid_indices] (plasma/base.py:320-321)
    # libm exp against numpy's exp: <= 1 ulp each on two factors
    assert rel_err(a, g[tag + "_alphas"]) < 2e-15


@pytest.mark.parametrize("short", [False, True])
def test_oracle_alpha_line_molecules(fx, short):
    g, tag = fx.g, ("molshort" if short else "molfull")
    ll = pm._molecule_lines(fx.atomic_data, short)
    pop, row, strength, g_lo = pm._alpha_inputs(ll, fx.mol_density, fx.mol_partition, short)
    a = oracle.alpha_line_linelist(ll.e_low.values, g_lo, strength, ll.nu.values, row, pop, fx.t, float(g["alpha_coefficient"]))
    assert rel_err(a, g[tag + "_alphas"]) < 2e-15


@pytest.mark.parametrize("short", [False, True])
def test_host_line_table_atoms(fx, short):
    """nu (astropy's AA -> Hz), level energies (eV -> erg), A_ul, the ionisation-energy merge, the truncation to the
    selected elements and — full lists — the auto-ionisation filter."""
    g, tag = fx.g, ("short" if short else "full")
    ll = pb._atom_lines(fx.atomic_data, fx.ionization_data, short)
    assert ll.atomic_number.max() <= 26 and len(ll) < len(fx.atomic_data.linelist_atoms)
    if not short:
        ll = ll[ll.level_energy_upper < ll.ionization_energy]
    assert np.array_equal(np.asarray(ll.index), g[tag + "_lines_index"])
    assert np.array_equal(ll.nu.values, g[tag + "_lines_nu"])  # bit-exact: the window centres depend on it
    assert np.array_equal(ll.level_energy_lower.values, g[tag + "_lines_level_energy_lower"])
    assert np.array_equal(ll.ionization_energy.values, g[tag + "_lines_ionization_energy"])
    assert rel_err(ll.level_energy_upper.values, g[tag + "_lines_level_energy_upper"]) < 4e-16
    assert rel_err(ll.e_up.values, g[tag + "_lines_e_up"]) < 4e-16
    assert rel_err(ll.A_ul.values, g[tag + "_lines_A_ul"]) < 4e-16  # numpy's pow differs by an ulp between versions


@pytest.mark.parametrize("short", [False, True])
def test_host_line_table_molecules(fx, short):
    g, tag = fx.g, ("molshort" if short else "molfull")
    ll = pm._molecule_lines(fx.atomic_data, short)
    assert np.array_equal(ll.nu.values, g[tag + "_lines_nu"])
    assert np.array_equal(ll.level_energy_lower.values, g[tag + "_lines_level_energy_lower"])
    assert rel_err(ll.level_energy_upper.values, g[tag + "_lines_level_energy_upper"]) < 4e-16
    assert rel_err(ll.A_ul.values, g[tag + "_lines_A_ul"]) < 4e-16


def test_oracle_chain_reproduces_reference_line_opacity(fx):
    """alpha (oracle) -> gamma, Doppler width (oracle) -> calc_alan_entries (oracle) against the reference's own
    calc_alpha_line_at_nu on its own dense tables: the whole f1 chain restated on the CPU."""
    g = fx.g
    nus = g["nus"]
    n_h = fx.plasma.ion_number_density.loc[1, 0].values  # broadening.py:716
    for tag, vb in (("full_vb", True), ("full_nb", False)):
        ll = pb._atom_lines(fx.atomic_data, fx.ionization_data, False)
        ll = ll[ll.level_energy_upper < ll.ionization_energy]
        sel = ll.sort_values("nu")
        sel = sel[sel.nu.between(nus.min(), nus.max())]
        pop, row = _atom_rows(fx, sel)
        alphas = oracle.alpha_line_linelist(sel.e_low.values, sel.g_lo.values, sel.f_lu.values, sel.nu.values, row, pop, fx.t,
                                            float(g["alpha_coefficient"]))
        mass = fx.model.composition.nuclide_masses.loc[sel.atomic_number].values
        args = (sel.atomic_number.values, sel.ion_number.values + 1, sel.ionization_energy.values, sel.level_energy_upper.values,
                sel.level_energy_lower.values, sel.A_ul.values)
        if vb:
            gam = oracle.calc_vald_gamma(*args, sel.stark.values, sel.waals.values, mass, g["n_e"], fx.t, n_h)
        else:
            gam = oracle.calc_gamma(*args, g["n_e"], fx.t, n_h)
        dop = oracle.doppler_widths(sel.nu.values, mass, fx.t, float(g["microturbulence"]))
        assert rel_err(gam, g[tag + "_gammas"]) < 1e-13
        assert rel_err(dop, g[tag + "_doppler"]) < 1e-15
        out = oracle.calc_alan_entries(fx.t.size, nus, sel.nu.values, dop, gam, alphas)
        assert rel_err(out, g[tag + "_alpha_line_at_nu"]) < 1e-12


def test_oracle_alpha_line_levels_bit_exact():
    """AlphaLine.calculate (plasma/base.py:146-175): products only, so bit-exact."""
    from conftest import load_golden

    g = load_golden("g13_alpha_line_levels")
    a = oracle.alpha_line_levels(g["level_number_density"], g["lines_lower_level_index"], g["stimulated_emission_factor"], g["f_lu"],
                                 0.026540088545744744)
    assert np.array_equal(a, g["alpha_line"])
