"""Rebuild the pandas objects of tests/golden/g11_linelist.npz (what make_golden.g11_linelist handed to the
reference's AlphaLine* classes and to calc_alpha_line_at_nu)."""
import types

import numpy as np
import pandas as pd

from conftest import load_golden

NS = types.SimpleNamespace


def rebuild():
    g = load_golden("g11_linelist")
    t = g["temperatures"]
    cols = np.arange(t.size)
    atoms = pd.DataFrame({k[len("atoms_"):]: g[k] for k in g.files if k.startswith("atoms_")})
    mols = pd.DataFrame({k[len("mols_"):]: g[k] for k in g.files if k.startswith("mols_")})
    idx = pd.MultiIndex.from_arrays(g["species"].T, names=["atomic_number", "ion_number"])
    ion_density = pd.DataFrame(g["ion_number_density"], index=idx, columns=cols)
    partition = pd.DataFrame(g["partition_function"], index=idx, columns=cols)
    ionization_data = pd.Series(
        g["ionization_energy"], index=pd.MultiIndex.from_arrays(g["ionization_keys"].T, names=["atomic_number", "ion_number"]),
        name="ionization_energy",
    )
    names = [str(m) for m in g["molecule_names"]]
    mol_density = pd.DataFrame(g["molecule_number_density"], index=names, columns=cols)
    mol_partition = pd.DataFrame(g["molecule_partition_function"], index=names, columns=cols)
    atomic_data = NS(linelist_atoms=atoms, linelist_molecules=mols, selected_atomic_numbers=pd.Index(g["selected_atomic_numbers"]))
    masses = pd.Series(g["mass_vals"], index=pd.Index(g["mass_keys"], name="atomic_number"))
    model = NS(
        temperatures=t, no_of_depth_points=t.size, spherical=False, composition=NS(nuclide_masses=masses),
        microturbulence=float(g["microturbulence"]),
    )
    plasma = NS(
        ion_number_density=ion_density, partition_function=partition, electron_densities=pd.Series(g["n_e"], index=cols),
        ionization_data=ionization_data, molecule_number_density=mol_density, molecule_partition_function=mol_partition,
        molecule_ion_map=pd.DataFrame(dict(Ion1=g["molecule_ion1"], Ion2=g["molecule_ion2"]), index=names),
    )
    return NS(g=g, atomic_data=atomic_data, ion_density=ion_density, partition=partition, ionization_data=ionization_data,
              mol_density=mol_density, mol_partition=mol_partition, model=model, plasma=plasma, t=t)


def line_config(vald_broadening, broadening=("linear_stark", "quadratic_stark", "van_der_waals", "radiation")):
    return NS(disable=False, broadening=list(broadening), vald_linelist=NS(use_linelist=True, use_vald_broadening=vald_broadening),
              include_molecules=False)
