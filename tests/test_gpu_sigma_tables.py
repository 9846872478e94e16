"""f2 on the GPU: sdx_sigma_table_2d_dev against the reference's sigma_file output (G12), and calc_alpha_file of the mirror
end to end — table files rewritten from the G12 arrays in the reference's file formats — against the reference's own
calc_alpha_file outputs in G5."""
import types

import numpy as np
import pandas as pd
import pytest

from conftest import load_golden
from stardis_amd import ops
from stardis_amd.radiation_field.opacities.opacities_solvers import base as solvers
from stardis_amd.radiation_field.opacities.opacities_solvers import util as U

pytestmark = pytest.mark.gpu
NS = types.SimpleNamespace
SOURCES = ("H2plus_bf", "Hminus_ff")


def max_rel(a, b):
    m = b != 0
    return float(np.max(np.abs(a - b)[m] / np.abs(b[m])))


@pytest.mark.parametrize("src", SOURCES)
def test_sigma_table_matches_reference(src):
    g = load_golden("g12_sigma_tables")
    wave, axis2, values = g[src + "_wave"], g[src + "_axis2"], g[src + "_values"]
    cells, vertex_values = U.cell_lookup(wave, axis2, g[src + "_simplices"], values)
    temps, lam = g[src + "_query_temperatures"], g[src + "_lambdas"]
    second, kind = (temps, 1) if src == "H2plus_bf" else (5040 / temps, 2)
    dev, zero_rows = ops.sigma_table_2d(wave, axis2, cells, g[src + "_transform"], vertex_values, lam, second, kind, temps)
    sig, ref = dev.numpy(), g[src + "_sigma"]
    assert np.array_equal(sig == 0, ref == 0)
    assert (sig == ref).mean() > 0.99
    assert max_rel(sig, ref) < 1e-15
    assert np.array_equal(zero_rows, np.unique(np.where(ref == 0)[0]))  # the rows the reference's warning names


def _mantissa_exponent(v):
    """The H2+ table's own notation: 7.34-5 for 7.34e-05 (the reader inserts the 'e', util.py:40)."""
    r = repr(float(v))
    return r.replace("e-", "-") if "e-" in r else r


def write_tables(tmp_path):
    g = load_golden("g12_sigma_tables")
    paths = {}
    wave, temps, values = g["H2plus_bf_wave"], g["H2plus_bf_axis2"], g["H2plus_bf_values"]
    nm = np.round(wave / 10.0)
    rows = ["(nxn) " + " ".join(str(int(t)) for t in temps)]
    rows += [f"{int(n)} " + " ".join(_mantissa_exponent(v) for v in row) for n, row in zip(nm, values)]
    paths["H2plus_bf"] = tmp_path / "h2_plus_bf.dat"
    paths["H2plus_bf"].write_text("# rewritten from tests/golden/g12_sigma_tables.npz\n" + "\n".join(rows) + "\n")
    wave, thetas, values = g["Hminus_ff_wave"], g["Hminus_ff_axis2"], g["Hminus_ff_values"]
    rows = ["wave, " + ", ".join(repr(float(t)) for t in thetas)]
    rows += [repr(float(w)) + " " + " ".join(repr(float(v)) for v in row) for w, row in zip(wave, values)]
    paths["Hminus_ff"] = tmp_path / "h_minus_ff.dat"
    paths["Hminus_ff"].write_text("# rewritten from tests/golden/g12_sigma_tables.npz\n" + "\n".join(rows) + "\n")
    return g, paths


def test_rewritten_tables_read_back_exactly(tmp_path):
    g, paths = write_tables(tmp_path)
    for src in SOURCES:
        _, wave, axis2, values = U.read_table(paths[src], src)
        assert np.array_equal(wave, g[src + "_wave"]) and np.array_equal(np.asarray(axis2, float), g[src + "_axis2"])
        assert np.array_equal(values, g[src + "_values"])


@pytest.mark.parametrize("src", SOURCES)
def test_calc_alpha_file_matches_reference(tmp_path, src):
    """opacities_solvers/base.py:40-70 through the mirror: table file -> triangulation (host) -> cross-sections and
    sigma x density (GPU) against the reference's calc_alpha_file."""
    _, paths = write_tables(tmp_path)
    g5 = load_golden("g5_continuum")
    nd = g5["temperatures"].size
    cols = np.arange(nd)
    ind = pd.MultiIndex.from_tuples([(1, 0), (1, 1), (2, 0)], names=["atomic_number", "ion_number"])
    plasma = NS(
        ion_number_density=pd.DataFrame(np.vstack([g5["n_h1"], g5["n_h2"], g5["n_he1"]]), index=ind, columns=cols),
        electron_densities=pd.Series(g5["n_e"], index=cols),
        h2_plus_density=pd.Series(g5["h2_plus_density"], index=cols),
    )
    model = NS(temperatures=g5["temperatures"], no_of_depth_points=nd)
    for tag in ("opt", "wide"):
        out = solvers.calc_alpha_file(plasma, model, g5[tag + "_nus"], src, str(paths[src]))
        ref = g5[f"{tag}_alpha_file_{src}"]
        assert np.array_equal(np.asarray(out) == 0, ref == 0)
        assert not (ref != 0).any() or max_rel(np.asarray(out), ref) < 1e-15
        sig = U.sigma_file(g5[tag + "_lambdas"], g5["temperatures"], paths[src], src)
        assert not (ref != 0).any() or max_rel(sig, g5[f"{tag}_sigma_{src}"]) < 1e-15


def test_bad_tables_are_rejected(ctx):
    with pytest.raises(ValueError, match="axes"):
        ops.sigma_table_2d(np.arange(1.0), np.arange(3.0), np.zeros((1, 2), np.int32), np.zeros((1, 3, 2)), np.zeros((1, 3)), [1.0], [1.0])


def test_cool_layers_fall_off_the_h2plus_table():
    """Temperatures below the first node of the H2+ table (3150 K): sigma 0 and the rows named, as the reference's
    fill_value=0 + warning do (util.py:54-62) — the cool-dwarf structure has 8 such layers."""
    from stardis_amd import synth

    g = load_golden("g12_sigma_tables")
    atm = synth.cool_dwarf_atmosphere()
    t = atm["temperatures"]
    wave, axis2, values = g["H2plus_bf_wave"], g["H2plus_bf_axis2"], g["H2plus_bf_values"]
    cells, vertex_values = U.cell_lookup(wave, axis2, g["H2plus_bf_simplices"], values)
    lam = np.linspace(wave.min(), wave.max(), 300)
    dev, zero_rows = ops.sigma_table_2d(wave, axis2, cells, g["H2plus_bf_transform"], vertex_values, lam, t, 1, t)
    sig = dev.numpy()
    cold = t < axis2.min()
    assert cold.sum() >= 5 and not sig[cold].any() and (sig[~cold] > 0).all()
    assert np.array_equal(zero_rows, np.flatnonzero(cold))
    import oracle

    ref = oracle.interp_triangulated(wave, axis2, cells, g["H2plus_bf_transform"], vertex_values, lam, t) * 1e-18
    assert np.array_equal(sig == 0, ref == 0) and max_rel(sig, ref) < 1e-15
