"""f2 (2-D cross-section tables) without a GPU: the oracle's restatement of scipy's LinearNDInterpolator against what the
reference's sigma_file returned (tests/golden/g12_sigma_tables.npz holds the tables, the triangulation the generating
interpreter's Qhull built, the queries and the reference's output), and the host-side cell index."""
import numpy as np
import pytest

import oracle
from conftest import load_golden
from stardis_amd import constants as K
from stardis_amd.radiation_field.opacities.opacities_solvers import util as U

SOURCES = ("H2plus_bf", "Hminus_ff")


def scipy_interp2d(wave, axis2, values, lambdas, second):
    """What the reference calls (util.py:47-56): the installed scipy, as an independent cross-check."""
    from scipy.interpolate import LinearNDInterpolator

    w_mesh, a_mesh = np.meshgrid(wave, axis2, indexing="ij")
    f = LinearNDInterpolator(np.vstack([w_mesh.ravel(), a_mesh.ravel()]).T, values.flatten(), fill_value=0)
    lam, sec = np.meshgrid(lambdas, second)
    return f(lam, sec)


def scaled(src, raw, temps):
    return raw * 1e-18 if src == "H2plus_bf" else raw * 1e-26 * K.K_B_CGS * temps[:, np.newaxis]  # util.py:58, :83-88


def max_rel(a, b):
    m = b != 0
    return float(np.max(np.abs(a - b)[m] / np.abs(b[m])))


@pytest.mark.parametrize("src", SOURCES)
def test_oracle_reproduces_reference_sigma_file(src):
    g = load_golden("g12_sigma_tables")
    wave, axis2, values = g[src + "_wave"], g[src + "_axis2"], g[src + "_values"]
    cells, vertex_values = U.cell_lookup(wave, axis2, g[src + "_simplices"], values)
    temps, lam = g[src + "_query_temperatures"], g[src + "_lambdas"]
    second = temps if src == "H2plus_bf" else 5040 / temps
    sig = scaled(src, oracle.interp_triangulated(wave, axis2, cells, g[src + "_transform"], vertex_values, lam, second), temps)
    ref = g[src + "_sigma"]
    assert np.array_equal(sig == 0, ref == 0)  # same points fall outside the table
    assert (sig == ref).mean() > 0.99  # the rest: points on a shared edge, where scipy may pick the neighbouring triangle
    assert max_rel(sig, ref) < 1e-15


@pytest.mark.parametrize("src", SOURCES)
def test_installed_scipy_builds_the_same_triangulation(src):
    """The diagonals are Qhull's choice; the interpreter that made the goldens (scipy 1.7.1) and the installed one agree on
    these tables, and the restatement agrees with the installed LinearNDInterpolator (a third-party cross-check)."""
    g = load_golden("g12_sigma_tables")
    wave, axis2, values = g[src + "_wave"], g[src + "_axis2"], g[src + "_values"]
    simplices, transform = U.triangulate(wave, axis2, values)
    canon = lambda s: np.sort(np.sort(s, axis=1), axis=0)  # noqa: E731
    assert np.array_equal(canon(simplices), canon(g[src + "_simplices"]))
    cells, vertex_values = U.cell_lookup(wave, axis2, simplices, values)
    assert cells.shape == ((wave.size - 1) * (axis2.size - 1), 2)
    temps, lam = g[src + "_query_temperatures"], g[src + "_lambdas"]
    second = temps if src == "H2plus_bf" else 5040 / temps
    raw = oracle.interp_triangulated(wave, axis2, cells, transform, vertex_values, lam, second)
    lib = scipy_interp2d(wave, axis2, values, lam, second)
    assert np.array_equal(raw == 0, lib == 0)
    assert max_rel(raw, lib) < 1e-15


def test_nm_to_angstrom_is_astropys():
    g = load_golden("g12_sigma_tables")
    nm = np.round(g["H2plus_bf_wave"] / 10.0)  # the file's index, nm
    assert np.array_equal(nm * K.NM_TO_ANGSTROM, g["H2plus_bf_wave"])  # (index * u.nm).to(u.AA), util.py:43
    assert not np.array_equal(nm * 10.0, g["H2plus_bf_wave"])


def test_cell_lookup_rejects_other_triangulations():
    wave, axis2 = np.arange(4.0), np.arange(3.0)
    fan = np.array([[0, 1, 5], [0, 5, 3]])  # a triangle spanning two cells
    assert U.cell_lookup(wave, axis2, fan, np.zeros(12)) is None


@pytest.mark.parametrize("src", SOURCES)
def test_oracle_reproduces_g5_cross_sections(src):
    """The continuum goldens (G5) hold sigma_file's output for the opacity tests' grids: same tables, other queries."""
    g, g5 = load_golden("g12_sigma_tables"), load_golden("g5_continuum")
    wave, axis2, values = g[src + "_wave"], g[src + "_axis2"], g[src + "_values"]
    cells, vertex_values = U.cell_lookup(wave, axis2, g[src + "_simplices"], values)
    temps = g5["temperatures"]
    for tag in ("opt", "wide"):
        lam = g5[tag + "_lambdas"]
        second = temps if src == "H2plus_bf" else 5040 / temps
        sig = scaled(src, oracle.interp_triangulated(wave, axis2, cells, g[src + "_transform"], vertex_values, lam, second), temps)
        ref = g5[f"{tag}_sigma_{src}"]
        assert np.array_equal(sig == 0, ref == 0)
        assert not (ref != 0).any() or max_rel(sig, ref) < 1e-15
