"""Cross-section tables and density look-ups feeding calc_alpha_file; host-side I/O that mirrors
stardis/radiation_field/opacities/opacities_solvers/util.py (sigma_file :14-108, get_number_density :111-166).

The tables are a few hundred numbers read once per run.  The 2-D tables (H- ff, H2+ bf) are triangulated once by
scipy/Qhull on the host — the same Delaunay call LinearNDInterpolator makes, because the cell diagonals are Qhull's
choice — and the N_d x N_nu evaluations run on the GPU (sdx_sigma_table_2d_dev restates scipy's barycentric
evaluation).  The 1-D H- bf table is interpolated on the GPU with np.interp semantics."""
import logging

import numpy as np

from stardis_amd import constants as K
from stardis_amd.util import species_string_to_tuple

logger = logging.getLogger(__name__)


_TABLE_CACHE = {}


def read_table(fpath, opacity_source):
    """-> ("1d", wavelength_AA, sigma) or ("2d", wavelength_AA, second_axis, values, scale_kind).  Parsed once per
    (file, modification time): the reference re-reads the file on every call, which is a millisecond of pandas per source
    next to a synthesis that takes a tenth of that."""
    import os

    try:
        stamp = (str(fpath), opacity_source, os.stat(fpath).st_mtime_ns)
    except OSError:
        stamp = None
    if stamp is not None and stamp in _TABLE_CACHE:
        return _TABLE_CACHE[stamp]
    table = _read_table(fpath, opacity_source)
    if stamp is not None:
        _TABLE_CACHE[stamp] = table
    return table


def _read_table(fpath, opacity_source):
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
        wave = tab.index.to_numpy(dtype=float) * K.NM_TO_ANGSTROM  # (index * u.nm).to(u.AA) (util.py:43)
        temps = tab.columns.to_numpy().astype(int)
        return "2d", wave, temps, tab.to_numpy()
    raise ValueError(f"Unknown opacity_source: {opacity_source}")


def cell_lookup(wave, axis2, simplices, values):
    """Index a Delaunay triangulation of the rectilinear table by grid cell: every triangle lies in one cell, two per
    cell (which diagonal is Qhull's choice).  -> (cell_simplices (n_cells, 2) int32, simplex_values (n_s, 3)), or None
    when the triangulation does not have that structure (the caller then lets scipy evaluate on the host)."""
    wave, axis2 = np.asarray(wave, dtype=float), np.asarray(axis2, dtype=float)
    nx, ny = wave.size, axis2.size
    simplices = np.asarray(simplices)
    pi, pj = np.divmod(simplices, ny)  # point k of the ravelled "ij" mesh is (k // ny, k % ny)
    ci, cj = pi.min(axis=1), pj.min(axis=1)
    if np.any(np.ptp(pi, axis=1) != 1) or np.any(np.ptp(pj, axis=1) != 1):
        return None
    cell = ci * (ny - 1) + cj
    n_cells = (nx - 1) * (ny - 1)
    if simplices.shape[0] != 2 * n_cells or np.any(np.bincount(cell, minlength=n_cells) != 2):
        return None
    order = np.argsort(cell, kind="stable")
    return order.reshape(n_cells, 2).astype(np.int32), np.asarray(values, dtype=float).ravel()[simplices]


_TRIANGULATION = {}  # id(parsed table) -> (the table — kept alive, so the id stays its own —, transform, cell lookup)


def triangulate(wave, axis2, values):
    """The triangulation LinearNDInterpolator builds for the table (util.py:47-53, :75-81), as arrays."""
    from scipy.spatial import Delaunay

    w_mesh, a_mesh = np.meshgrid(wave, axis2, indexing="ij")
    tri = Delaunay(np.vstack([w_mesh.ravel(), a_mesh.ravel()]).T)
    return tri.simplices, tri.transform


def sigma_file_device(tracing_lambdas, temperatures, fpath, opacity_source):
    """sigma_file for the 2-D tables with the result left in HBM -> DeviceArray (N_T, N_lambda)."""
    from stardis_amd import ops
    from stardis_amd._lib import default_context

    tracing_lambdas = np.asarray(tracing_lambdas, dtype=float)
    temperatures = np.asarray(temperatures, dtype=float)
    table = read_table(fpath, opacity_source)
    _, wave, axis2, values = table
    axis2 = np.asarray(axis2, dtype=float)
    # the triangulation belongs to the table: made once per parsed table (read_table hands out the same tuple until the file
    # changes), not per call — scipy's Delaunay is a millisecond of host time
    prepared = _TRIANGULATION.get(id(table))
    if prepared is None or prepared[0] is not table:
        simplices, transform = triangulate(wave, axis2, values)
        prepared = (table, transform, cell_lookup(wave, axis2, simplices, values), {})
        if len(_TRIANGULATION) > 16:
            _TRIANGULATION.clear()
        _TRIANGULATION[id(table)] = prepared
    transform, lookup = prepared[1], prepared[2]
    if opacity_source == "Hminus_ff":
        second, kind, what = 5040 / temperatures, 2, "H- FF"
    else:
        second, kind, what = temperatures, 1, "H2+ BF"
    if lookup is None:
        # never seen for these tables; there is no host evaluation path to fall back to
        raise NotImplementedError(f"the Delaunay triangulation of the {what} table is not two triangles per grid cell")
    ctx = default_context()
    table_dev = prepared[3].get(id(ctx))  # the table's arrays stay on the device of the context that evaluates it
    if table_dev is None or table_dev[0] is not ctx:
        table_dev = prepared[3][id(ctx)] = (ctx, ops.upload_sigma_table(wave, axis2, lookup[0], transform, lookup[1], ctx))
    dev, zero_rows = ops.sigma_table_2d(wave, axis2, lookup[0], transform, lookup[1], tracing_lambdas, second, kind, temperatures, ctx=ctx,
                                        table_dev=table_dev[1])
    if zero_rows.size:
        logger.warning(
            "Outside of interpolation range for %s cross-sections at depth points %s. Assuming 0 opacity there.", what, zero_rows
        )
    return dev


def sigma_file(tracing_lambdas, temperatures, fpath, opacity_source=None):
    """Cross-sections (N_T, N_lambda) for the 2-D tables, (N_lambda,) for Hminus_bf — as util.py:14-108."""
    table = read_table(fpath, opacity_source)
    if table[0] == "1d":
        return np.interp(np.asarray(tracing_lambdas, dtype=float), table[1], table[2])
    return sigma_file_device(tracing_lambdas, temperatures, fpath, opacity_source).numpy()


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
