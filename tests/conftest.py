import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def pytest_sessionstart(session):
    """Bring the two native libraries up to date (no-ops when they are): the HIP product library — hipcc cross-compiles
    gfx950 without a GPU — and the CPU oracle.  A fresh checkout has neither (they are git-ignored build artefacts)."""
    import shutil
    import subprocess

    if shutil.which("make") is None:
        return
    for sub in (os.path.join("stardis_amd", "csrc"), "oracle"):
        try:
            subprocess.run(["make", "-C", os.path.join(ROOT, sub)], check=False, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL,
                           timeout=900)
        except Exception:  # the tests that need the library then fail with its own message
            pass


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)


_SPECIES_Z = {"H": 1, "He": 2}
_STAGE = {"I": 1, "II": 2, "III": 3}


def species_arrays_from_g14(g, keys):
    """Flat bf / ff arguments for a species list out of the G14 fixture's plasma tables, the way the reference walks them
    (opacities_solvers/base.py:204-226 and util.py:154-167): species-major, levels in the plasma's index order.
    -> (offsets, bf_ions, cutoffs, level_density, ff_ions, ff_density)."""
    H = 6.62607015e-27
    ions = {tuple(k): g["ion_number_density"][i] for i, k in enumerate(g["ion_index"])}
    chi = {tuple(k): g["ionization_energy"][i] for i, k in enumerate(g["ionization_index"])}
    offsets, bf_ions, cutoffs, dens, ff_ions, ff_dens = [0], [], [], [], [], []
    for key in keys:
        sym, stage = key.split("_")
        z = _SPECIES_Z[sym]
        ion = (int(stage) if stage.isdigit() else _STAGE[stage]) - 1
        for i, lev in enumerate(g["level_index"]):
            if lev[0] == z and lev[1] == ion:
                cutoffs.append((chi[(z, ion + 1)] - g["level_excitation"][i]) / H)
                dens.append(g["level_density"][i])
        offsets.append(len(cutoffs))
        bf_ions.append(ion)
        ff_ions.append(ion + 1)
        ff_dens.append(g["n_e"] * ions[(z, ion + 1)])  # util.py:160-165: n_e first, then the ion density
    return offsets, bf_ions, np.array(cutoffs), np.vstack(dens), ff_ions, np.vstack(ff_dens)


@pytest.fixture(scope="session")
def golden():
    return load_golden


def rel_err(a, b):
    """max |a-b| / |b| where b != 0, absolute where b == 0; NaNs must coincide."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.array_equal(np.isnan(a), np.isnan(b)), "NaN pattern differs"
    ok = ~np.isnan(b)
    a, b = a[ok], b[ok]
    if a.size == 0:
        return 0.0
    d = np.abs(a - b)
    scale = np.where(b == 0, 1.0, np.abs(b))
    return float(np.max(d / scale))


@pytest.fixture(scope="session")
def ctx():
    from stardis_amd._lib import default_context

    return default_context()
