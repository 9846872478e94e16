"""No-GPU checks of the drop-in boundary: the shared library loads here (hipcc cross-compiled it), exports every
symbol include/stardis_hip.h declares, the ctypes prototype table covers exactly that set, the product fails loudly
without a device, and nothing under stardis_amd/ touches the CPU oracle."""
import ctypes
import os
import re

import numpy as np
import pytest

from conftest import ROOT
from stardis_amd import _lib

HEADER = os.path.join(ROOT, "include", "stardis_hip.h")


def declared_functions():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sdx_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    names = declared_functions()
    assert len(names) >= 40
    lib = ctypes.CDLL(_lib.LIB_PATH)
    missing = [n for n in names if not hasattr(lib, n)]
    assert not missing, missing


def test_prototype_table_matches_header():
    assert sorted(_lib.PROTOTYPES) == declared_functions()


def test_continuum_struct_matches_header():
    text = open(HEADER).read()
    body = re.search(r"typedef struct sdx_continuum \{(.*?)\} sdx_continuum;", text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"\b([a-z_0-9]+)(?:\[\d+\])?\s*;", body)
    assert fields == [f[0] for f in _lib.Continuum._fields_]
    assert _lib.Continuum.file_plane.size == 4 * 8 and "file_plane[4]" in body


def test_linelist_struct_matches_header():
    text = open(HEADER).read()
    body = re.search(r"typedef struct sdx_linelist \{(.*?)\} sdx_linelist;", text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"\b([A-Za-z_0-9]+)\s*;", body)
    assert fields == [f[0] for f in _lib.LineListStruct._fields_]


def test_synthesis_options_struct_matches_header():
    text = open(HEADER).read()
    body = re.search(r"typedef struct sdx_synthesis_options \{(.*?)\} sdx_synthesis_options;", text, flags=re.S).group(1)
    body = re.sub(r"/\*.*?\*/", "", body, flags=re.S)
    fields = re.findall(r"\b([A-Za-z_0-9]+)(?:\[\d+\])?\s*;", body)
    assert fields == [f[0] for f in _lib.SynthesisOptions._fields_]
    assert _lib.SynthesisOptions.line_plane.size == 2 * 8 and "line_plane[2]" in body


def test_library_reports_no_device_and_product_raises():
    lib = _lib.load()
    assert lib.sdx_version().startswith(b"stardis_hip")
    if lib.sdx_device_count() > 0:
        pytest.skip("a GPU is visible")
    with pytest.raises(RuntimeError, match="no HIP device"):
        _lib.Context(0)
    from stardis_amd import ops

    with pytest.raises(RuntimeError):
        ops.voigt_profile(0.0, 1.0, 0.0)  # no silent CPU fallback
    assert lib.sdx_create(0, None) is None
    assert b"hipSetDevice" in lib.sdx_last_error_string() or lib.sdx_last_error_string()
    assert lib.sdx_set_device(0) == -2 and lib.sdx_last_error_code() == -2  # SDX_ERR_HIP: no device to select
    assert lib.sdx_host_alloc(None, 4096) is None and lib.sdx_last_error_code() == -1  # page-locked memory needs a context
    assert lib.sdx_host_free(None) == 0


def test_product_never_imports_the_oracle():
    bad = []
    for base, _, files in os.walk(os.path.join(ROOT, "stardis_amd")):
        for f in files:
            if f.endswith((".py", ".h", ".hip", ".cpp")):
                text = open(os.path.join(base, f)).read()
                if re.search(r"^\s*(import|from)\s+oracle\b", text, flags=re.M) or "stardis_oracle" in text or "/root/reference" in text:
                    bad.append(os.path.join(base, f))
    assert not bad, bad


def test_gpu_tests_and_bench_do_not_read_the_reference_tree():
    for rel in ("bench.py", "__graft_entry__.py", "tests/test_gpu_parity.py", "tests/test_gpu_engine.py"):
        text = open(os.path.join(ROOT, rel)).read()
        assert "/root/reference" not in text.replace("/root/reference is a Python package", ""), rel


def test_species_parser_and_shards():
    from stardis_amd.engine import shard_bounds
    from stardis_amd.util import species_string_to_tuple as s

    assert s("H I") == (1, 0) and s("He II") == (2, 1) and s("Fe II") == (26, 1)
    # digits are the spectroscopic stage too (TARDIS's own vectors for tardis.util.base.species_string_to_tuple)
    assert s("si ii") == (14, 1) and s("si 2") == (14, 1) and s("si ix") == (14, 8)
    assert s("Ca 2") == (20, 1) and s("H 1") == (1, 0) and s("H_1") == (1, 0) and s("Fe26") == (26, 25)
    for bad in ("Xx I", "H 0", "He 4", "H"):
        with pytest.raises(ValueError):
            s(bad)
    for n, world in ((7634, 1), (7634, 8), (10, 3), (5, 8), (120398, 8)):
        blocks = [shard_bounds(n, world, r) for r in range(world)]
        assert blocks[0][0] == 0 and sum(c for _, c in blocks) == n
        for (b0, c0), (b1, _) in zip(blocks, blocks[1:]):
            assert b0 + c0 == b1


def test_host_side_tables_match_reference():
    """sigma_file on the reference's own tables would need /root/reference; what can be pinned anywhere is the
    wavelength conversion and the host interpolation contract on the committed H- bf table."""
    import json

    from conftest import load_golden
    from stardis_amd import constants as K

    g = load_golden("g5_continuum")
    for tag in ("opt", "wide"):
        assert np.array_equal(K.nu_to_angstrom(g[tag + "_nus"]), g[tag + "_lambdas"])
    with open(os.path.join(ROOT, "stardis_amd", "data", "hminus_bf_wishart1979.json")) as fh:
        tab = json.load(fh)
    sig = np.interp(g["opt_lambdas"], tab["wavelength"], tab["cross_section"])
    assert np.array_equal(sig * g["n_hminus"][:, None], g["opt_alpha_file_Hminus_bf"])


def test_group_entry_fails_loudly_without_rccl_or_devices():
    """sdx_group_create on a box without GPUs: RCCL is opened first (a missing library is SDX_ERR_COMM = -3), then the devices
    are checked (SDX_ERR_ARG = -1).  Nothing falls back to one device or to the CPU."""
    import subprocess
    import sys

    code = ("import sys; sys.path.insert(0, %r)\n"
            "from stardis_amd import _lib\nlib = _lib.load()\n"
            "h = lib.sdx_group_create(1, None)\nprint('HANDLE', h, 'CODE', lib.sdx_last_error_code(), lib.sdx_last_error_string().decode())\n" % ROOT)
    proc = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, SDX_RCCL_LIB="/nonexistent/librccl.so.1"), capture_output=True, text=True,
                          timeout=300)
    assert "HANDLE None CODE -3" in proc.stdout and "RCCL not available" in proc.stdout, proc.stdout + proc.stderr
    if _lib.load().sdx_device_count() == 0:
        proc = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
        assert "HANDLE None CODE -1" in proc.stdout and "not visible" in proc.stdout, proc.stdout + proc.stderr


def test_planner_constants_are_the_librarys():
    """stardis_amd.parallel carries the far-field rule only as defaults: they equal what the library reports, and the planner asks
    the library (sdx_far_field_rule needs no device)."""
    from stardis_amd import parallel

    lib = _lib.load()
    mn, tile, near = ctypes.c_int64(), ctypes.c_int(), ctypes.c_int()
    assert lib.sdx_far_field_rule(ctypes.byref(mn), ctypes.byref(tile), ctypes.byref(near)) == 0
    assert (mn.value, near.value) == (parallel.FAR_FIELD_MIN_POINTS, parallel.FAR_NEAR_POINTS) == parallel.far_field_rule()
    assert tile.value == 256 and near.value == int(3.5 * tile.value)
    assert parallel.far_field_active(mn.value) and not parallel.far_field_active(mn.value - 1)
    assert lib.sdx_far_field_active(None, 100000) < 0  # null context
