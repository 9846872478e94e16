"""Round-4 review items on the GPU: the environment cannot change a result without the documented switch; the host-buffer
continuum entry point; the fused drop-in call on molecules, spherical models and line lists without dense tables."""
import hashlib
import os
import subprocess
import sys
import types

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu
NS = types.SimpleNamespace

_CHILD = r"""
import hashlib, sys
import numpy as np
sys.path.insert(0, %r)
from stardis_amd import synth
from stardis_amd.engine import SpectralSynthesizer
atm = synth.solar_atmosphere()
cont = synth.synth_continuum_state(atm)
th, w = synth.thetas_and_weights(20)
out = []
for grid, n_lines, seed in (((6560.0, 6570.0, dict(step=0.01)), 400, 81), ((4000.0, 5000.0, dict(R=1.0e5)), 9000, 82)):
    nus = synth.tracing_grid(grid[0], grid[1], **grid[2])
    lines = synth.synth_lines(nus, atm, n_lines, seed=seed, mix=(0.7, 0.25, 0.05))
    for shard in (None, (nus.size // 3, nus.size // 2)):
        syn = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, lines, cont, shard=shard, track_evaluations=False)
        syn.step()
        out.append(hashlib.sha256(np.ascontiguousarray(syn.F_nu()).tobytes() + np.ascontiguousarray(syn.alpha_line()).tobytes()).hexdigest())
print("HASHES", " ".join(out))
""" % ROOT

HOSTILE = dict(SDX_WIDE_BLOCKS="7", SDX_RT_SEG="0", SDX_RT_P="2", SDX_RT_NS="4", SDX_NARROW_F="2", SDX_R_MIXED="8", SDX_NO_CULL="1",
               SDX_NARROW_ORDER="1", SDX_WIDE_GROUP="2", SDX_CONT_DGS="0", SDX_NO_HSCAN="1", SDX_NO_CONT_RIDE="1", SDX_NO_PREPASS_FRONT="1",
               SDX_PRE_LINES="32", SDX_NO_NARROW_SUBSETS="1", SDX_NARROW_SUBSETS_DENSITY="0", SDX_FAR="1", SDX_FAR_RF="1", SDX_FAR_SPLIT="3", SDX_FAR_LAUNCH="1")


def _hashes(extra_env):
    env = {k: v for k, v in os.environ.items() if not k.startswith("SDX_")}
    env.update(extra_env)
    r = subprocess.run([sys.executable, "-c", _CHILD], env=env, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout + r.stderr
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("HASHES")][-1]
    return line.split()[1:]


def test_a_hostile_environment_does_not_change_a_bit_without_the_switch():
    """The library's experiment knobs (include/stardis_hip.h, "Environment") choose kernels and summation orders; ranks of one
    spectrum that inherit different environments would disagree in the last bits.  They are read only under SDX_EXPERIMENT=1:
    the same syntheses (a small grid on the segmented formal solution, a long list on the indexed path, whole and as a shard)
    in a clean environment and in a hostile one without the switch give identical bytes; with the switch the knobs are live."""
    clean = _hashes({})
    assert _hashes(HOSTILE) == clean
    assert _hashes(dict(HOSTILE, SDX_EXPERIMENT="0")) == clean
    live = _hashes(dict(HOSTILE, SDX_EXPERIMENT="1"))
    assert live != clean  # SDX_RT_SEG=0 / SDX_WIDE_BLOCKS=7 change the last bits: the knobs do reach the library under the switch


def test_host_buffer_continuum_entry_point_against_the_reference(ctx):
    """sdx_continuum_f64 — numpy arrays in, every continuum plane of calc_alphas out (the last export without a
    host-buffer form) — against the reference's own outputs (G5: both grids, the wide one crossing the bound-free edges, the table
    ends and the Rayleigh cut-off, whose in-place clipping of the caller's frequencies is reproduced)."""
    import ctypes as C
    import json

    from conftest import load_golden, rel_err
    from stardis_amd import _lib
    from stardis_amd import constants as K

    g = load_golden("g5_continuum")
    with open(os.path.join(ROOT, "stardis_amd", "data", "hminus_bf_wishart1979.json")) as fh:
        tab = json.load(fh)
    keep = []

    def ptr(a, dt=np.float64):
        a = np.ascontiguousarray(a, dtype=dt)
        keep.append(a)
        return a.ctypes.data

    nd = g["temperatures"].size
    cutoff = (g["ionization_energy"] - g["level_excitation"]) / K.H_CGS
    for tag in ("opt", "wide"):
        nus = g[tag + "_nus"].copy()
        c = _lib.Continuum()
        c.lambdas = ptr(g[tag + "_lambdas"])
        c.n_table = len(tab["wavelength"])
        c.table_wavelength, c.table_sigma, c.table_density = ptr(tab["wavelength"]), ptr(tab["cross_section"]), ptr(g["n_hminus"])
        c.bf_n_species, c.bf_n_levels = 1, cutoff.size
        c.bf_species_offsets, c.bf_species_ion_number = ptr([0, cutoff.size], np.int32), ptr([0], np.int32)
        c.bf_cutoff, c.bf_level_density = ptr(cutoff), ptr(g["level_density"])
        c.ff_n_species, c.ff_species_ion_number, c.ff_number_density = 1, ptr([1], np.int32), ptr(g["n_e"] * g["n_h2"])
        c.ray_n_h, c.ray_n_he, c.ray_n_h2, c.rayleigh_enabled = ptr(g["n_h1"]), ptr(g["n_he1"]), ptr(g["h2_density"]), 1
        c.electron_density, c.temperature = ptr(g["n_e"]), ptr(g["temperatures"])
        out = {k: np.full((nd, nus.size), np.nan) for k in ("file", "bf", "ff", "rayleigh", "electron", "total")}
        ctx.call("sdx_continuum_f64", nd, nus.size, nus.ctypes.data, C.byref(c), *(out[k].ctypes.data for k in ("file", "bf", "ff", "rayleigh", "electron", "total")))
        assert rel_err(out["file"], g[tag + "_alpha_file_Hminus_bf"]) < 1e-15
        assert rel_err(out["bf"], g[tag + "_alpha_bf"]) < 1e-14
        assert rel_err(out["ff"], g[tag + "_alpha_ff"]) < 1e-14
        assert rel_err(out["rayleigh"], g[tag + "_alpha_rayleigh"]) < 1e-14
        assert np.array_equal(out["electron"], g[tag + "_alpha_electron"])
        assert np.array_equal(nus, g[tag + "_nus_after_rayleigh"])  # (:99)
        want = (((g[tag + "_alpha_file_Hminus_bf"] + g[tag + "_alpha_bf"]) + g[tag + "_alpha_ff"]) + g[tag + "_alpha_rayleigh"]) + g[tag + "_alpha_electron"]
        assert rel_err(out["total"], want) < 1e-14
        # the planes are those of the per-source device entry points, bit for bit, and the total is their sum in calc_alphas' order
        assert np.array_equal(out["total"], (((out["file"] + out["bf"]) + out["ff"]) + out["rayleigh"]) + out["electron"])
        # a subset of the outputs, electron scattering disabled, no Rayleigh species
        c.electron_density, c.rayleigh_enabled = None, 0
        nus2 = g[tag + "_nus"].copy()
        total2, el = np.empty((nd, nus.size)), np.full((nd, nus.size), np.nan)
        ctx.call("sdx_continuum_f64", nd, nus.size, nus2.ctypes.data, C.byref(c), None, None, None, None, el.ctypes.data, total2.ctypes.data)
        assert not el.any() and np.array_equal(nus2, g[tag + "_nus"])
        assert np.array_equal(total2, (out["file"] + out["bf"]) + out["ff"])
    with pytest.raises(ValueError):
        ctx.call("sdx_continuum_f64", nd, nus.size, nus.ctypes.data, C.byref(c), None, None, None, None, None, None)


# ------------------------------------------------------------------------------------------------ the fused drop-in call, widened
def _composite(tmp_path):
    """One pandas stand-in for a TARDIS plasma that carries everything at once: the continuum state of the reference's end-to-end
    fixture (G9) and the VALD atomic + molecular line lists of G11 (same grid, same temperature structure), dense alpha tables
    included; the tests below drop the dense tables to reach the per-line-scalar route."""
    import pandas as pd

    import linelist_fixture
    from stardis_amd.plasma.base import AlphaLineVald
    from stardis_amd.plasma.molecules import AlphaLineValdMolecule
    from test_gpu_dropin import rebuild

    g9, plasma9, model9, cfg = rebuild("vald", tmp_path)
    fx = linelist_fixture.rebuild()
    assert np.array_equal(fx.g["nus"], g9["nus"]) and np.array_equal(fx.t, g9["temperatures"])
    p = fx.plasma
    for name in ("levels", "excitation_energy", "level_number_density", "h_minus_density", "h2_density"):
        setattr(p, name, getattr(plasma9, name))
    p.ion_number_density = pd.concat([fx.ion_density, plasma9.ion_number_density.loc[[(1, 1), (2, 1)]]]).sort_index()
    alphas, lines = AlphaLineVald().calculate(fx.atomic_data, fx.ion_density, fx.t, fx.ionization_data, fx.partition)
    p.lines_from_linelist, p.alpha_line_from_linelist = lines, alphas
    m_alphas, m_lines = AlphaLineValdMolecule().calculate(fx.atomic_data, fx.mol_density, fx.t, fx.mol_partition)
    p.molecule_lines_from_linelist, p.molecule_alpha_line_from_linelist = m_lines, m_alphas
    model = fx.model
    model.geometry = model9.geometry
    cfg.line.vald_linelist.use_linelist, cfg.line.vald_linelist.use_vald_broadening = True, True
    return fx.g["nus"], p, model, cfg


def _both_paths(monkeypatch, nus, model, plasma, config, positive=True):
    import stardis_amd.radiation_field.base as rf

    fields = {}
    for fused in (True, False):
        monkeypatch.setattr(rf, "FUSED", fused)
        fields[fused] = rf.create_stellar_radiation_field(nus.copy(), model, plasma, config)
    a, b = fields[True], fields[False]
    assert type(a.opacities).__name__ == "FusedOpacities" and type(b.opacities).__name__ == "Opacities"
    assert np.array_equal(a.F_nu, b.F_nu, equal_nan=True) and np.isfinite(b.F_nu).all() and (not positive or (b.F_nu[-1] > 0).all())
    assert list(a.opacities.opacities_dict.keys()) == list(b.opacities.opacities_dict.keys())
    for key in b.opacities.opacities_dict:
        va, vb = a.opacities.opacities_dict[key], b.opacities.opacities_dict[key]
        assert np.shape(va) == np.shape(vb) and np.array_equal(np.asarray(va), np.asarray(vb)), key
    assert np.array_equal(a.opacities.total_alphas, b.opacities.total_alphas)
    return a, b


@pytest.mark.parametrize("dense_atoms,dense_molecules", [(True, True), (True, False), (False, True), (False, False)])
def test_fused_call_with_molecules_equals_the_general_path(ctx, monkeypatch, tmp_path, dense_atoms, dense_molecules):
    """include_molecules (opacities_solvers/base.py:444-484, :716-736): a second list whose plane is added after the atomic one,
    keys `molecule_*` in the reference's order — with the dense alpha tables on the plasma and without (per-line scalars, the
    pre-pass generates alpha, gamma and the Doppler width), in every combination; every entry, the total and F_nu bit for bit."""
    nus, plasma, model, cfg = _composite(tmp_path)
    cfg.line.include_molecules = True
    if not dense_atoms:
        plasma.alpha_line_from_linelist = None
    if not dense_molecules:
        plasma.molecule_alpha_line_from_linelist = None
    config = NS(opacity=cfg, no_of_thetas=6, result_options=NS(return_radiation_field=False))
    a, b = _both_paths(monkeypatch, nus, model, plasma, config)
    keys = list(b.opacities.opacities_dict.keys())
    assert keys[-6:] == ["alpha_line_at_nu", "alpha_line_at_nu_gammas", "alpha_line_at_nu_doppler_widths", "molecule_alpha_line_at_nu",
                         "molecule_alpha_line_at_nu_gammas", "molecule_alpha_line_at_nu_doppler_widths"]
    od = b.opacities.opacities_dict
    assert np.asarray(od["molecule_alpha_line_at_nu"]).any() and np.asarray(od["alpha_line_at_nu"]).any()
    assert np.shape(od["molecule_alpha_line_at_nu_gammas"])[1] == 1  # A_ul as one column (broadening.py:799-801)
    # against the reference's own line opacities for these lists (G11)
    from conftest import load_golden, rel_err

    g = load_golden("g11_linelist")
    assert rel_err(np.asarray(od["alpha_line_at_nu"]), g["full_vb_alpha_line_at_nu"]) < 1e-12
    assert rel_err(np.asarray(od["molecule_alpha_line_at_nu"]), g["molfull_alpha_line_at_nu"]) < 1e-12
    # without "radiation" the molecular gammas are the reference's (N_l, N_d) zeros (:803-806)
    cfg.line.broadening = ["linear_stark", "quadratic_stark", "van_der_waals"]
    a, b = _both_paths(monkeypatch, nus, model, plasma, config)
    assert not np.asarray(b.opacities.opacities_dict["molecule_alpha_line_at_nu_gammas"]).any()


@pytest.mark.parametrize("tracked", [False, True])
def test_fused_call_on_a_spherical_model_equals_the_general_path(ctx, monkeypatch, tmp_path, tracked):
    """spherical models (radiation_field_solvers/base.py:141-198, :296-300, :340-344): chord table, inward sweep, photospheric
    correction inside the fused step; with and without tracked intensities, and together with molecules and f1 inputs."""
    nus, plasma, model, cfg = _composite(tmp_path)
    model.spherical = True
    r = 7.0e10 + np.concatenate([[0.0], np.cumsum(np.asarray(model.geometry.dist_to_next_depth_point))])
    model.geometry = NS(dist_to_next_depth_point=model.geometry.dist_to_next_depth_point, r=r, reference_r=r[-12])
    config = NS(opacity=cfg, no_of_thetas=8, result_options=NS(return_radiation_field=tracked))
    a, b = _both_paths(monkeypatch, nus, model, plasma, config)
    if tracked:
        assert np.array_equal(a.I_nus, b.I_nus, equal_nan=True)
    cfg.line.include_molecules = True
    plasma.alpha_line_from_linelist = None
    _both_paths(monkeypatch, nus, model, plasma, config)
    # the plane-parallel result differs: the geometry did reach the step
    model.spherical = False
    import stardis_amd.radiation_field.base as rf

    monkeypatch.setattr(rf, "FUSED", True)
    flat = rf.create_stellar_radiation_field(nus.copy(), model, plasma, config)
    assert not np.array_equal(flat.F_nu, a.F_nu)


def test_fused_call_with_line_parameters_generated_on_the_device(ctx, monkeypatch, tmp_path):
    """A plasma with lines_from_linelist but no dense alpha table (f1): both paths build the same LineList and run the same
    generating pre-pass; against the reference's calc_alpha_line_at_nu on ITS dense tables (G11) within the opacity tolerance."""
    from conftest import load_golden, rel_err

    nus, plasma, model, cfg = _composite(tmp_path)
    plasma.alpha_line_from_linelist = None
    g = load_golden("g11_linelist")
    for vb in (True, False):
        cfg.line.vald_linelist.use_vald_broadening = vb
        config = NS(opacity=cfg, no_of_thetas=6, result_options=NS(return_radiation_field=False))
        a, b = _both_paths(monkeypatch, nus, model, plasma, config)
        tag = f"full_{'vb' if vb else 'nb'}_"
        od = a.opacities.opacities_dict
        assert rel_err(np.asarray(od["alpha_line_at_nu"]), g[tag + "alpha_line_at_nu"]) < 1e-12
        assert rel_err(np.asarray(od["alpha_line_at_nu_gammas"]), g[tag + "gammas"]) < 1e-13
        assert rel_err(np.asarray(od["alpha_line_at_nu_doppler_widths"]), g[tag + "doppler"]) < 1e-15


def test_fused_fields_give_their_device_memory_back(ctx, monkeypatch, tmp_path):
    """A fused RadiationField keeps device twins for its lazy entries.  release_device() materialises them on the host and frees
    the device side; a per-process budget does the same to the oldest live fields when their twins add up (round-3 advisor
    finding: a caller that keeps many outputs must not run out of HBM where the reference only holds host arrays)."""
    import stardis_amd.radiation_field.base as rf
    from stardis_amd.radiation_field import fused
    from test_gpu_dropin import rebuild

    g, plasma, model, cfg = rebuild("vald", tmp_path)
    config = NS(opacity=cfg, no_of_thetas=4, result_options=NS(return_radiation_field=True))
    monkeypatch.setattr(rf, "FUSED", True)
    ref = rf.create_stellar_radiation_field(g["nus"].copy(), model, plasma, config)
    want = {k: np.array(v) for k, v in ref.opacities.opacities_dict.items()}
    want_total, want_I = ref.opacities.total_alphas.copy(), ref.I_nus.copy()
    a = rf.create_stellar_radiation_field(g["nus"].copy(), model, plasma, config)
    assert a.opacities._device_bytes > 0
    fused.release_device(a)
    assert a.opacities._device_bytes == 0 and a.opacities._total_twin is None and a._I_dev is None and a._device_blob is None
    for k, v in want.items():
        assert np.array_equal(np.asarray(a.opacities.opacities_dict[k]), v), k
    assert np.array_equal(a.opacities.total_alphas, want_total) and np.array_equal(a.I_nus, want_I)
    # the budget: with room for two fields' twins, a third creation releases the oldest — nothing is lost
    per = ref.opacities._device_bytes
    monkeypatch.setattr(fused, "DEVICE_BUDGET_BYTES", int(2.5 * per))
    fused._LIVE.clear()
    fields = [rf.create_stellar_radiation_field(g["nus"].copy(), model, plasma, config) for _ in range(4)]
    held = [f.opacities._device_bytes for f in fields]
    assert held[0] == 0 and held[1] == 0 and held[2] > 0 and held[3] > 0
    for f in fields:
        assert np.array_equal(f.opacities.total_alphas, want_total)
        assert np.array_equal(np.asarray(f.opacities.opacities_dict["alpha_line_at_nu"]), want["alpha_line_at_nu"])
    b = rf.create_stellar_radiation_field(g["nus"].copy(), model, plasma, config)
    fused.release_device(b, materialize=False)
    with pytest.raises(RuntimeError, match="released"):
        b.opacities.opacities_dict["alpha_line_at_nu"]


@pytest.mark.gpu
def test_lazy_entries_belong_to_the_call_that_made_the_field(ctx, monkeypatch, tmp_path):
    """The continuum entries of a fused field are formed on first read.  They are formed from the plasma tables of the call
    (references taken then), not from whatever the plasma holds at the time of the read: replacing the plasma's attributes — the next
    iteration of a fit — must not give entries that disagree with the field's F_nu (round-3 advisor finding)."""
    import stardis_amd.radiation_field.base as rf
    from test_gpu_dropin import rebuild

    g, plasma, model, cfg = rebuild("vald", tmp_path)
    config = NS(opacity=cfg, no_of_thetas=4, result_options=NS(return_radiation_field=True))
    monkeypatch.setattr(rf, "FUSED", False)
    ref = rf.create_stellar_radiation_field(g["nus"].copy(), model, plasma, config)
    want = {k: np.array(v) for k, v in ref.opacities.opacities_dict.items()}
    monkeypatch.setattr(rf, "FUSED", True)
    a = rf.create_stellar_radiation_field(g["nus"].copy(), model, plasma, config)
    assert type(a.opacities).__name__ == "FusedOpacities"
    plasma.electron_densities = plasma.electron_densities * 3.0
    plasma.ion_number_density = plasma.ion_number_density * 0.5
    plasma.level_number_density = plasma.level_number_density * 2.0
    for k, v in want.items():
        assert np.array_equal(np.asarray(a.opacities.opacities_dict[k]), v), k
    b = rf.create_stellar_radiation_field(g["nus"].copy(), model, plasma, config)  # the edited plasma gives another field
    assert not np.array_equal(b.F_nu, a.F_nu)
    assert not np.array_equal(np.asarray(b.opacities.opacities_dict["alpha_electron"]), want["alpha_electron"])


@pytest.mark.gpu
@pytest.mark.parametrize("n_lines,grid", [(400, (6560.0, 6570.0, dict(step=0.01))), (9000, (4000.0, 4400.0, dict(R=1.0e5)))])
def test_options_entry_point_equals_its_parts(ctx, n_lines, grid):
    """sdx_synthesize_opt_dev called directly through the C ABI: with no option set it is sdx_synthesize_dev bit for bit; with extra
    line planes, the inward sweep and the photospheric correction it equals the same step put together from the individual entry
    points (line opacity -> total -> accumulate the planes -> sdx_raytrace_spherical_dev); bad arguments are refused with -1."""
    import ctypes as C

    from stardis_amd import _lib, synth
    from stardis_amd.engine import SpectralSynthesizer

    atm = synth.solar_atmosphere()
    cont = synth.synth_continuum_state(atm)
    th, w = synth.thetas_and_weights(20)
    nus = synth.tracing_grid(grid[0], grid[1], **grid[2])
    lines = synth.synth_lines(nus, atm, n_lines, seed=91, mix=(0.7, 0.25, 0.05))
    syn = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, lines, cont, ctx=ctx, track_evaluations=False)
    syn.keep_total = True
    syn.step()
    F0, line0, total0 = syn.F_nu().copy(), syn.alpha_line().copy(), syn.total_alphas().copy()
    nd, n = syn.n_depth, nus.size
    args = (nd, n, syn.d_nus.ptr, 0, n, syn.n_lines, syn.d_ln.ptr, syn.d_dw.ptr, syn.d_g.ptr, syn.gamma_cols, syn.d_a.ptr, C.byref(syn.cont),
            syn.n_theta, syn.d_t.ptr, syn.d_ray.ptr, syn.d_w.ptr)
    d_line, d_total, d_F = ctx.empty((nd, n)), ctx.empty((nd, n)), ctx.empty((nd, n))
    # 1. no options: the plain fused step
    opt = _lib.SynthesisOptions()
    ctx.call("sdx_synthesize_opt_dev", *args, d_line.ptr, d_total.ptr, d_F.ptr, n, C.byref(opt), None)
    assert np.array_equal(d_F.numpy(), F0) and np.array_equal(d_line.numpy(), line0) and np.array_equal(d_total.numpy(), total0)
    assert ctx.lib.sdx_synthesize_opt_dev(ctx.handle, *args, d_line.ptr, d_total.ptr, d_F.ptr, n, None, None) == -1  # the description is required
    # 2. two extra planes, inward rays, correction — against the individual entry points
    rng = np.random.default_rng(5)
    extra = [ctx.upload(np.ascontiguousarray(total0 * rng.uniform(0.0, 0.3, size=total0.shape))) for _ in range(2)]
    opt = _lib.SynthesisOptions()
    opt.inward_rays, opt.photospheric_correction, opt.n_line_planes, opt.line_plane_ld = 1, 1.21, 2, n
    opt.line_plane[0], opt.line_plane[1] = extra[0].ptr, extra[1].ptr
    ctx.call("sdx_synthesize_opt_dev", *args, d_line.ptr, d_total.ptr, d_F.ptr, n, C.byref(opt), None)
    F1, line1, total1 = d_F.numpy(), d_line.numpy(), d_total.numpy()
    r_total, r_F = ctx.upload(total0.copy()), ctx.empty((nd, n))
    for e in extra:
        ctx.call("sdx_accumulate_dev", nd, n, r_total.ptr, n, e.ptr, n)
    ctx.call("sdx_raytrace_spherical_dev", nd, n, syn.n_theta, syn.d_nus.ptr, syn.d_t.ptr, syn.d_ray.ptr, syn.d_w.ptr, r_total.ptr, n, r_F.ptr, n, None, 0,
             C.c_double(1.21))
    assert np.array_equal(line1, line0) and np.array_equal(total1, r_total.numpy())
    assert np.array_equal(F1, r_F.numpy()) and np.isfinite(F1).all() and not np.array_equal(F1, F0)
    # 3. refused: more than two planes, a missing plane, a leading dimension below the shard
    for bad in (dict(n_line_planes=3), dict(n_line_planes=1, plane0=None), dict(n_line_planes=1, line_plane_ld=n - 1)):
        o = _lib.SynthesisOptions()
        o.n_line_planes, o.line_plane_ld = bad["n_line_planes"], bad.get("line_plane_ld", n)
        o.line_plane[0] = bad.get("plane0", extra[0].ptr)
        assert ctx.lib.sdx_synthesize_opt_dev(ctx.handle, *args, d_line.ptr, d_total.ptr, d_F.ptr, n, C.byref(o), None) == -1
        assert b"synthesize" in ctx.lib.sdx_last_error_string()
    syn.close()
