"""Round-3 regressions: kernel choice independent of the shard, per-line window rows of unsorted lists,
caller-provided source functions (radiation_field_solvers/base.py:85-93,133), block-pool misuse, optional planes."""
import ctypes as C

import numpy as np
import pytest

import oracle
from conftest import rel_err
from stardis_amd import _lib, ops, synth
from stardis_amd.engine import SpectralSynthesizer, shard_bounds

pytestmark = pytest.mark.gpu


def grid_workload(n_nu, n_lines=300, n_theta=20, seed=3):
    atm = synth.solar_atmosphere()
    nus = synth.tracing_grid(6500.0, 6600.0, None, None, n_override=n_nu)
    lines = synth.synth_lines(nus, atm, n_lines, seed=seed, mix=(0.85, 0.12, 0.03))
    th, w = synth.thetas_and_weights(n_theta)
    return atm, nus, lines, synth.synth_continuum_state(atm), th, w


@pytest.mark.parametrize("n_nu", [9000, 12000])
def test_shards_run_the_kernel_the_whole_grid_runs(ctx, n_nu):
    """With 20 angles the segmented formal solution is chosen for grids under 9216 points.  A 12000-point grid takes the plain
    kernel; its two 6000-point shards must take it too (and the 9000-point grid's shards the segmented one): the two kernels
    differ by a few ulp, so the union of the shards would otherwise not be the single-GPU result bit for bit."""
    atm, nus, lines, cont, th, w = grid_workload(n_nu)
    full = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, lines, cont, ctx=ctx)
    full.step()
    parts = []
    for rank in range(2):
        s = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, lines, cont, ctx=ctx, shard=shard_bounds(n_nu, 2, rank))
        s.step()
        parts.append(s.F_nu())
    assert np.array_equal(np.concatenate(parts, axis=1), full.F_nu())


def test_explicit_kernel_choice_changes_only_the_last_bits(ctx):
    atm, nus, lines, cont, th, w = grid_workload(6000)
    out = {}
    try:
        for mode in (0, 1):
            ctx.set_option("segmented_raytrace", mode)
            syn = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, lines, cont, ctx=ctx)
            syn.step()
            out[mode] = syn.F_nu()
    finally:
        ctx.set_option("segmented_raytrace", -1)
    assert rel_err(out[0][1:], out[1][1:]) < 1e-12
    syn = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, lines, cont, ctx=ctx)
    syn.step()
    assert np.array_equal(syn.F_nu(), out[1])  # 6000 points: the default is the segmented kernel


def test_line_windows_rows_follow_the_callers_order(ctx):
    """The window rule is per line (base.py:556-575): row k of line_windows belongs to line k of the list as given, sorted or not."""
    atm, nus, lines, *_ = grid_workload(3000, n_lines=120)
    rng = np.random.default_rng(8)
    perm = rng.permutation(lines["line_nus"].size)
    shuffled = {k: np.ascontiguousarray(v[perm]) for k, v in lines.items()}
    lo, hi = ops.line_windows(56, nus, lines["line_nus"], lines["doppler_widths"], lines["gammas"], lines["alphas"])
    lo_s, hi_s = ops.line_windows(56, nus, shuffled["line_nus"], shuffled["doppler_widths"], shuffled["gammas"], shuffled["alphas"])
    assert np.array_equal(lo_s, lo[perm]) and np.array_equal(hi_s, hi[perm])
    for k in (0, 5, 77):
        for d in (0, 30, 55):
            assert (lo_s[k, d], hi_s[k, d]) == oracle.window(nus, shuffled["line_nus"][k], shuffled["gammas"][k, d], shuffled["doppler_widths"][k, d],
                                                             shuffled["alphas"][k, d])
    # the opacity of the shuffled list is the sorted list's (calc_alan_entries accepts any order, :548-590)
    a = ops.calc_alan_entries(56, nus, lines["line_nus"], lines["doppler_widths"], lines["gammas"], lines["alphas"])
    b = ops.calc_alan_entries(56, nus, shuffled["line_nus"], shuffled["doppler_widths"], shuffled["gammas"], shuffled["alphas"])
    assert np.array_equal(a, b)


@pytest.mark.parametrize("n_nu,n_theta", [(700, 4), (700, 20), (12000, 20)])
def test_foreign_source_function_is_evaluated_on_the_host(ctx, n_nu, n_theta):
    """raytrace() calls whatever source_function(nus, temps) the field carries (radiation_field_solvers/base.py:85-93,133):
    a callable that is not blackbody_flux_at_nu is evaluated on the host and its plane handed to the kernel.  Planck as a
    foreign callable reproduces the in-kernel Planck run; twice the source gives exactly twice the intensities (the recurrence
    is linear in S and a factor 2 is exact)."""
    from types import SimpleNamespace as NS

    from stardis_amd.radiation_field.radiation_field_solvers import raytrace
    from stardis_amd.radiation_field.source_functions.blackbody import blackbody_flux_at_nu

    atm, nus, lines, cont, th, w = grid_workload(n_nu, n_theta=n_theta)
    syn = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, lines, cont, ctx=ctx)
    syn.step()
    total = syn.total_alphas()
    model = NS(spherical=False, geometry=NS(dist_to_next_depth_point=atm["dist"]), temperatures=atm["temperatures"], no_of_depth_points=56)

    def field(source, track=False):
        f = NS(frequencies=nus, source_function=source, thetas=th, I_nus_weights=w, track_individual_intensities=track,
               F_nu=np.zeros((56, n_nu)), opacities=NS(total_alphas=total))
        if track:
            f.I_nus = np.zeros((56, n_nu, n_theta))
        return f

    calls = []

    def my_planck(nu, temps):
        calls.append((np.shape(nu), np.shape(temps)))
        return oracle.blackbody_flux_at_nu(np.asarray(nu), np.asarray(temps).reshape(-1))

    def twice(nu, temps):
        return 2.0 * my_planck(nu, temps)

    ref = raytrace(model, field(blackbody_flux_at_nu)).copy()
    assert np.array_equal(ref, syn.F_nu())
    f1 = field(my_planck, track=True)
    got = raytrace(model, f1).copy()
    assert calls and calls[0] == ((n_nu,), (56, 1))
    assert rel_err(got[1:], ref[1:]) < 1e-12
    f2 = field(twice, track=True)
    assert np.array_equal(raytrace(model, f2), 2.0 * got) and np.array_equal(f2.I_nus, 2.0 * f1.I_nus)
    with pytest.raises(ValueError, match="shape"):
        ops.raytrace_arrays(nus, atm["temperatures"], np.asarray(atm["dist"]).reshape(-1, 1) / np.cos(th), w, total, source=np.zeros((3, 3)))


def test_block_pool_rejects_a_double_free(ctx):
    p = ctx.lib.sdx_malloc(ctx.handle, 4096)
    assert p
    assert ctx.lib.sdx_free(ctx.handle, p) == 0
    assert ctx.lib.sdx_free(ctx.handle, p) == -1 and b"double free" in ctx.lib.sdx_last_error_string()
    bogus = C.c_void_p(0x1000)
    assert ctx.lib.sdx_free(ctx.handle, bogus) == -1
    assert ctx.lib.sdx_free(ctx.handle, None) == 0


def test_block_pool_is_thread_safe(ctx):
    """DeviceArray.__del__ may run on any thread while another allocates or copies: the pool and the bounce buffer are
    locked (uploads from several threads arrive intact)."""
    import threading

    errors = []

    def worker(seed):
        rng = np.random.default_rng(seed)
        try:
            for _ in range(60):
                a = rng.standard_normal(rng.integers(1, 20000))
                d = ctx.upload(a)
                if not np.array_equal(d.numpy(), a):
                    errors.append("corrupt")
                d.free()
        except Exception as exc:  # noqa: BLE001
            errors.append(repr(exc))

    threads = [threading.Thread(target=worker, args=(k,)) for k in range(6)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors[:3]


def test_planes_that_were_not_kept_are_not_readable(ctx):
    atm, nus, lines, cont, th, w = grid_workload(900, n_theta=4)
    syn = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, lines, cont, ctx=ctx, keep_line=False, keep_total=False)
    syn.step()
    assert syn.d_total is None and syn.d_line is None
    with pytest.raises(RuntimeError, match="keep_total"):
        syn.total_alphas()
    with pytest.raises(RuntimeError, match="keep_line"):
        syn.alpha_line()
    both = SpectralSynthesizer(nus, atm["temperatures"], atm["dist"], th, w, lines, cont, ctx=ctx)
    both.step()
    assert np.array_equal(both.F_nu(), syn.F_nu())
    syn.keep_total = True
    syn.step()
    assert np.array_equal(syn.total_alphas(), both.total_alphas())
