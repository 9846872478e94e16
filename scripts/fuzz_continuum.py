"""GPU experiment: the continuum sources with RANDOM structure against the oracle (opacities_solvers/base.py:40-317): 0 - 3 bound-free
species with 1 - 4000 levels in all (few levels: the per-depth factors of a tile are staged in LDS; thousands: they do not fit), 0 - 3
free-free species, any subset of the Rayleigh species, electron scattering on or off, a tabulated cross-section of 2 - 3000 nodes (up to
1024 the table is searched from LDS) or none, 2 - 90 depth points, 1 - 30000 frequencies on either side of the bound-free edges and of
the Rayleigh cut-off — through the host-buffer entry point sdx_continuum_f64 (every plane) and as the continuum plane of the fused
step (sdx_synthesize_dev without lines: total_alphas), whole and as a frequency shard.  python scripts/fuzz_continuum.py FIRST LAST"""
import ctypes as C, os, sys, traceback
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import oracle
from stardis_amd import _lib, constants as K, synth

ctx = _lib.default_context()


def rel(a, b):
    m = np.abs(b) > 0
    return float(np.max(np.abs(a[m] - b[m]) / np.abs(b[m]))) if m.any() else float(np.max(np.abs(a)))


bad = 0
for seed in range(int(sys.argv[1]), int(sys.argv[2])):
    try:
        rng = np.random.default_rng(77000 + seed)
        nd = int(rng.choice([2, 7, 56, 64, 65, 90]))
        n_nu = int(rng.choice([1, 2, 65, 1000, 7634, 30000]))
        lam0 = rng.uniform(800.0, 12000.0)
        nus = synth.tracing_grid(lam0, lam0 * float(rng.choice([1.0005, 1.05, 3.0])), R=float(rng.choice([2e4, 3e5])))
        nus = np.ascontiguousarray(nus[:: max(1, nus.size // n_nu)][:n_nu])
        n_nu = nus.size
        temps = np.sort(rng.uniform(2500.0, 12000.0, nd))
        keep = []

        def ptr(a, dt=np.float64):
            a = np.ascontiguousarray(a, dtype=dt)
            keep.append(a)
            return a.ctypes.data

        c = _lib.Continuum()
        c.lambdas = ptr(K.nu_to_angstrom(nus))
        c.temperature = ptr(temps)
        want, parts = np.zeros((nd, n_nu)), {}
        desc = []
        # tabulated cross-section (calc_alpha_file, :40-71)
        if rng.random() < 0.7:
            nt = int(rng.choice([2, 85, 1024, 1025, 3000]))
            tw = np.sort(rng.uniform(500.0, 20000.0, nt)); ts = rng.uniform(0.0, 5e-17, nt); dens = 10.0 ** rng.uniform(3, 9, nd)
            c.n_table, c.table_wavelength, c.table_sigma, c.table_density = nt, ptr(tw), ptr(ts), ptr(dens)
            parts["file"] = oracle.alpha_file_1d(K.nu_to_angstrom(nus), tw, ts, dens); desc.append(f"table {nt}")
        # bound-free (:178-271)
        n_bf = int(rng.integers(0, 4))
        if n_bf:
            per = [int(rng.choice([1, 3, 40, 700, 1300])) for _ in range(n_bf)]
            off = np.concatenate([[0], np.cumsum(per)]).astype(np.int32)
            ion = rng.integers(0, 3, n_bf).astype(np.int32)
            cutoff = rng.uniform(0.2, 1.6, off[-1]) * np.median(nus) * 10.0 ** rng.uniform(-1.0, 0.5, off[-1])
            ld = 10.0 ** rng.uniform(0, 12, (off[-1], nd))
            c.bf_n_species, c.bf_n_levels = n_bf, int(off[-1])
            c.bf_species_offsets, c.bf_species_ion_number, c.bf_cutoff, c.bf_level_density = ptr(off, np.int32), ptr(ion, np.int32), ptr(cutoff), ptr(ld)
            parts["bf"] = oracle.alpha_bf(nus, off, ion, cutoff, ld); desc.append(f"bf {per}")
        # free-free (:274-317)
        n_ff = int(rng.integers(0, 4))
        if n_ff:
            ion = rng.integers(1, 4, n_ff).astype(np.int32); dens = 10.0 ** rng.uniform(15, 28, (n_ff, nd))
            c.ff_n_species, c.ff_species_ion_number, c.ff_number_density = n_ff, ptr(ion, np.int32), ptr(dens)
            parts["ff"] = oracle.alpha_ff(nus, temps, ion, dens); desc.append(f"ff {n_ff}")
        # Rayleigh (:74-135): the reference clips the caller's frequencies in place
        sp = [rng.random() < 0.5 for _ in range(3)]
        nus_after = nus.copy()
        if any(sp):
            d3 = [10.0 ** rng.uniform(8, 17, nd) if s else None for s in sp]
            c.ray_n_h, c.ray_n_he, c.ray_n_h2 = (ptr(d) if d is not None else None for d in d3)
            c.rayleigh_enabled = 1
            parts["rayleigh"] = oracle.alpha_rayleigh(nus_after, *d3); desc.append("rayleigh " + "".join("H He H2".split()[k] + " " for k in range(3) if sp[k]).strip())
        if rng.random() < 0.7:
            ne = 10.0 ** rng.uniform(9, 15, nd)
            c.electron_density = ptr(ne)
            parts["electron"] = oracle.alpha_electron(n_nu, ne); desc.append("electrons")
        for k in ("file", "bf", "ff", "rayleigh", "electron"):
            if k in parts:
                want = want + parts[k]
        if not parts:
            continue
        out = {k: np.full((nd, n_nu), np.nan) for k in ("file", "bf", "ff", "rayleigh", "electron", "total")}
        nus_io = nus.copy()
        # (an output for a source that is not configured is refused: alpha_file without a table, ...)
        ctx.call("sdx_continuum_f64", nd, n_nu, nus_io.ctypes.data, C.byref(c),
                 *(out[k].ctypes.data if (k in parts or k == "total") else None for k in ("file", "bf", "ff", "rayleigh", "electron", "total")))
        errs = {}
        for k in ("file", "bf", "ff", "rayleigh", "electron"):
            if k in parts:
                errs[k] = rel(out[k], parts[k])
                assert errs[k] < 1e-13, (k, errs[k])
        assert np.array_equal(nus_io, nus_after), "the in-place clip of the frequencies"
        assert rel(out["total"], want) < 1e-13
        # the continuum plane of the fused step (no lines): whole grid and a shard.  (A grid the Rayleigh source would zero IN PLACE
        # (:99) is not a fused-step input: calc_alphas' earlier sources see the unclipped frequencies, its later ones the clipped — the
        # drop-in call sends such grids source by source.)
        clipped = not np.array_equal(nus_after, nus)
        if clipped:
            desc.append("CLIPPED grid: fused step not applicable")
            print(f"seed {seed}: ok  depth {nd} nu {n_nu}: {'; '.join(desc)}; worst {max(errs.values()):.1e}", flush=True)
            continue
        th, w = synth.thetas_and_weights(4)
        dist = np.full(nd - 1, 3.0e6)
        ray = np.ascontiguousarray(dist.reshape(-1, 1) / np.cos(th))
        dev = lambda a, dt=np.float64: ctx.upload(np.ascontiguousarray(a, dtype=dt), dtype=dt)  # noqa: E731
        cd = _lib.Continuum()
        hold = []
        for name, _t in _lib.Continuum._fields_:
            v = getattr(c, name)
            setattr(cd, name, v) if name in ("n_table", "bf_n_species", "bf_n_levels", "ff_n_species", "rayleigh_enabled", "n_file_planes", "file_plane_ld") else None
        host_of = {a.ctypes.data: a for a in keep}
        for name in ("lambdas", "table_wavelength", "table_sigma", "table_density", "bf_species_offsets", "bf_species_ion_number", "bf_cutoff", "bf_level_density",
                     "ff_species_ion_number", "ff_number_density", "ray_n_h", "ray_n_he", "ray_n_h2", "electron_density", "temperature"):
            v = getattr(c, name)
            if v:
                a = host_of[v]
                if name == "lambdas":
                    a = K.nu_to_angstrom(nus_after)
                d = ctx.upload(a, dtype=a.dtype); hold.append(d); setattr(cd, name, d.ptr)
        d_nus, d_t, d_ray, d_w = dev(nus_after), dev(temps), dev(ray), dev(w)
        for b, cnt in ((0, n_nu), (n_nu // 3, max(1, n_nu // 2))):
            if b + cnt > n_nu:
                continue
            d_tot, d_F = ctx.empty((nd, cnt)), ctx.empty((nd, cnt))
            ctx.call("sdx_synthesize_dev", nd, n_nu, d_nus.ptr, b, cnt, 0, None, None, None, 1, None, C.byref(cd), 4, d_t.ptr, d_ray.ptr, d_w.ptr, None, d_tot.ptr,
                     d_F.ptr, cnt, None)
            tot = d_tot.numpy()
            ref_t = out["total"][:, b:b + cnt]
            if not np.array_equal(tot, ref_t):
                dm = tot != ref_t
                raise AssertionError(("fused continuum plane", b, cnt, "differing", int(dm.sum()), "NaN fused / entry", int(np.isnan(tot).sum()), int(np.isnan(ref_t).sum()),
                                      "inf", int(np.isinf(tot).sum()), int(np.isinf(ref_t).sum()), "first", np.argwhere(dm)[:3].tolist(),
                                      [float(tot[tuple(i)]) for i in np.argwhere(dm)[:3]], [float(ref_t[tuple(i)]) for i in np.argwhere(dm)[:3]]))
            assert np.isfinite(d_F.numpy()).all()
        print(f"seed {seed}: ok  depth {nd} nu {n_nu}: {'; '.join(desc)}; worst {max(errs.values()):.1e}", flush=True)
    except Exception:
        bad += 1
        print(f"seed {seed}: FAILED  [depth {nd} nu {n_nu}: {'; '.join(desc)}]", flush=True)
        traceback.print_exc()
print("failures:", bad)
