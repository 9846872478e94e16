"""Seeded random configurations of the drop-in call — create_stellar_radiation_field on the pandas stand-in for a TARDIS plasma
(the reference's end-to-end fixture G9 + the VALD atomic and molecular lists of G11) — through the fused device pass and through
the general source-by-source path: every dictionary entry, the total opacity, F_nu and (when tracked) I_nus bit for bit, same keys
in the same order.  Drawn per seed: which continuum sources are configured, line lists dense or as per-line scalars, molecules,
VALD broadening, the broadening list, spherical geometry, tracked intensities, the number of angles, a slice of the frequency
grid, the line opacity disabled.  scripts/fuzz_fused_dropin.py runs further seeds."""
import pathlib
import types

import numpy as np
import pytest

import test_gpu_round4 as T4

pytestmark = pytest.mark.gpu
NS = types.SimpleNamespace


class Patch:  # the one pytest fixture the helpers use
    def setattr(self, obj, name, value):
        setattr(obj, name, value)


def random_dropin_case(seed, tmp):
    rng = np.random.default_rng(4200 + seed)
    nus, plasma, model, cfg = T4._composite(pathlib.Path(tmp))
    desc = []
    if rng.random() < 0.5:
        cfg.line.include_molecules = True; desc.append("molecules")
    if rng.random() < 0.5:
        plasma.alpha_line_from_linelist = None; desc.append("atoms as scalars")
    if cfg.line.include_molecules and rng.random() < 0.5:
        plasma.molecule_alpha_line_from_linelist = None; desc.append("molecules as scalars")
    if rng.random() < 0.3:
        cfg.line.vald_linelist.use_vald_broadening = False; desc.append("no vald broadening")
    all_b = ["linear_stark", "quadratic_stark", "van_der_waals", "radiation"]
    keep = [b for b in all_b if rng.random() < 0.7]
    cfg.line.broadening = keep; desc.append("broadening " + ",".join(k[:3] for k in keep))
    if rng.random() < 0.15:
        cfg.line.disable = True; desc.append("lines disabled")
    if rng.random() < 0.3:
        cfg.file = {}; desc.append("no file source")
    if rng.random() < 0.3:
        cfg.bf = {}; desc.append("no bf")
    if rng.random() < 0.3:
        cfg.ff = {}; desc.append("no ff")
    r = rng.random()
    if r < 0.25:
        cfg.rayleigh = []; desc.append("no rayleigh")
    elif r < 0.5:
        cfg.rayleigh = ["H"]; desc.append("rayleigh H")
    if rng.random() < 0.3:
        cfg.disable_electron_scattering = True; desc.append("no electron scattering")
    tracked = bool(rng.random() < 0.4)
    if rng.random() < 0.4:
        model.spherical = True
        rr = 7.0e10 + np.concatenate([[0.0], np.cumsum(np.asarray(model.geometry.dist_to_next_depth_point))])
        model.geometry = NS(dist_to_next_depth_point=model.geometry.dist_to_next_depth_point, r=rr, reference_r=rr[-int(rng.integers(2, 20))])
        desc.append("spherical")
    n_theta = int(rng.choice([1, 2, 6, 20, 33]))
    if rng.random() < 0.5:  # a slice of the grid (the line selection follows the grid)
        a = int(rng.integers(0, nus.size // 2)); b = int(rng.integers(a + 2, nus.size + 1))
        nus = nus[a:b].copy(); desc.append(f"grid [{a}:{b}]")
    config = NS(opacity=cfg, no_of_thetas=n_theta, result_options=NS(return_radiation_field=tracked))
    # (with every source switched off the atmosphere is transparent and the flux zero: both paths must still agree)
    opaque = bool(cfg.file or cfg.bf or cfg.ff or cfg.rayleigh or not cfg.disable_electron_scattering or not cfg.line.disable)
    a, b = T4._both_paths(Patch(), nus, model, plasma, config, positive=opaque)
    if not opaque:
        assert not np.asarray(b.F_nu).any(); desc.append("TRANSPARENT")
    if tracked:
        assert np.array_equal(a.I_nus, b.I_nus, equal_nan=True)
    return f"{n_theta} angles, {'tracked, ' if tracked else ''}" + "; ".join(desc)




@pytest.mark.parametrize("seed", range(8))
def test_random_dropin_configuration(ctx, seed, tmp_path):
    import stardis_amd.radiation_field.base as rf

    try:
        random_dropin_case(seed, tmp_path)
    finally:
        rf.FUSED = True
