"""The north star's named drop-in surface: stardis_amd.base.run_stardis / patch_stardis (reference: stardis/base.py:13-45,
:120-141) and the fused create_stellar_radiation_field behind them.  The reference package, TARDIS and astropy are not
installed on the GPU box: minimal stand-ins for `stardis.base`, `stardis.io.base`, `stardis.plasma` and `astropy.units` are put
into sys.modules (they hand back the pandas stand-in plasma / model / config the other drop-in tests use); everything between
them — the wavelength -> frequency conversion, the radiation field, the output object's inputs — is the product's."""
import sys
import types

import numpy as np
import pytest

from conftest import rel_err
from stardis_amd import constants as K
from stardis_amd import synth

pytestmark = pytest.mark.gpu
NS = types.SimpleNamespace


class Quantity:
    """The three things the pipeline asks of an astropy Quantity: .value, .unit, .to(unit, equivalencies)."""

    def __init__(self, value, unit):
        self.value, self.unit = np.asarray(value, dtype=np.float64), unit

    def __len__(self):
        return len(self.value)

    def to(self, unit, equivalencies=None):
        if unit == self.unit:
            return Quantity(self.value.copy(), unit)
        assert equivalencies == "spectral" and {unit, self.unit} == {"Hz", "AA"}
        return Quantity(K.C_CGS * 1.0e8 / self.value, unit)  # nu = c / lambda


def install_stubs(monkeypatch, plasma, model, config, calls):
    units = types.ModuleType("astropy.units")
    units.Hz, units.AA, units.spectral = "Hz", "AA", lambda: "spectral"
    astropy = types.ModuleType("astropy")
    astropy.units = units

    class STARDISOutput:  # the quantities stardis/base.py:120-141 derives from the radiation field
        def __init__(self, result_options, stellar_model, stellar_plasma, stellar_radiation_field):
            self.stellar_radiation_field = stellar_radiation_field
            self.nus = stellar_radiation_field.frequencies
            self.lambdas = self.nus.to("AA", "spectral")
            F_nu = stellar_radiation_field.F_nu
            self.spectrum_nu = F_nu[-1]
            self.spectrum_lambda = (F_nu * self.nus.value / self.lambdas.value)[-1]

    def reference_radiation_field(*a, **k):
        raise AssertionError("the reference's create_stellar_radiation_field must not run")

    base = types.ModuleType("stardis.base")
    base.STARDISOutput = STARDISOutput
    base.set_num_threads = lambda n: calls.append(("set_num_threads", n))
    base.create_stellar_radiation_field = reference_radiation_field

    def parse_config_to_model(config_fname, add_config_dict=None):
        calls.append(("parse_config_to_model", config_fname, add_config_dict))
        return config, "adata", model

    def create_stellar_plasma(stellar_model, adata, cfg):
        calls.append(("create_stellar_plasma", adata))
        return plasma

    def run_stardis(config_fname, tracing_lambdas_or_nus, add_config_dict=None):  # the call sequence of stardis/base.py:34-45
        tracing_nus = tracing_lambdas_or_nus.to(units.Hz, units.spectral())
        cfg, adata, stellar_model = parse_config_to_model(config_fname, add_config_dict)
        base.set_num_threads(cfg.n_threads)
        field = base.create_stellar_radiation_field(tracing_nus, stellar_model, create_stellar_plasma(stellar_model, adata, cfg), cfg)
        return STARDISOutput(cfg.result_options, stellar_model, plasma, field)

    base.run_stardis = run_stardis
    io_base = types.ModuleType("stardis.io.base")
    io_base.parse_config_to_model = parse_config_to_model
    plasma_mod = types.ModuleType("stardis.plasma")
    plasma_mod.create_stellar_plasma = create_stellar_plasma
    stardis = types.ModuleType("stardis")
    stardis.base, stardis.plasma = base, plasma_mod
    io = types.ModuleType("stardis.io")
    io.base = io_base
    for name, mod in {"astropy": astropy, "astropy.units": units, "stardis": stardis, "stardis.base": base, "stardis.io": io,
                      "stardis.io.base": io_base, "stardis.plasma": plasma_mod}.items():
        monkeypatch.setitem(sys.modules, name, mod)
    return base


def oracle_flux(nus, atm, field, arrays):
    import oracle

    od = field.opacities.opacities_dict
    cont = synth.synth_continuum_state(atm)
    cutoff = (cont["ionization_energy"] - cont["level_excitation"]) / K.H_CGS
    total = oracle.alpha_file_1d(K.nu_to_angstrom(nus), cont["hminus_bf_wavelength"], cont["hminus_bf_cross_section"], cont["n_hminus"])
    total = total + oracle.alpha_bf(nus, [0, len(cutoff)], [0], cutoff, cont["level_density"])
    total = total + oracle.alpha_ff(nus, atm["temperatures"], [1], cont["n_e"] * cont["n_h2"])
    total = total + oracle.alpha_electron(nus.size, cont["n_e"])
    total = total + oracle.calc_alan_entries(56, nus, arrays["line_nus"], np.asarray(od["alpha_line_at_nu_doppler_widths"]),
                                             np.asarray(od["alpha_line_at_nu_gammas"]), arrays["alphas"])
    F, _ = oracle.raytrace(nus, atm["temperatures"], atm["dist"], field.thetas, field.I_nus_weights, total)
    return F, total


@pytest.mark.parametrize("fused", [True, False])
def test_run_stardis_and_patch_stardis_end_to_end(ctx, monkeypatch, fused):
    import stardis_amd.base as gpu_base
    import stardis_amd.radiation_field.base as rf

    monkeypatch.setattr(rf, "FUSED", fused)
    atm = synth.solar_atmosphere()
    lambdas = np.arange(6555.0, 6575.0, 0.02)  # ascending wavelengths, what a user passes (stardis/base.py:34)
    nus = K.C_CGS * 1.0e8 / lambdas
    plasma, model, config, arrays = synth.fake_plasma(nus, atm, 400, seed=77)
    config.n_threads = 3
    config.result_options = NS(return_model=False, return_plasma=False, return_radiation_field=False)
    calls = []
    base = install_stubs(monkeypatch, plasma, model, config, calls)

    sim = gpu_base.run_stardis("sun.yml", Quantity(lambdas, "AA"), add_config_dict={"n_threads": 3})
    assert [c[0] for c in calls] == ["parse_config_to_model", "set_num_threads", "create_stellar_plasma"]
    assert calls[0][1:] == ("sun.yml", {"n_threads": 3}) and calls[1][1] == 3
    field = sim.stellar_radiation_field
    assert type(field).__name__ == "RadiationField" and field.frequencies.unit == "Hz"
    assert np.array_equal(field.frequencies.value, nus) and np.all(np.diff(field.frequencies.value) < 0)  # lambda ascending -> nu descending
    assert sim.spectrum_nu.shape == (lambdas.size,) and np.array_equal(sim.spectrum_nu, field.F_nu[-1])
    assert np.array_equal(sim.spectrum_lambda, (field.F_nu * nus / lambdas)[-1])  # stardis/base.py:137-141
    F_ref, total_ref = oracle_flux(nus, atm, field, arrays)
    assert rel_err(field.F_nu[-1], F_ref[-1]) < 1e-10 and rel_err(field.F_nu[1:], F_ref[1:]) < 1e-10
    assert rel_err(field.opacities.total_alphas, total_ref) < 1e-12
    keys = ["alpha_file_Hminus_bf", "alpha_bf", "alpha_ff", "alpha_rayleigh", "alpha_electron", "alpha_line_at_nu", "alpha_line_at_nu_gammas",
            "alpha_line_at_nu_doppler_widths"]
    assert list(field.opacities.opacities_dict.keys()) == keys  # opacities_solvers/base.py:655-736

    # patch_stardis: the reference's own run_stardis now reaches the GPU radiation field through its module global (:5, :39)
    assert gpu_base.patch_stardis() is base and base.create_stellar_radiation_field is rf.create_stellar_radiation_field
    sim2 = base.run_stardis("sun.yml", Quantity(lambdas, "AA"))
    assert np.array_equal(sim2.spectrum_nu, sim.spectrum_nu) and np.array_equal(sim2.spectrum_lambda, sim.spectrum_lambda)


def test_fused_and_general_paths_agree_bit_for_bit(ctx, monkeypatch, tmp_path):
    """create_stellar_radiation_field as one fused device pass (lazy dictionary entries) against the source-by-source path,
    on the reference's own end-to-end fixtures (G9: TARDIS lines, VALD lines, VALD lines with classical broadening): every
    dictionary entry, the total and F_nu are identical, keys in the reference's order, and both match the reference's output."""
    import stardis_amd.radiation_field.base as rf
    from test_gpu_dropin import rebuild

    for tag in ("tardis", "vald", "vald_nb"):
        g, plasma, model, cfg = rebuild(tag, tmp_path)
        config = NS(opacity=cfg, no_of_thetas=6, result_options=NS(return_radiation_field=False))
        fields = {}
        for fused in (True, False):
            monkeypatch.setattr(rf, "FUSED", fused)
            fields[fused] = rf.create_stellar_radiation_field(g["nus"].copy(), model, plasma, config)
        a, b = fields[True], fields[False]
        assert type(a.opacities).__name__ == "FusedOpacities" and type(b.opacities).__name__ == "Opacities"
        assert np.array_equal(a.F_nu, b.F_nu) and rel_err(a.F_nu, g["F_nu"]) < 1e-10
        assert list(a.opacities.opacities_dict.keys()) == list(b.opacities.opacities_dict.keys()) == [str(k) for k in g["dict_keys"]]
        for key in b.opacities.opacities_dict:
            va, vb = a.opacities.opacities_dict[key], b.opacities.opacities_dict[key]
            assert np.shape(va) == np.shape(vb) and np.array_equal(np.asarray(va), np.asarray(vb)), (tag, key)
            assert a.opacities.opacities_dict[key] is va  # materialised once
        assert a.opacities.total_alphas is a.opacities.total_alphas
        assert np.array_equal(a.opacities.total_alphas, b.opacities.total_alphas) and rel_err(a.opacities.total_alphas, g["total_alphas"]) < 1e-12
        assert np.array_equal(a.thetas, b.thetas) and np.array_equal(a.I_nus_weights, b.I_nus_weights)
        # Opacities.calc_total_alphas keeps its += semantics on the lazy object too (opacities/base.py:24-28)
        before = a.opacities.total_alphas.copy()
        again = a.opacities.calc_total_alphas()
        assert again is a.opacities.total_alphas and rel_err(again, 2.0 * before) < 1e-15


def test_fused_path_declines_what_it_does_not_cover(ctx, monkeypatch, tmp_path):
    import stardis_amd.radiation_field.base as rf
    from stardis_amd.radiation_field import fused
    from stardis_amd.radiation_field.source_functions.blackbody import blackbody_flux_at_nu
    from test_gpu_dropin import rebuild

    g, plasma, model, cfg = rebuild("vald", tmp_path)
    config = NS(opacity=cfg, no_of_thetas=4, result_options=NS(return_radiation_field=False))
    args = (rf.RadiationField, g["nus"].copy(), model, plasma, config, blackbody_flux_at_nu)
    assert fused.try_fused(*args) is not None
    config.result_options.return_radiation_field = True  # tracked intensities: covered (I_nus stays on the device until read)
    field = rf.create_stellar_radiation_field(g["nus"].copy(), model, plasma, config)
    assert type(field.opacities).__name__ == "FusedOpacities" and isinstance(field, rf.RadiationField)
    assert field.I_nus.shape == (56, g["nus"].size, 4)
    monkeypatch.setattr(rf, "FUSED", False)
    assert np.array_equal(field.I_nus, rf.create_stellar_radiation_field(g["nus"].copy(), model, plasma, config).I_nus)
    monkeypatch.setattr(rf, "FUSED", True)
    config.result_options.return_radiation_field = False
    model.spherical, model.geometry.reference_r = True, 0.97 * model.geometry.r[-1]  # spherical models: covered (tests/test_gpu_round4.py)
    assert fused.try_fused(*args) is not None
    model.spherical = False
    # a source function other than the Planck function: evaluated on the host as the reference calls it (:133), covered
    def grey(nu, t):
        return 1.0e-13 * t**4 * (nu / nu[0]) ** 2

    own = fused.try_fused(rf.RadiationField, g["nus"].copy(), model, plasma, config, grey)
    assert own is not None and own.source_function is grey
    from stardis_amd.radiation_field.radiation_field_solvers import raytrace

    general = rf.RadiationField(g["nus"].copy(), grey, model, 4)
    general.opacities.total_alphas = own.opacities.total_alphas.copy()
    raytrace(model, general)
    assert np.array_equal(own.F_nu, general.F_nu) and not np.array_equal(own.F_nu, fused.try_fused(*args).F_nu)
    assert fused.try_fused(rf.RadiationField, g["nus"].copy(), model, plasma, config, lambda nu, t: np.zeros((3, 3))) is None  # does not broadcast
    hot = g["nus"].copy()
    hot[0] = 2.4e15  # the Rayleigh cut-off would clip the caller's grid in place (opacities_solvers/base.py:99)
    assert fused.try_fused(rf.RadiationField, hot, model, plasma, config, blackbody_flux_at_nu) is None
    cfg.line.disable = True
    cfg.disable_electron_scattering = True
    quiet = fused.try_fused(*args)
    od = quiet.opacities.opacities_dict
    assert od["alpha_electron"] == 0 and od["alpha_line_at_nu"] == 0 and od["alpha_line_at_nu_gammas"] == 0
    monkeypatch.setattr(rf, "FUSED", False)
    plain_field = rf.create_stellar_radiation_field(g["nus"].copy(), model, plasma, config)
    assert np.array_equal(quiet.F_nu, plain_field.F_nu)


def test_plasma_tables_are_cached_per_object_and_verified_by_value(ctx, monkeypatch, tmp_path):
    """The fused path keeps what it derives from the plasma's pandas objects per object AND checks a digest of their values on
    every call: the same objects with the same values hit the cache, a replaced table (what a recomputed plasma hands out) misses
    it, and so does ANY edit in place — the call returns what the reference, which recomputes everything per call
    (radiation_field/base.py:71-117), returns for the same call sequence.  No clear_cache() anywhere."""
    import stardis_amd.radiation_field.base as rf
    from stardis_amd.radiation_field import fused
    from test_gpu_dropin import rebuild

    g, plasma, model, cfg = rebuild("vald", tmp_path)
    config = NS(opacity=cfg, no_of_thetas=4, result_options=NS(return_radiation_field=False))

    def run(use_fused=True):
        monkeypatch.setattr(rf, "FUSED", use_fused)
        field = rf.create_stellar_radiation_field(g["nus"].copy(), model, plasma, config)
        assert (type(field.opacities).__name__ == "FusedOpacities") == use_fused
        return field.F_nu

    fused.clear_cache()
    first = run()
    n_cached = len(fused._MEMO)
    assert n_cached >= 3 and np.array_equal(run(), first) and len(fused._MEMO) == n_cached  # second call: all hits
    stronger = plasma.alpha_line_from_linelist.copy()
    stronger[stronger.columns[:-1]] *= 3.0  # every column but "nu"
    assert stronger.columns[-1] == "nu"
    plasma.alpha_line_from_linelist = stronger  # a NEW object: must not be served from the cache
    changed = run()
    assert not np.array_equal(changed, first) and np.array_equal(changed, run(use_fused=False))
    plasma.electron_densities = plasma.electron_densities * 1.0  # new object, same values
    assert np.array_equal(run(), changed)
    # ---- edits IN PLACE: seen by the very next call
    stronger.iloc[:, 0] *= 2.0  # one depth column of the dense alpha table (none of the values a sampled fingerprint would read)
    edited = run()
    assert not np.array_equal(edited, changed) and np.array_equal(edited, run(use_fused=False))
    stronger.iloc[stronger.shape[0] // 3, 5] *= 50.0  # ONE element
    one = run()
    assert not np.array_equal(one, edited) and np.array_equal(one, run(use_fused=False))
    plasma.electron_densities *= 1.5  # Thomson scattering, the Stark widths
    ne = run()
    assert not np.array_equal(ne, one) and np.array_equal(ne, run(use_fused=False))
    plasma.ion_number_density.iloc[:, :] = plasma.ion_number_density.to_numpy() * 0.7  # free-free, Rayleigh, van der Waals, H-
    ions = run()
    assert not np.array_equal(ions, ne) and np.array_equal(ions, run(use_fused=False))
    plasma.level_number_density.iloc[2, :] *= 4.0  # bound-free (n = 3: its edge, 3.65e14 Hz, lies below the grid's frequencies)
    lev = run()
    assert not np.array_equal(lev, ions) and np.array_equal(lev, run(use_fused=False))
    plasma.lines_from_linelist["A_ul"] *= 30.0  # a per-line scalar: radiation damping
    aul = run()
    assert not np.array_equal(aul, lev) and np.array_equal(aul, run(use_fused=False))
    # ---- the dictionary entries of a field belong to ITS call, whatever happens to the plasma afterwards (the reference forms
    # them eagerly): read late, after another edit, they are the planes of the call that made the field
    monkeypatch.setattr(rf, "FUSED", False)
    eager = rf.create_stellar_radiation_field(g["nus"].copy(), model, plasma, config)
    monkeypatch.setattr(rf, "FUSED", True)
    lazy = rf.create_stellar_radiation_field(g["nus"].copy(), model, plasma, config)
    plasma.electron_densities *= 2.0
    plasma.level_number_density.iloc[:, :] = 0.0
    plasma.ion_number_density.iloc[:, :] = plasma.ion_number_density.to_numpy() * 3.0
    for key, value in eager.opacities.opacities_dict.items():
        assert np.array_equal(np.asarray(lazy.opacities.opacities_dict[key]), np.asarray(value)), key
    assert np.array_equal(lazy.opacities.total_alphas, eager.opacities.total_alphas)
    # ---- and with the cache switched off altogether: the same answers
    monkeypatch.setattr(fused, "CACHE", False)
    assert np.array_equal(run(), run(use_fused=False))


def test_fused_path_with_no_line_on_the_grid(ctx, monkeypatch, tmp_path):
    """A grid between the lines: the selection of calc_alpha_line_at_nu (:392-395) is empty.  Fused and general path agree
    (continuum only), the dictionary has its eight keys, the line entry is a zero plane."""
    import stardis_amd.radiation_field.base as rf
    from test_gpu_dropin import rebuild

    g, plasma, model, cfg = rebuild("vald", tmp_path)
    config = NS(opacity=cfg, no_of_thetas=4, result_options=NS(return_radiation_field=False))
    line_nu = np.sort(plasma.lines_from_linelist.nu.to_numpy())
    gap = int(np.argmax(np.diff(line_nu)))
    nus = np.linspace(line_nu[gap + 1], line_nu[gap], 64)[1:-1].copy()  # strictly inside the widest gap, descending
    assert nus[0] > nus[-1]
    fields = {}
    for fused_on in (True, False):
        monkeypatch.setattr(rf, "FUSED", fused_on)
        fields[fused_on] = rf.create_stellar_radiation_field(nus.copy(), model, plasma, config)
    a, b = fields[True], fields[False]
    assert type(a.opacities).__name__ == "FusedOpacities"
    assert np.array_equal(a.F_nu, b.F_nu) and (a.F_nu[-1] > 0).all()
    assert list(a.opacities.opacities_dict.keys()) == list(b.opacities.opacities_dict.keys())
    assert not np.asarray(a.opacities.opacities_dict["alpha_line_at_nu"]).any()
    assert np.array_equal(a.opacities.total_alphas, b.opacities.total_alphas)
