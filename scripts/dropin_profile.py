"""GPU experiment: where the host time of one drop-in call (RadiationField + calc_alphas + raytrace on the pandas stand-in) goes.
python scripts/dropin_profile.py [S-c1|S-c2]"""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from stardis_amd import synth
from stardis_amd.radiation_field import RadiationField
from stardis_amd.radiation_field.opacities.opacities_solvers import calc_alphas
from stardis_amd.radiation_field.radiation_field_solvers import raytrace
from stardis_amd.radiation_field.source_functions.blackbody import blackbody_flux_at_nu

tag = sys.argv[1] if len(sys.argv) > 1 else "S-c1"
cfg = synth.WORKLOADS[tag]
atm = synth.solar_atmosphere()
nus = synth.tracing_grid(cfg["lam0"], cfg["lam1"], cfg.get("R"), cfg.get("step"))
plasma, model, config, arrays = synth.fake_plasma(nus, atm, 2000, synth.SEED)


def one():
    field = RadiationField(nus.copy(), blackbody_flux_at_nu, model, synth.N_THETAS)
    calc_alphas(plasma, model, field, config.opacity)
    raytrace(model, field)
    return field


for _ in range(3):
    one()
t0 = time.perf_counter()
for _ in range(20):
    one()
print(tag, "steady ms per call", (time.perf_counter() - t0) / 20 * 1e3)
pr = cProfile.Profile()
pr.enable()
for _ in range(20):
    one()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
