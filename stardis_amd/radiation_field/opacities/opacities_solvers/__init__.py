from stardis_amd.radiation_field.opacities.opacities_solvers.base import calc_alphas  # noqa: F401
