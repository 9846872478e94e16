from stardis_amd.radiation_field.radiation_field_solvers.base import raytrace  # noqa: F401
