from stardis_amd.radiation_field.base import RadiationField, create_stellar_radiation_field  # noqa: F401
