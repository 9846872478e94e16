from stardis_amd.radiation_field.opacities.base import Opacities  # noqa: F401
