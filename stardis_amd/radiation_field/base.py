"""Radiation field container and driver, call-compatible with stardis/radiation_field/base.py."""
import logging
import os

import numpy as np

from stardis_amd.radiation_field.opacities import Opacities
from stardis_amd.radiation_field.opacities.opacities_solvers import calc_alphas
from stardis_amd.radiation_field.radiation_field_solvers import raytrace
from stardis_amd.radiation_field.source_functions.blackbody import blackbody_flux_at_nu

logger = logging.getLogger(__name__)

try:  # keeps to_hdf() working when TARDIS is installed next to this package
    from tardis.io.util import HDFWriterMixin as _Base
except Exception:  # pragma: no cover - TARDIS is not a dependency of the hot path
    _Base = object


class RadiationField(_Base):
    """Frequencies, source function, opacities, flux and the angular quadrature.  Same attributes as the
    reference's class (:12-68): `thetas` are Gauss-Legendre nodes mapped to x/2 + pi/4 and `I_nus_weights`
    the weights times pi/2 (:61-63) — reproduced verbatim, they define the flux normalisation."""

    hdf_properties = ["frequencies", "opacities", "F_nu"]

    def __init__(self, frequencies, source_function, stellar_model, num_of_thetas, track_individual_intensities=False):
        n_depth = stellar_model.no_of_depth_points
        self.frequencies = frequencies
        self.source_function = source_function
        self.opacities = Opacities(frequencies, stellar_model)
        self.F_nu = np.zeros((n_depth, len(frequencies)))
        nodes, weights = np.polynomial.legendre.leggauss(num_of_thetas)
        self.thetas = (nodes / 2) + 0.5 * np.pi / 2
        self.I_nus_weights = weights * np.pi / 2
        self.track_individual_intensities = track_individual_intensities
        if track_individual_intensities:
            self.I_nus = np.zeros((n_depth, len(frequencies), len(self.thetas)))


FUSED = os.environ.get("STARDIS_AMD_FUSED", "1") != "0"  # one fused device pass when the configuration allows it


def create_stellar_radiation_field(tracing_nus, stellar_model, stellar_plasma, config):
    """Opacities then formal solution, as the reference's driver (:71-117).  Configurations the fused synthesis covers run
    as ONE device pass with lazily materialised dictionary entries (stardis_amd/radiation_field/fused.py); everything else —
    and STARDIS_AMD_FUSED=0 — takes the source-by-source path below.  Both produce the same numbers."""
    if FUSED:
        from stardis_amd.radiation_field.fused import try_fused

        field = try_fused(RadiationField, tracing_nus, stellar_model, stellar_plasma, config, blackbody_flux_at_nu)
        if field is not None:
            logger.info("Radiation field computed by the fused synthesis")
            return field
    field = RadiationField(
        tracing_nus,
        blackbody_flux_at_nu,
        stellar_model,
        config.no_of_thetas,
        track_individual_intensities=config.result_options.return_radiation_field,
    )
    logger.info("Calculating alphas")
    calc_alphas(stellar_plasma=stellar_plasma, stellar_model=stellar_model, stellar_radiation_field=field,
                opacity_config=config.opacity)
    logger.info("Raytracing")
    raytrace(stellar_model, field)
    return field
