"""Drop-in driver: the reference's pipeline (stardis/base.py:13-45) with the radiation field on MI355X.

Config/model I/O and the TARDIS plasma stay the reference's (they are imported from an installed `stardis`); only
`create_stellar_radiation_field` — opacity assembly + formal solution — is replaced.  Needs `stardis`, `tardis` and
`astropy` installed, exactly like the reference; the hot path itself needs none of them.
"""
import logging

from stardis_amd.radiation_field.base import create_stellar_radiation_field

logger = logging.getLogger(__name__)


def run_stardis(config_fname, tracing_lambdas_or_nus, add_config_dict=None):
    """Same signature and return type as stardis.base.run_stardis."""
    try:
        from astropy import units as u
        from stardis.base import STARDISOutput, set_num_threads
        from stardis.io.base import parse_config_to_model
        from stardis.plasma import create_stellar_plasma
    except ImportError as exc:  # pragma: no cover - depends on the user's environment
        raise ImportError(
            "run_stardis needs the reference package for configuration, model I/O and the TARDIS plasma "
            "(pip install stardis); stardis_amd replaces only the radiation-field stage"
        ) from exc

    tracing_nus = tracing_lambdas_or_nus.to(u.Hz, u.spectral())  # wavelengths ascending -> frequencies descending
    config, adata, stellar_model = parse_config_to_model(config_fname, add_config_dict)
    set_num_threads(config.n_threads)  # still governs the plasma stage
    stellar_plasma = create_stellar_plasma(stellar_model, adata, config)
    stellar_radiation_field = create_stellar_radiation_field(tracing_nus, stellar_model, stellar_plasma, config)
    return STARDISOutput(config.result_options, stellar_model, stellar_plasma, stellar_radiation_field)


def patch_stardis():
    """Make an installed `stardis` use the GPU radiation field everywhere: stardis.base.run_stardis looks the function
    up in its own module namespace (stardis/base.py:5,39)."""
    import stardis.base as ref

    ref.create_stellar_radiation_field = create_stellar_radiation_field
    return ref
