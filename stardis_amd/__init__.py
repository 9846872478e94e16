"""stardis_amd — the STARDIS radiation-field hot path (opacity assembly + LTE formal solution) on MI355X.

Layout mirrors the reference package for the modules on the path:
    stardis_amd.radiation_field.{base, opacities, radiation_field_solvers, source_functions}
plus
    stardis_amd.engine      device-resident fused synthesis (benchmark / multi-GPU path)
    stardis_amd.parallel    frequency sharding over torch.distributed (RCCL)
    stardis_amd.ops         array-level front-end of the C ABI (include/stardis_hip.h)
"""
__version__ = "0.1.0"
