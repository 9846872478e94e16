"""GPU counterparts of the reference's line-strength plasma properties (stardis/plasma/base.py,
stardis/plasma/molecules.py): only the alpha_line calculators that feed the hot path."""
from stardis_amd.plasma.base import AlphaLine, AlphaLineShortlistVald, AlphaLineVald  # noqa: F401
from stardis_amd.plasma.molecules import AlphaLineShortlistValdMolecule, AlphaLineValdMolecule  # noqa: F401
