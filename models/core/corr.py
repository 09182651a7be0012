"""``models.core.corr`` of the reference (corr.py:47-104), hot-path part: the gfx950 ``CorrBlock1D`` (fp32-MFMA pyramid build,
multi-level lookup) and ``coords_grid``.  ``TFCL`` / ``AAPC`` belong to other model families (BiDAStereo, StereoAnyVideo)."""
from ppmstereo_amd.corr import CorrBlock1D, coords_grid  # noqa: F401

__all__ = ["CorrBlock1D", "coords_grid"]
