"""``models.core.ppmstereo`` of the reference: ``PPMStereo`` (ppmstereo.py:44-810) with the same constructor arguments, ``state_dict``
layout, ``forward`` and ``forward_batch_test`` -- what ``models/ppm_stereo_model.py:12,27-50`` imports, builds, loads and calls."""
from ppmstereo_amd.ppmstereo import PPMStereo, forward_update_block  # noqa: F401

__all__ = ["PPMStereo", "forward_update_block"]
