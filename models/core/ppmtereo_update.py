"""``models.core.ppmtereo_update`` of the reference, the names ``models/core/ppmstereo.py:17-24`` imports that live on the hot path
(ppmtereo_update.py:25-88, 118-133, 880-1003), executing on the gfx950 kernels.  The 2-D ``SequenceUpdateBlock`` is not on the
path of the shipped configuration (``use_3d_update_block=True``, models/ppm_stereo_model.py:27-33)."""
from ppmstereo_amd.update import Attention_qk, SequenceUpdateBlock3D, get_temporal_positional_encoding  # noqa: F401

__all__ = ["Attention_qk", "SequenceUpdateBlock3D", "get_temporal_positional_encoding"]
