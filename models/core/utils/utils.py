"""``models.core.utils.utils`` of the reference (utils.py:10-44): the padder and the align_corners=True resize."""
from ppmstereo_amd.ppmstereo import InputPadder, interp  # noqa: F401

__all__ = ["InputPadder", "interp"]
