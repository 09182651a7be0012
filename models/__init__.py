"""Import path of the reference (``models.core.*``) on top of ``ppmstereo_amd``: ``models/ppm_stereo_model.py:12`` of the reference does
``from models.core.ppmstereo import PPMStereo`` -- with this package on ``sys.path`` instead of the reference's, that line (and the
imports of ``models/core/ppmstereo.py:17-33``) resolve to the gfx950 implementation unchanged.  See INTEGRATION.md."""
