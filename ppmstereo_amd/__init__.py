"""MI355X-native (gfx950) implementation of the PPMStereo per-frame hot path.

Host code is Python on PyTorch-ROCm; every kernel is hand-written HIP behind the C ABI declared in
``include/ppms.h`` (``ppmstereo_amd/csrc``).  Importing the package does not need a GPU; calling
any op without the built HIP library raises (there is no CPU fallback).
"""
__version__ = "0.1.0"
