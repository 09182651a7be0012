"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/ppms.h declares.  No compute."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    src = open(os.path.join(ROOT, "include", "ppms.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(ppms_[a-z0-9_]+)\s*\(", src)))


def test_library_builds_and_exports_the_declared_abi():
    from ppmstereo_amd import _lib as L
    lib = L.load()
    names = declared_symbols()
    assert len(names) >= 24
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/ppms.h but not exported by libppms.so"
    assert set(L.EXPORTS) == set(names), set(L.EXPORTS) ^ set(names)
    assert lib.ppms_version() == 4
    assert os.path.dirname(L.lib_path()) == os.path.join(ROOT, "ppmstereo_amd"), "the .so must live in-tree"


def test_struct_layout_matches_header():
    from ppmstereo_amd import _lib as L
    lib = L.load()
    a, b, c = ctypes.c_int(), ctypes.c_int(), ctypes.c_int()
    assert lib.ppms_struct_sizes(ctypes.byref(a), ctypes.byref(b), ctypes.byref(c)) == 0
    assert (a.value, b.value, c.value) == (ctypes.sizeof(L.SP), ctypes.sizeof(L.Epilogue), ctypes.sizeof(L.Conv)) == (24, 112, 336)


def test_argument_errors_are_reported_not_raised_in_c():
    """Contract violations return PPMS_EINVAL with a message (no exceptions across the ABI, no GPU touched)."""
    from ppmstereo_amd import _lib as L
    lib = L.load()
    ptrs = (ctypes.c_void_p * 5)()
    assert lib.ppms_corr_build(None, None, ptrs, 1, 256, 4, 8, None) == -1           # W < 16: pyramid impossible
    assert b"too small" in lib.ppms_last_error()
    assert lib.ppms_conv_gemm2(None, None, 0, None) == -1
    with pytest.raises(RuntimeError):
        L.check(lib.ppms_bilinear(None, None, 1, 1, 1, 1, 1, 1, 0, 1.0, None))


def test_product_package_never_imports_the_oracle():
    """The oracle is test infrastructure: nothing under ppmstereo_amd/ may import it."""
    pkg = os.path.join(ROOT, "ppmstereo_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            text = open(os.path.join(pkg, f)).read()
            assert "oracle" not in re.sub(r'""".*?"""', "", text, flags=re.S).replace("# oracle", ""), f


def test_reference_import_path_and_signatures():
    """``models.core.*`` (the reference's import path, models/ppm_stereo_model.py:12, models/core/ppmstereo.py:17-33) resolves to this
    package without a GPU, with the reference's parameter names (ppmstereo.py:45-55, 238, 426-441, 601; corr.py:56; ppmtereo_update.py:881)."""
    import inspect

    from models.core.corr import CorrBlock1D
    from models.core.ppmstereo import PPMStereo
    from models.core.ppmtereo_update import SequenceUpdateBlock3D
    names = lambda f: list(inspect.signature(f).parameters)
    assert names(PPMStereo.__init__)[1:9] == ["max_disp", "mixed_precision", "num_frames", "attention_type", "use_3d_update_block",
                                             "different_update_blocks", "use_convex_3d", "init_flow"]
    assert names(PPMStereo.forward)[1:6] == ["image1", "image2", "flow_init", "iters", "test_mode"]
    assert names(PPMStereo.forward_batch_test)[1:4] == ["batch_dict", "kernel_size", "iters"]
    assert names(PPMStereo.forward_update_block)[1:] == ["image1", "update_block", "corr_fn", "flow", "net", "inp", "motion_hidden_state", "attn_block",
                                                         "predictions", "uncertainties", "iters", "interp_scale", "t"]
    assert names(CorrBlock1D.__init__)[1:] == ["fmap1", "fmap2", "num_levels", "radius"]
    assert names(SequenceUpdateBlock3D.__init__)[1:] == ["hidden_dim", "cor_planes", "mask_size", "use_convex_3d", "attention_type"]
    assert names(SequenceUpdateBlock3D.forward)[1:] == ["net", "inp", "motion_features", "motion_features_global", "t"]
