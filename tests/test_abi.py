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
    assert lib.ppms_version() == 2
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
