"""A/B switches for measurements on the GPU box: maps PPMS_* environment variables onto ppmstereo_amd.engine.TUNING (the product reads
no environment variable itself).  Import this module before the first engine is built:

    PPMS_CONV5=0 PPMS_SLICE=0 python -c "import tools.ab_switches; ..."      or      import ab_switches  (from tools/)

    PPMS_CONV6, PPMS_CONV5, PPMS_CONV5_SLICED, PPMS_CONV5_GEMM, PPMS_CONV6_GROUPED, PPMS_PWCHAIN, PPMS_SLICE, PPMS_HOIST, PPMS_CONV5_PAD2X: 0 / 1
    PPMS_HID: 0 / 1 (GRU convs of the hoisted blocks on [h | mf, hid] with folded weights, products with hid's zero lo plane skipped)
    PPMS_STREAM: 0 / 1 / all (conv_stream.hip: off / where the library rates it faster / wherever it serves a small map); PPMS_STREAM_HINT: 0 / 1 / 2
    PPMS_ATTN_P: fp16 / bf16 (ppms_mem_attn's p_format)
    PPMS_YSWEEP: 0 = off, 1 = y-swept (1, kh, 1) convs, 2d = also the 2-D window for kh, kw > 1
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppmstereo_amd import engine as _engine  # noqa: E402

_MAP = dict(PPMS_CONV6="conv6", PPMS_CONV6_STREAM="conv6_stream", PPMS_CONV6_PAD2X="conv6_pad2x", PPMS_CONV5="conv5", PPMS_CONV5_SLICED="conv5_sliced", PPMS_CONV5_GEMM="conv5_gemm", PPMS_CONV6_GROUPED="conv6_grouped", PPMS_PWCHAIN="pwchain",
            PPMS_SLICE="slices", PPMS_HOIST="hoist", PPMS_CONV5_PAD2X="conv5_pad2x", PPMS_CONVF2_UNSLICED="convf2_unsliced", PPMS_GEMM1="gemm1", PPMS_CONV5_M192="conv5_m192", PPMS_HID="hid_exact")
for _env, _key in _MAP.items():
    if _env in os.environ:
        _engine.TUNING[_key] = os.environ[_env] != "0"
if "PPMS_STREAM" in os.environ:
    _engine.TUNING["stream"] = {"0": False, "1": True}.get(os.environ["PPMS_STREAM"], os.environ["PPMS_STREAM"])
if "PPMS_CONV6_ONLY" in os.environ:          # debugging: conv_gemm6 only for these weight names (comma separated)
    _engine.TUNING["conv6_only"] = set(os.environ["PPMS_CONV6_ONLY"].split(","))
if "PPMS_STREAM_HINT" in os.environ:
    _engine.TUNING["stream_hint"] = int(os.environ["PPMS_STREAM_HINT"])
if "PPMS_FORK_MIN" in os.environ:
    _engine.TUNING["fork_min_pixels"] = int(os.environ["PPMS_FORK_MIN"])
if "PPMS_YSWEEP" in os.environ:
    _engine.TUNING["ysweep"] = os.environ["PPMS_YSWEEP"] != "0"
    _engine.TUNING["win2d"] = os.environ["PPMS_YSWEEP"] == "2d"
if "PPMS_ATTN_P" in os.environ:               # fp16 / bf16: format of P~ in the memory read-out's P~ V product
    _engine.TUNING["attn_p"] = os.environ["PPMS_ATTN_P"]
if "PPMS_LIB" in os.environ:                 # A/B of two builds on ONE box: another libppms.so (same ABI) instead of the in-tree one
    from ppmstereo_amd import build as _build
    _build.LIB = os.path.abspath(os.environ["PPMS_LIB"])
    _build.build = lambda *a, **k: _build.LIB
print("[ab_switches]", _engine.TUNING, file=sys.stderr)
