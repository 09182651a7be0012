"""ctypes binding of libppms.so (C ABI: include/ppms.h).  No CPU fallback: if the library cannot be
loaded (or built) every op raises RuntimeError."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import build as _build

c_void_p, c_int, c_float, c_int64 = C.c_void_p, C.c_int, C.c_float, C.c_int64

EPI_STORE, EPI_RESID, EPI_RH, EPI_GRU, EPI_ADDF32 = range(5)
ATTN_P_BF16, ATTN_P_FP16 = 0, 1
ACT_NONE, ACT_RELU, ACT_GELU, ACT_SIGMOID, ACT_TANH, ACT_ELU1 = range(6)


def vt_image(v: torch.Tensor, p_format: int) -> torch.Tensor:
    """The 16-bit storage of ppms_mem_attn's transposed value operand for `p_format` (include/ppms.h), as a bfloat16-typed tensor:
    bf16(v) itself, or -- ATTN_P_FP16 -- the fp16 numbers equal to those bf16 values (saturated at +-65504), bit-cast.  What the
    to_v convolution's epilogue writes with ppms_epilogue.vt_f16; callers that build V^T themselves (tests, other hosts) use this."""
    b = v.to(torch.bfloat16)
    if p_format == ATTN_P_BF16:
        return b.contiguous()
    return b.float().clamp(-65504.0, 65504.0).to(torch.float16).contiguous().view(torch.bfloat16)


def vt_values(vt: torch.Tensor, p_format: int) -> torch.Tensor:
    """fp32 values held by a V^T storage tensor (inverse of vt_image)."""
    return vt.float() if p_format == ATTN_P_BF16 else vt.view(torch.float16).float()


class SP(C.Structure):
    """ppms_sp: channel-last split-bf16 view."""
    _fields_ = [("hi", c_void_p), ("lo", c_void_p), ("ld", C.c_int32), ("c", C.c_int32)]


class Epilogue(C.Structure):
    _fields_ = [("kind", C.c_int32), ("act", C.c_int32), ("scale", c_float), ("n_valid", C.c_int32),
                ("out_sp", SP), ("out_f32", c_void_p), ("out_f32_ld", C.c_int32), ("vt_f16", C.c_int32), ("out_vt", c_void_p),
                ("aux_sp", SP), ("aux_f32", c_void_p), ("aux_f32_ld", C.c_int32), ("pre_f32_ld", C.c_int32),
                ("pre_f32", c_void_p)]


class Conv(C.Structure):
    _fields_ = [("seg", SP * 2), ("nseg", C.c_int32), ("groups", C.c_int32), ("w", c_void_p), ("bias", c_void_p),
                ("T", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("kt", C.c_int32), ("kh", C.c_int32), ("kw", C.c_int32),
                ("M", C.c_int32), ("m_split", C.c_int32), ("t_halo", C.c_int32), ("lo_zero_from", C.c_int32), ("epi", Epilogue * 2)]


class ChainLayer(C.Structure):
    _fields_ = [("w", c_void_p), ("bias", c_void_p), ("post_s", c_void_p), ("post_t", c_void_p),
                ("M", C.c_int32), ("n_valid", C.c_int32), ("resid", C.c_int32)]


class ChainParams(C.Structure):
    _fields_ = [("inp", SP), ("out", SP), ("layer", ChainLayer * 3), ("nlayers", C.c_int32), ("P", C.c_int64)]


_SIGS = {
    "ppms_version": (c_int, []),
    "ppms_last_error": (C.c_char_p, []),
    "ppms_device_info": (c_int, [C.c_char_p, c_int, C.POINTER(c_int), C.POINTER(c_int)]),
    "ppms_corr_build": (c_int, [c_void_p, c_void_p, C.POINTER(c_void_p), c_int, c_int, c_int, c_int, c_void_p]),
    "ppms_corr_lookup": (c_int, [C.POINTER(c_void_p), c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p, c_void_p,
                                 c_int, c_int, c_int, c_int, c_void_p]),
    "ppms_conv_gemm2": (c_int, [C.POINTER(Conv), c_void_p, c_int, c_void_p]),
    "ppms_conv_gemm2_slices": (c_int, [C.POINTER(Conv)]),
    "ppms_conv_gemm2_slice_workspace_bytes": (C.c_int64, [C.POINTER(Conv), c_int]),
    "ppms_conv_gemm2_sliced": (c_int, [C.POINTER(Conv), c_void_p, c_int, c_void_p, c_void_p]),
    "ppms_conv_gemm2_ysweep_slices": (c_int, [C.POINTER(Conv)]),
    "ppms_conv_gemm2_ysweep": (c_int, [C.POINTER(Conv), c_void_p, c_int, c_void_p, c_void_p]),
    "ppms_gemm1_applicable": (c_int, [C.POINTER(Conv)]),
    "ppms_gemm1": (c_int, [C.POINTER(Conv), c_void_p, c_int, c_void_p]),
    "ppms_conv_stream_applicable": (c_int, [C.POINTER(Conv)]),
    "ppms_conv_stream": (c_int, [C.POINTER(Conv), c_void_p, c_int, c_void_p]),
    "ppms_conv_gemm5_applicable": (c_int, [C.POINTER(Conv)]),
    "ppms_conv_gemm5": (c_int, [C.POINTER(Conv), c_void_p, c_int, c_void_p]),
    "ppms_conv_gemm6_applicable": (c_int, [C.POINTER(Conv)]),
    "ppms_conv_gemm6": (c_int, [C.POINTER(Conv), c_void_p, c_void_p]),
    "ppms_conv_gemm5_slices": (c_int, [C.POINTER(Conv)]),
    "ppms_conv_gemm5_sliced": (c_int, [C.POINTER(Conv), c_void_p, c_int, c_int, c_void_p, c_void_p]),
    "ppms_struct_sizes": (c_int, [C.POINTER(c_int), C.POINTER(c_int), C.POINTER(c_int)]),
    "ppms_dwconv_gelu": (c_int, [SP, SP, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "ppms_flow_patch7": (c_int, [c_void_p, SP, c_int, c_int, c_int, c_void_p]),
    "ppms_unc_tail": (c_int, [SP, c_void_p, c_float, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "ppms_nchw_to_sp": (c_int, [c_void_p, SP, c_int, c_int, c_int, c_void_p]),
    "ppms_sp_to_nchw": (c_int, [SP, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ppms_nchw_to_nhwc": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "ppms_nhwc_to_nchw": (c_int, [c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ppms_f32_to_sp": (c_int, [c_void_p, c_int, SP, c_int64, c_void_p]),
    "ppms_sp_to_f32": (c_int, [SP, c_void_p, c_int, c_int64, c_void_p]),
    "ppms_tap_gather_sum": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ppms_flow_add": (c_int, [c_void_p, c_void_p, c_int, c_int64, c_void_p]),
    "ppms_convex_upsample": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ppms_convex_upsample_3d": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "ppms_bilinear": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_int, c_int, c_int, c_float, c_void_p]),
    "ppms_sp_resize_blend": (c_int, [SP, SP, c_int, c_int, c_int, c_int, c_int, c_float, c_float, c_void_p]),
    "ppms_avgpool": (c_int, [c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "ppms_axpby": (c_int, [c_void_p, c_void_p, c_void_p, c_float, c_float, c_int64, c_int64, c_void_p]),
    "ppms_ctx_mix": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "ppms_img_s2d": (c_int, [c_void_p, SP, c_int, c_int, c_int, c_int, c_int, c_void_p]),
    "ppms_dwconv": (c_int, [SP, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_int, c_void_p]),
    "ppms_layernorm_any": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_float, SP, c_int64, c_int, c_void_p]),
    "ppms_grn_workspace_bytes": (c_int64, [c_int, c_int, c_int]),
    "ppms_grn": (c_int, [c_void_p, c_int, c_void_p, c_void_p, SP, c_int, c_int, c_int, c_void_p, c_void_p]),
    "ppms_sp_upsample2": (c_int, [SP, SP, c_int, c_int, c_int, c_void_p]),
    "ppms_sp_s2d": (c_int, [SP, SP, c_int, c_int, c_int, c_void_p]),
    "ppms_instnorm_workspace_bytes": (c_int64, [c_int, c_int, c_int]),
    "ppms_instnorm_stats": (c_int, [c_void_p, c_int, c_int, c_int, c_int, C.c_float, c_void_p, c_void_p, c_void_p]),
    "ppms_instnorm_apply": (c_int, [c_void_p, c_int, c_void_p, SP, c_int, SP, c_int, c_int, c_int, c_void_p]),
    "ppms_qk_similarity": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ppms_qk_pool": (c_int, [c_void_p, c_void_p, c_int, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ppms_qk_cos": (c_int, [c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "ppms_qam_select": (c_int, [c_void_p, c_void_p, c_void_p, c_int, c_int, c_void_p, c_void_p, c_void_p, c_int, c_void_p]),
    "ppms_attn_prep_q": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_int, c_int, c_void_p]),
    "ppms_attn_prep_k": (c_int, [c_void_p, c_int, c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_int, c_int, c_void_p]),
    "ppms_pwchain": (c_int, [c_void_p, c_int64, c_void_p]),
    "ppms_pwchain_param_bytes": (c_int, []),
    "ppms_time_attn": (c_int, [SP, c_void_p, c_void_p, SP, c_int, c_int, c_int, c_void_p]),
    "ppms_layernorm": (c_int, [c_void_p, c_int, c_void_p, c_void_p, SP, SP, c_int64, c_int, c_void_p]),
    "ppms_linear_attention_workspace_floats": (c_int64, [c_int, c_int, c_int, c_int]),
    "ppms_linear_attention": (c_int, [c_void_p, c_int, c_void_p, c_int, c_void_p, c_int, c_void_p, SP, c_int, c_int, c_int, c_int, c_void_p]),
    "ppms_mem_attn": (c_int, [c_void_p, c_void_p, c_void_p, c_void_p, c_int, c_float, c_void_p, SP, SP, c_void_p, c_int, c_int, c_void_p, c_int, c_int, c_void_p]),
    "ppms_mem_attn_workspace_bytes": (c_int64, [c_int, c_int, c_int]),
}
EXPORTS = tuple(_SIGS)

_lib: Optional[C.CDLL] = None


def lib_path() -> str:
    return _build.LIB


def load() -> C.CDLL:
    """Load (building first if the .so is absent or stale).  Raises RuntimeError when that is impossible."""
    global _lib
    if _lib is not None:
        return _lib
    try:
        path = _build.build(verbose=not os.path.exists(_build.LIB))
        lib = C.CDLL(path)
    except Exception as exc:  # noqa: BLE001
        raise RuntimeError(f"ppmstereo_amd: the HIP library libppms.so is missing and could not be built ({exc}); "
                           "there is no CPU fallback") from exc
    for name, (res, args) in _SIGS.items():
        fn = getattr(lib, name)
        fn.restype, fn.argtypes = res, args
    if lib.ppms_version() != 4:
        raise RuntimeError("ppmstereo_amd: libppms.so ABI version mismatch")
    a, b, c = c_int(), c_int(), c_int()
    lib.ppms_struct_sizes(C.byref(a), C.byref(b), C.byref(c))
    if (a.value, b.value, c.value) != (C.sizeof(SP), C.sizeof(Epilogue), C.sizeof(Conv)):
        raise RuntimeError("ppmstereo_amd: ctypes struct layout differs from include/ppms.h")
    if lib.ppms_pwchain_param_bytes() != C.sizeof(ChainParams):
        raise RuntimeError("ppmstereo_amd: ChainParams layout differs from pwchain.hip")
    _lib = lib
    return lib


def check(rc: int) -> None:
    if rc != 0:
        raise RuntimeError("libppms: " + load().ppms_last_error().decode())


def stream_ptr() -> int:
    """hipStream_t of torch's current stream (kernels are enqueued there; nothing synchronises)."""
    return torch.cuda.current_stream().cuda_stream


def require_gpu(*tensors: torch.Tensor) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("ppmstereo_amd ops run on the GPU only (got a CPU tensor); there is no CPU fallback")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    return None if t is None else t.data_ptr()


class SPTensor:
    """Owner of a split-bf16 channel-last activation: data (2, before + P + after, ld) bf16; views select a channel range of
    the P "own" pixels.  before / after: extra pixel rows around them -- the temporal halo slabs of a frame-sharded window
    (ppmstereo_amd/dist.py) or, for gathered tensors, the other ranks' frames."""

    def __init__(self, pixels: int, channels: int, device, zero: bool = True, before: int = 0, after: int = 0):
        alloc = torch.zeros if zero else torch.empty
        self.data = alloc((2, before + pixels + after, channels), dtype=torch.bfloat16, device=device)
        self.pixels, self.channels, self.before, self.after = pixels, channels, before, after

    def view(self, c0: int = 0, c: Optional[int] = None, all_rows: bool = False) -> SP:
        """SP view of channels [c0, c0 + c) starting at the first own pixel (all_rows: at the first allocated row)."""
        c = self.channels - c0 if c is None else c
        assert 0 <= c0 and c0 + c <= self.channels
        total = self.before + self.pixels + self.after
        base = self.data.data_ptr() + (0 if all_rows else self.before) * self.channels * 2
        plane = total * self.channels * 2
        return SP(base + c0 * 2, base + plane + c0 * 2, self.channels, c)

    def own(self) -> torch.Tensor:
        """(2, P, ld) view of the own pixels."""
        return self.data[:, self.before:self.before + self.pixels]

    def to_f32(self, c0: int = 0, c: Optional[int] = None) -> torch.Tensor:
        """(P, c) fp32 = hi + lo of the own pixels (torch ops; for tests and glue)."""
        c = self.channels - c0 if c is None else c
        d = self.own()
        return d[0, :, c0:c0 + c].float() + d[1, :, c0:c0 + c].float()

    def set_f32(self, x: torch.Tensor, c0: int = 0) -> None:
        hi = x.to(torch.bfloat16)
        lo = (x - hi.float()).to(torch.bfloat16)
        d = self.own()
        d[0, :, c0:c0 + x.shape[1]] = hi
        d[1, :, c0:c0 + x.shape[1]] = lo
