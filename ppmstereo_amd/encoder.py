"""fnet on MI355X: drop-in for the reference's ``BasicEncoder(output_dim=256, norm_fn="instance")``
(/root/reference/models/core/extractor.py:302-423; built by PPMStereo at ppmstereo.py:64, called at :618) -- SURVEY.md section 8 row f3,
the producer of fmap1 / fmap2, the step before the hot path.

Same constructor arguments, same ``state_dict`` keys / shapes / order (``tools/gen_golden.py`` loads this repo's weights into the reference
module with ``strict=True``), same call: ``fmap1, fmap2 = fnet([image1, image2])`` with (N, 3, H, W) images normalised to [-1, 1].

How it runs: every convolution is an implicit-GEMM launch of libppms (bf16x3 split MFMA, fp32 accumulate) on channel-last
split-bf16 activations; the two stride-2 stages (conv1 7x7 s2; layer2.0's conv1 3x3 s2 and its 1x1 s2 skip) are stride-1 convolutions
over a 2x2 space-to-depth copy of their input with the weights re-laid to match (``_s2d_weight``); InstanceNorm2d(affine=False) is
two small kernels (per (sample, channel) statistics; normalise [+ residual] [+ ReLU] -> the next conv's input).  No torch op touches
the data between the image and the NCHW feature maps; buffers and descriptors are built once per (N, H, W).
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn as nn

from . import _lib as L
from . import packing as _packing
from .engine import TUNING, ConvOp, epilogue


def _s2d_weight(w: torch.Tensor, pad: int) -> torch.Tensor:
    """A stride-2 conv (cout, cin, k, k) with padding `pad` as a stride-1 'same' conv over the 2x2 space-to-depth input:
    channel (2 dy + dx) * cin + c of pixel (i, j) holds x[c, 2 i + dy, 2 j + dx], so tap offset d = ky - pad = 2 a + dy lands on
    kernel index a of phase dy.  The a range is made symmetric (odd kernel) with zero taps."""
    cout, cin, k, _ = w.shape
    offs = [divmod(ky - pad, 2) for ky in range(k)]                 # (a, dy) per original tap
    amax = max(abs(a) for a, _ in offs)
    ks = 2 * amax + 1
    out = torch.zeros(cout, 4 * cin, ks, ks, dtype=w.dtype, device=w.device)
    for ky, (a, dy) in enumerate(offs):
        for kx, (b, dx) in enumerate(offs):
            ph = 2 * dy + dx
            out[:, ph * cin:(ph + 1) * cin, a + amax, b + amax] = w[:, :, ky, kx]
    return out


class _Holder(nn.Module):
    """ResidualBlock's parameter tree (conv1, conv2, downsample.0), extractor.py:303-341."""

    def __init__(self, cin: int, planes: int, stride: int):
        super().__init__()
        self.conv1 = nn.Conv2d(cin, planes, 3, padding=1, stride=stride)
        self.conv2 = nn.Conv2d(planes, planes, 3, padding=1)
        self.downsample = nn.Sequential(nn.Conv2d(cin, planes, 1, stride=stride))
        self.stride = stride


class BasicEncoder(nn.Module):
    """The reference's fnet.  Only the configuration PPMStereo uses is implemented: norm_fn="instance", dropout=0."""

    def __init__(self, output_dim: int = 256, norm_fn: str = "instance", dropout: float = 0.0):
        super().__init__()
        if norm_fn != "instance" or dropout != 0.0:
            raise NotImplementedError("ppmstereo_amd BasicEncoder: norm_fn='instance', dropout=0 only (ppmstereo.py:64)")
        self.norm_fn = norm_fn
        self.conv1 = nn.Conv2d(3, 64, kernel_size=7, stride=2, padding=3)
        cin = 64
        for layer, (dim, stride) in enumerate(((64, 1), (96, 2), (128, 1)), start=1):
            setattr(self, f"layer{layer}", nn.Sequential(_Holder(cin, dim, stride), _Holder(dim, dim, 1)))
            cin = dim
        self.conv2 = nn.Conv2d(128, output_dim, kernel_size=1)
        self.output_dim = output_dim
        self._engines: "OrderedDict[tuple, _FnetEngine]" = OrderedDict()
        self._packed: Optional[Dict[str, tuple]] = None

    # ------------------------------------------------------------------------------------------------ weights
    def load_state_dict(self, sd, strict: bool = True, **kw):
        r = super().load_state_dict(sd, strict=strict, **kw)
        self.invalidate()
        return r

    def invalidate(self) -> None:
        """Forget the packed weight copies and launch plans (call after changing parameters in place)."""
        self._packed = None
        self._engines.clear()

    def _pack(self, device) -> Dict[str, tuple]:
        if self._packed is not None:
            return self._packed
        pk: Dict[str, tuple] = {}

        def put(name, w, b, cin_real, cin_pad):
            packed, bias, meta = _packing.pack_conv2(w.detach().to(device), b.detach().to(device), [cin_real], [cin_pad])
            pk[name] = (packed, bias, meta, tuple(w.shape[2:]))
            kh, kw = w.shape[2:]
            if kh == 3 and kw == 3 and cin_pad % 32 == 0 and 64 < w.shape[0] <= 128:
                # the 3x3 layers with 96 / 128 couts (the 1/4-resolution stages): conv_gemm6 (conv_gemm6.hip; pack_conv6 with (ky, kx) flattened into
                # the sweep axis, as engine.PackedBlock packs the update block's 3x3 convs) where the library rates its tile fill
                w5 = w.detach().to(device)[:, :, None]
                sweep = w5.reshape(w5.shape[0], w5.shape[1], 1, 1, kh * kw).contiguous()
                pk[name + "@6"] = _packing.pack_conv6(sweep, b.detach().to(device), [cin_real], [cin_pad], None, 128)

        put("conv1", _s2d_weight(self.conv1.weight, 3), self.conv1.bias, 12, 32)
        for layer in (1, 2, 3):
            for bi, blk in enumerate(getattr(self, f"layer{layer}")):
                pre = f"layer{layer}.{bi}."
                cin, cpad = blk.conv1.in_channels, blk.conv1.in_channels
                if blk.stride == 2:
                    put(pre + "conv1", _s2d_weight(blk.conv1.weight, 1), blk.conv1.bias, 4 * cin, 4 * cpad)
                    put(pre + "down", _s2d_weight(blk.downsample[0].weight, 0), blk.downsample[0].bias, 4 * cin, 4 * cpad)
                else:
                    put(pre + "conv1", blk.conv1.weight, blk.conv1.bias, cin, cpad)
                    put(pre + "down", blk.downsample[0].weight, blk.downsample[0].bias, cin, cpad)
                put(pre + "conv2", blk.conv2.weight, blk.conv2.bias, blk.conv2.in_channels, blk.conv2.in_channels)
        put("conv2", self.conv2.weight, self.conv2.bias, 128, 128)
        self._packed = pk
        return pk

    # ------------------------------------------------------------------------------------------------ forward
    def forward(self, x):
        """x: (N, 3, H, W) fp32 on the GPU, or a pair of such (extractor.py:398-401: concatenated on the batch axis, split back at
        :419-420).  Returns (N, output_dim, H/4, W/4) fp32 (a pair for a pair)."""
        is_list = isinstance(x, (tuple, list))
        xs = list(x) if is_list else [x]
        for t in xs:
            if not (torch.is_tensor(t) and t.is_cuda and t.dtype == torch.float32 and t.dim() == 4 and t.shape[1] == 3):
                raise RuntimeError("ppmstereo_amd BasicEncoder: (N, 3, H, W) fp32 tensors on the MI355X expected (no CPU path)")
        img = torch.cat(xs, dim=0).contiguous() if is_list else xs[0].contiguous()
        N, _, H, W = img.shape
        if H % 4 or W % 4:
            raise ValueError(f"ppmstereo_amd BasicEncoder: H = {H}, W = {W} must be multiples of 4 (the reference pads inputs to multiples of 32, ppmstereo.py:251)")
        key = (N, H, W, img.device.index)
        eng = self._engines.get(key)
        if eng is None:
            with torch.cuda.device(img.device):
                eng = _FnetEngine(self._pack(img.device), N, H, W, img.device, self.output_dim)
            self._engines[key] = eng
            while len(self._engines) > 2:
                self._engines.popitem(last=False)
        with torch.cuda.device(img.device):                      # launches go to the current stream OF THE TENSOR'S device
            out = eng.run(img)
        if is_list:
            return torch.split(out, out.shape[0] // 2, dim=0)
        return out


class _FnetEngine:
    """Buffers + launch list of one (N, H, W): built once, replayed per call."""

    EPS = 1e-5

    def __init__(self, pk: Dict[str, tuple], N: int, H: int, W: int, device, output_dim: int):
        self.lib = L.load()
        self.N, self.H, self.W, self.device, self.odim = N, H, W, device, output_dim
        H2, W2, H4, W4 = H // 2, W // 2, H // 4, W // 4
        P2, P4 = N * H2 * W2, N * H4 * W4
        self.P2, self.P4 = P2, P4
        keep: List = []                 # EVERY buffer of the plan: the launch descriptors hold raw pointers only

        def sp(p, c):
            t = L.SPTensor(p, c, device)
            keep.append(t)
            return t

        def f32(p, c):
            t = torch.empty(p, c, device=device, dtype=torch.float32)
            keep.append(t)
            return t

        self.stats = torch.empty(N * 256 * 2, device=device, dtype=torch.float32)
        ws_bytes = max(int(self.lib.ppms_instnorm_workspace_bytes(N, hw, c)) for hw, c in ((H2 * W2, 64), (H4 * W4, 96), (H4 * W4, 128)))
        self.ws = torch.empty(ws_bytes, device=device, dtype=torch.uint8)
        self.ops: List[tuple] = []                                   # ("conv", ConvOp) | ("call", fn)

        def conv(name, src: L.SPTensor, dst_f32: torch.Tensor, n, h, w):
            packed, bias, meta, k2 = pk[name]
            d = L.Conv()
            d.seg[0] = src.view()
            d.nseg, d.w, d.bias = 1, packed.data_ptr(), bias.data_ptr()
            d.T, d.H, d.W = n, h, w
            d.kt, d.kh, d.kw = 1, k2[0], k2[1]
            d.M = d.m_split = meta["M"]
            assert src.channels == meta["cpad"] and dst_f32.shape[1] == meta["M"], (name, src.channels, meta["cpad"], dst_f32.shape, meta["M"])
            d.epi[0] = epilogue(n_valid=meta["M"], out_f32=dst_f32, out_f32_ld=meta["M"])
            if TUNING["conv6"] and name + "@6" in pk and meta["M"] == 128:      # (TUNING["conv6"] = False: every conv of the encoder on conv_gemm2, as the engine's fallback)
                packed6, bias6, meta6 = pk[name + "@6"]
                d6 = L.Conv.from_buffer_copy(bytes(d))
                d6.w, d6.bias = packed6.data_ptr(), bias6.data_ptr()
                if meta6["M"] == 128 and self.lib.ppms_conv_gemm6_applicable(C.byref(d6)) == 1:
                    self.ops.append(("conv", ConvOp(d6, [src, dst_f32, packed6, bias6], 8, device=device)))
                    return
            self.ops.append(("conv", ConvOp(d, [src, dst_f32, packed, bias], 2, device=device)))

        def norm(src_f32: torch.Tensor, C_, hw, dst: L.SPTensor, relu: bool, res: Optional[L.SPTensor] = None):
            ld = src_f32.shape[1]
            rv = res.view() if res is not None else L.SP(None, None, 0, 0)
            dv = dst.view()

            def fn():
                L.check(self.lib.ppms_instnorm_stats(src_f32.data_ptr(), ld, N, hw, C_, self.EPS, self.stats.data_ptr(), self.ws.data_ptr(), L.stream_ptr()))
                L.check(self.lib.ppms_instnorm_apply(src_f32.data_ptr(), ld, self.stats.data_ptr(), rv, int(relu), dv, N, hw, C_, L.stream_ptr()))
            keep.extend([src_f32, dst, res])
            self.ops.append(("call", fn))

        # ---- stem: conv1 7x7 s2 (as 5x5 over the space-to-depth image) + IN + ReLU ------------------------------------------------
        self.s0 = sp(P2, 32)
        a = sp(P2, 64)
        t64 = f32(P2, 64)
        conv("conv1", self.s0, t64, N, H2, W2)
        norm(t64, 64, H2 * W2, a, True)

        def block(pre, x: L.SPTensor, cin, planes, stride, n, h, w):
            """ResidualBlock (extractor.py:337-345) at the OUTPUT resolution h x w; x is at the input resolution."""
            hw = h * w
            P_ = n * hw
            M = ((planes + 63) // 64) * 64
            if stride == 2:
                xin = sp(P_, 4 * cin)
                xv, dv = x.view(), xin.view()
                self.ops.append(("call", lambda: L.check(self.lib.ppms_sp_s2d(xv, dv, n, 2 * h, 2 * w, L.stream_ptr()))))
                keep.extend([x, xin])
            else:
                xin = x
            t1, t2 = f32(P_, M), f32(P_, M)
            y1, y2, out = sp(P_, planes), sp(P_, planes), sp(P_, planes)
            conv(pre + "conv1", xin, t1, n, h, w)
            norm(t1, planes, hw, y1, True)
            conv(pre + "conv2", y1, t2, n, h, w)
            norm(t2, planes, hw, y2, True)
            conv(pre + "down", xin, t1, n, h, w)                      # (t1 is free again: y1 was produced from it)
            norm(t1, planes, hw, out, True, res=y2)                  # relu(IN(down(x)) + y)
            return out

        x = block("layer1.0.", a, 64, 64, 1, N, H2, W2)
        x = block("layer1.1.", x, 64, 64, 1, N, H2, W2)
        x = block("layer2.0.", x, 64, 96, 2, N, H4, W4)
        x = block("layer2.1.", x, 96, 96, 1, N, H4, W4)
        x = block("layer3.0.", x, 96, 128, 1, N, H4, W4)
        x = block("layer3.1.", x, 128, 128, 1, N, H4, W4)
        self.tout = f32(P4, output_dim)
        conv("conv2", x, self.tout, N, H4, W4)
        self.out = torch.empty(N, output_dim, H4, W4, device=device, dtype=torch.float32)
        self.keep = keep

    def run(self, img: torch.Tensor) -> torch.Tensor:
        L.check(self.lib.ppms_img_s2d(img.data_ptr(), self.s0.view(), self.N, 3, self.H, self.W, 2, L.stream_ptr()))
        for kind, op in self.ops:
            op()
        out = torch.empty_like(self.out)
        L.check(self.lib.ppms_nhwc_to_nchw(self.tout.data_ptr(), self.odim, out.data_ptr(), self.N, self.odim, (self.H // 4) * (self.W // 4), L.stream_ptr()))
        return out
