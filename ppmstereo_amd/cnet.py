"""cnet on MI355X: drop-in for the reference's ``Feature("tiny", 256)`` -- frozen ConvNeXt-V2-tiny backbone + FPN decoder
(/root/reference/models/core/convnext.py:50-264; built by PPMStereo at ppmstereo.py:69, called at :624) -- SURVEY.md section 8 row f5.

Same ``state_dict`` keys / shapes / order as the reference module (``tools/gen_golden.py`` loads this repo's weights into it with
``strict=True``; the backbone's unused classifier head and final norm are kept so that the ConvNeXt checkpoint the reference loads at
:221-222 loads here too), same call: ``c4, c8, c16 = cnet(image1)`` with (N, 3, H, W) images normalised to [-1, 1], H, W multiples of 32.
Unlike the reference's constructor this one reads no checkpoint file: weights come through ``load_state_dict``.

How it runs: channel-last split-bf16 activations end to end.  Linear layers and convolutions are implicit-GEMM launches of libppms
(fp32-accurate bf16x3 MFMA): the patchify stem (4x4 s4) and the 2x2 s2 downsamplers as 1x1 convolutions over space-to-depth copies, the
1x1 / 3x3 decoder convolutions directly, the skip concatenations as two input segments of one launch.  Depthwise 7x7, LayerNorm, GRN,
nearest upsampling and InstanceNorm are small kernels of encoder_ops.hip.  108 block launches x 6 + decoder: built once per (N, H, W).
"""
from __future__ import annotations

from collections import OrderedDict
import ctypes as C
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import _lib as L
from . import packing as _packing
from .engine import ConvOp, TUNING, epilogue
from .weights import CNET_DEPTHS, CNET_DIMS


class _LN(nn.Module):                               # convnext.py:11-24 (parameter holder)
    def __init__(self, c):
        super().__init__()
        self.weight = nn.Parameter(torch.ones(c))
        self.bias = nn.Parameter(torch.zeros(c))


class _GRN(nn.Module):                              # convnext.py:37-43
    def __init__(self, c):
        super().__init__()
        self.gamma = nn.Parameter(torch.zeros(1, 1, 1, c))
        self.beta = nn.Parameter(torch.zeros(1, 1, 1, c))


class _Block(nn.Module):                            # convnext.py:57-65
    def __init__(self, dim):
        super().__init__()
        self.dwconv = nn.Conv2d(dim, dim, kernel_size=7, padding=3, groups=dim)
        self.norm = _LN(dim)
        self.pwconv1 = nn.Linear(dim, 4 * dim)
        self.grn = _GRN(4 * dim)
        self.pwconv2 = nn.Linear(4 * dim, dim)


class _ConvNeXtV2(nn.Module):                       # convnext.py:91-125, depths / dims of convnextv2_tiny (:161-163)
    def __init__(self):
        super().__init__()
        d = CNET_DIMS
        self.downsample_layers = nn.ModuleList([nn.Sequential(nn.Conv2d(3, d[0], kernel_size=4, stride=4), _LN(d[0]))])
        for i in range(3):
            self.downsample_layers.append(nn.Sequential(_LN(d[i]), nn.Conv2d(d[i], d[i + 1], kernel_size=2, stride=2)))
        self.stages = nn.ModuleList([nn.Sequential(*[_Block(d[i]) for _ in range(CNET_DEPTHS[i])]) for i in range(4)])
        self.norm = nn.LayerNorm(d[-1], eps=1e-6)   # unused by Feature.forward; present in the checkpoint
        self.head = nn.Linear(d[-1], 1000)


def _patch_weight(w: torch.Tensor) -> torch.Tensor:
    """A k x k conv with stride k (non-overlapping patches) as a 1x1 conv over the k x k space-to-depth input, channel (k dy + dx) * cin + c."""
    cout, cin, k, _ = w.shape
    return w.permute(0, 2, 3, 1).reshape(cout, k * k * cin)[:, :, None, None].contiguous()


class Feature(nn.Module):
    """The reference's cnet.  Only model_name="tiny" (what PPMStereo builds) is implemented."""

    def __init__(self, model_name: str = "tiny", output_dim: int = 256):
        super().__init__()
        if model_name != "tiny" or output_dim != 256:
            raise NotImplementedError("ppmstereo_amd Feature: model_name='tiny', output_dim=256 only (ppmstereo.py:69)")
        o, d = output_dim, CNET_DIMS
        self.convnext = _ConvNeXtV2()
        up = lambda cin: nn.Sequential(nn.Identity(), nn.Conv2d(cin, o, 3, 1, 1))                       # [0] = nn.Upsample, [1] = the conv
        dec = lambda cin: nn.Sequential(nn.Conv2d(cin + o, o, 1, 1, 0), nn.Identity(), nn.Identity(), nn.Conv2d(o, o, 3, 1, 1))
        self.upconv_16, self.upconv_8, self.upconv_4 = up(d[3]), up(o), up(o)
        self.decode_16x, self.decode_8x, self.decode_4x = dec(d[2]), dec(d[1]), dec(d[0])
        self.output_dim = o
        self._engines: "OrderedDict[tuple, _CnetEngine]" = OrderedDict()
        self._packed = None

    def load_state_dict(self, sd, strict: bool = True, **kw):
        r = super().load_state_dict(sd, strict=strict, **kw)
        self.invalidate()
        return r

    def invalidate(self) -> None:
        self._packed = None
        self._engines.clear()

    # ------------------------------------------------------------------------------------------------ weights
    def _pack(self, device):
        if self._packed is not None:
            return self._packed
        pk: Dict[str, tuple] = {}
        vec: Dict[str, torch.Tensor] = {}

        def put(name, w, b, segs, pads=None):
            w4 = w.detach().to(device)
            if w4.dim() == 2:
                w4 = w4[:, :, None, None]
            packed, bias, meta = _packing.pack_conv2(w4, b.detach().to(device), segs, pads if pads is not None else [((c + 31) // 32) * 32 for c in segs])
            pk[name] = (packed, bias, meta, tuple(w4.shape[2:]))
            if tuple(w4.shape[2:]) == (1, 1) and sum(meta["seg_padded"]) % 64 == 0:          # 1x1 layers the thin-GEMM kernel may serve (gemm1.hip)
                pk[name + "@1"] = _packing.pack_gemm1(w4, b.detach().to(device), segs, meta["seg_padded"], None, meta["M"])
            if TUNING["stream"] and sum(meta["seg_padded"]) % 64 == 0:      # layers the register-streamed small-map kernel may serve (conv_stream.hip; same switch as the loop's engine)
                pk[name + "@7"] = _packing.pack_stream(w4, b.detach().to(device), segs, meta["seg_padded"], None, meta["M"])

        def v(name, t):
            vec[name] = t.detach().float().reshape(-1).to(device).contiguous()

        cn, d = self.convnext, CNET_DIMS
        stem = cn.downsample_layers[0]
        put("stem", _patch_weight(stem[0].weight), stem[0].bias, [48], [64])
        v("stem.ln.w", stem[1].weight), v("stem.ln.b", stem[1].bias)
        for i in range(1, 4):
            ds = cn.downsample_layers[i]
            v(f"ds{i}.ln.w", ds[0].weight), v(f"ds{i}.ln.b", ds[0].bias)
            put(f"ds{i}", _patch_weight(ds[1].weight), ds[1].bias, [4 * d[i - 1]])
        for i in range(4):
            for j, blk in enumerate(cn.stages[i]):
                q = f"s{i}.{j}."
                v(q + "dw.w", blk.dwconv.weight), v(q + "dw.b", blk.dwconv.bias)
                v(q + "ln.w", blk.norm.weight), v(q + "ln.b", blk.norm.bias)
                put(q + "pw1", blk.pwconv1.weight, blk.pwconv1.bias, [d[i]])
                v(q + "grn.g", blk.grn.gamma), v(q + "grn.b", blk.grn.beta)
                put(q + "pw2", blk.pwconv2.weight, blk.pwconv2.bias, [4 * d[i]])
        put("up16", self.upconv_16[1].weight, self.upconv_16[1].bias, [d[3]])
        put("up8", self.upconv_8[1].weight, self.upconv_8[1].bias, [256])
        put("up4", self.upconv_4[1].weight, self.upconv_4[1].bias, [256])
        for tag, c in (("16", d[2]), ("8", d[1]), ("4", d[0])):
            m = getattr(self, f"decode_{tag}x")
            put(f"dec{tag}.0", m[0].weight, m[0].bias, [c, 256])
            put(f"dec{tag}.3", m[3].weight, m[3].bias, [256])
        self._packed = (pk, vec)
        return self._packed

    # ------------------------------------------------------------------------------------------------ forward
    @torch.no_grad()
    def forward(self, x: torch.Tensor):
        """x: (N, 3, H, W) fp32 on the GPU, H, W multiples of 32 -> (c4, c8, c16), each (N, 256, H/s, W/s) fp32 (convnext.py:256-264)."""
        if not (torch.is_tensor(x) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 4 and x.shape[1] == 3):
            raise RuntimeError("ppmstereo_amd Feature: an (N, 3, H, W) fp32 tensor on the MI355X expected (no CPU path)")
        N, _, H, W = x.shape
        if H % 32 or W % 32:
            raise ValueError(f"ppmstereo_amd Feature: H = {H}, W = {W} must be multiples of 32 (InputPadder(divis_by=32), ppmstereo.py:251)")
        key = (N, H, W, x.device.index)
        eng = self._engines.get(key)
        if eng is None:
            with torch.cuda.device(x.device):
                eng = _CnetEngine(self._pack(x.device), N, H, W, x.device)
            self._engines[key] = eng
            while len(self._engines) > 2:
                self._engines.popitem(last=False)
        with torch.cuda.device(x.device):                        # launches go to the current stream OF THE TENSOR'S device
            return eng.run(x.contiguous())


class _CnetEngine:
    LN_EPS, IN_EPS = 1e-6, 1e-5

    def __init__(self, packed, N: int, H: int, W: int, device):
        pk, vec = packed
        self.lib = lib = L.load()
        self.N, self.H, self.W = N, H, W
        d = CNET_DIMS
        hs = [H // 4, H // 8, H // 16, H // 32]
        ws = [W // 4, W // 8, W // 16, W // 32]
        Ps = [N * hs[i] * ws[i] for i in range(4)]
        self.steps: List = []
        self.keep: List = []            # EVERY buffer of the plan: the launch descriptors hold raw pointers only

        def sp(p, c):
            t = L.SPTensor(p, c, device)
            self.keep.append(t)
            return t

        def f32(p, c):
            t = torch.empty(p, c, device=device, dtype=torch.float32)
            self.keep.append(t)
            return t

        s = L.stream_ptr
        none_sp = L.SP(None, None, 0, 0)
        self.stats = torch.empty(N * 256 * 2, device=device, dtype=torch.float32)
        in_ws = max(int(lib.ppms_instnorm_workspace_bytes(N, hs[i] * ws[i], 256)) for i in range(3))
        grn_ws = max(int(lib.ppms_grn_workspace_bytes(N, hs[i] * ws[i], 4 * d[i])) for i in range(4))
        self.ws = torch.empty(max(in_ws, grn_ws), device=device, dtype=torch.uint8)

        def conv(name, segs: List[L.SP], n_h_w, e0: L.Epilogue):
            packed_w, bias, meta, k2 = pk[name]
            dd = L.Conv()
            for i, t in enumerate(segs):
                dd.seg[i] = t
            dd.nseg, dd.w, dd.bias = len(segs), packed_w.data_ptr(), bias.data_ptr()
            dd.T, dd.H, dd.W = n_h_w
            dd.kt, dd.kh, dd.kw = 1, k2[0], k2[1]
            dd.M = dd.m_split = meta["M"]
            assert [t.c for t in segs] == list(meta["seg_padded"]), (name, [t.c for t in segs], meta["seg_padded"])
            dd.epi[0] = e0
            if name + "@1" in pk:
                p1, b1, _ = pk[name + "@1"]
                d1 = L.Conv.from_buffer_copy(bytes(dd))
                d1.w, d1.bias = p1.data_ptr(), b1.data_ptr()
                if lib.ppms_gemm1_applicable(C.byref(d1)) == 1:
                    self.steps.append(ConvOp(d1, [p1, b1], 6, device=device))
                    return
            if TUNING["stream"] and name + "@7" in pk:                        # small maps: no K slices, no reduce launch (the library rates it)
                p7, b7, _ = pk[name + "@7"]
                d7 = L.Conv.from_buffer_copy(bytes(dd))
                d7.w, d7.bias = p7.data_ptr(), b7.data_ptr()
                if lib.ppms_conv_stream_applicable(C.byref(d7)) == 1:
                    self.steps.append(ConvOp(d7, [p7, b7], 7, wm_hint=TUNING["stream_hint"], device=device))
                    return
            self.steps.append(ConvOp(dd, [packed_w, bias], 2, device=device))

        def call(fn):
            self.steps.append(fn)

        def layernorm(src_f32, C_, wn, bn, dst: L.SPTensor, P_):
            w_, b_ = vec[wn], vec[bn]
            dv = dst.view()
            call(lambda: L.check(lib.ppms_layernorm_any(src_f32.data_ptr(), src_f32.shape[1], w_.data_ptr(), b_.data_ptr(), self.LN_EPS, dv, P_, C_, s())))

        def inorm(src_f32, C_, hw, dst_view: L.SP, relu: bool):
            ld = src_f32.shape[1]

            def fn():
                L.check(lib.ppms_instnorm_stats(src_f32.data_ptr(), ld, N, hw, C_, self.IN_EPS, self.stats.data_ptr(), self.ws.data_ptr(), s()))
                L.check(lib.ppms_instnorm_apply(src_f32.data_ptr(), ld, self.stats.data_ptr(), none_sp, int(relu), dst_view, N, hw, C_, s()))
            call(fn)

        # ---- backbone ------------------------------------------------------------------------------------------------------------
        self.s0 = sp(Ps[0], 64)                                     # 4x4 patches of the image: 48 values + padding
        outs: List[L.SPTensor] = []
        x: Optional[L.SPTensor] = None
        for i in range(4):
            C_, P_, h_, w_ = d[i], Ps[i], hs[i], ws[i]
            t = f32(P_, ((C_ + 63) // 64) * 64)                     # conv output (couts padded to the 64-row MFMA block)
            if i == 0:
                conv("stem", [self.s0.view()], (N, h_, w_), epilogue(n_valid=t.shape[1], out_f32=t, out_f32_ld=t.shape[1]))
                x = sp(P_, C_)
                layernorm(t, C_, "stem.ln.w", "stem.ln.b", x, P_)
            else:
                Cp, Pp = d[i - 1], Ps[i - 1]
                xf = f32(Pp, Cp)                                   # LN wants fp32 input: the stream as fp32
                xn, xs = sp(Pp, Cp), sp(P_, 4 * Cp)
                xv, xnv, xsv = x.view(), xn.view(), xs.view()
                call(lambda xv=xv, xf=xf, Pp=Pp, Cp=Cp: L.check(lib.ppms_sp_to_f32(xv, xf.data_ptr(), Cp, Pp, s())))
                layernorm(xf, Cp, f"ds{i}.ln.w", f"ds{i}.ln.b", xn, Pp)
                call(lambda xnv=xnv, xsv=xsv, hp=hs[i - 1], wp=ws[i - 1]: L.check(lib.ppms_sp_s2d(xnv, xsv, N, hp, wp, s())))
                x = sp(P_, C_)
                conv(f"ds{i}", [xs.view()], (N, h_, w_), epilogue(n_valid=C_, out_sp=x.view()))
                self.keep += [xf, xn, xs]
            t1, hbuf = f32(P_, C_), f32(P_, 4 * C_)
            a, g = sp(P_, C_), sp(P_, 4 * C_)
            for j in range(CNET_DEPTHS[i]):
                q = f"s{i}.{j}."
                y = sp(P_, C_)
                xv = x.view()
                dw_w, dw_b = vec[q + "dw.w"], vec[q + "dw.b"]
                call(lambda xv=xv, t1=t1, dw_w=dw_w, dw_b=dw_b, C_=C_, h_=h_, w_=w_: L.check(
                    lib.ppms_dwconv(xv, t1.data_ptr(), C_, dw_w.data_ptr(), dw_b.data_ptr(), 7, N, h_, w_, s())))
                layernorm(t1, C_, q + "ln.w", q + "ln.b", a, P_)
                conv(q + "pw1", [a.view()], (N, h_, w_), epilogue(act=L.ACT_GELU, n_valid=4 * C_, out_f32=hbuf, out_f32_ld=4 * C_))
                gg, gb, gv = vec[q + "grn.g"], vec[q + "grn.b"], g.view()
                call(lambda hbuf=hbuf, gg=gg, gb=gb, gv=gv, C_=C_, hw=h_ * w_: L.check(
                    lib.ppms_grn(hbuf.data_ptr(), 4 * C_, gg.data_ptr(), gb.data_ptr(), gv, N, hw, 4 * C_, self.ws.data_ptr(), s())))
                conv(q + "pw2", [g.view()], (N, h_, w_), epilogue(L.EPI_RESID, n_valid=C_, out_sp=y.view(), aux_sp=x.view()))
                self.keep.append(x)
                x = y
            outs.append(x)
            self.keep += [t, t1, hbuf, a, g]
        x4, x8, x16, x32 = outs

        # ---- FPN decoder (convnext.py:225-253, 259-261) -------------------------------------------------------------------------
        self.finals = []
        prev = x32
        for lvl, skip, tag in ((2, x16, "16"), (1, x8, "8"), (0, x4, "4")):
            P_, h_, w_ = Ps[lvl], hs[lvl], ws[lvl]
            upx = sp(P_, prev.channels)
            pv, uv = prev.view(), upx.view()
            call(lambda pv=pv, uv=uv, hh=hs[lvl + 1], ww=ws[lvl + 1]: L.check(lib.ppms_sp_upsample2(pv, uv, N, hh, ww, s())))
            t = f32(P_, 256)
            conv(f"up{tag}", [upx.view()], (N, h_, w_), epilogue(n_valid=256, out_f32=t, out_f32_ld=256))
            u = sp(P_, 256)
            inorm(t, 256, h_ * w_, u.view(), True)
            t2 = f32(P_, 256)
            conv(f"dec{tag}.0", [skip.view(), u.view()], (N, h_, w_), epilogue(n_valid=256, out_f32=t2, out_f32_ld=256))
            m = sp(P_, 256)
            inorm(t2, 256, h_ * w_, m.view(), True)
            fo = f32(P_, 256)
            conv(f"dec{tag}.3", [m.view()], (N, h_, w_), epilogue(n_valid=256, out_f32=fo, out_f32_ld=256))
            nxt = sp(P_, 256)                                       # the next level's input, as split planes
            nv = nxt.view()
            call(lambda fo=fo, nv=nv, P_=P_: L.check(lib.ppms_f32_to_sp(fo.data_ptr(), 256, nv, P_, s())))
            self.finals.append((fo, h_, w_))
            self.keep += [upx, t, u, t2, m, nxt, skip]
            prev = nxt

    def run(self, img: torch.Tensor):
        lib, s = self.lib, L.stream_ptr
        L.check(lib.ppms_img_s2d(img.data_ptr(), self.s0.view(), self.N, 3, self.H, self.W, 4, s()))
        for st in self.steps:
            st()
        outs = []
        for fo, h_, w_ in self.finals:                              # 1/16, 1/8, 1/4
            o = torch.empty(self.N, 256, h_, w_, device=img.device, dtype=torch.float32)
            L.check(lib.ppms_nhwc_to_nchw(fo.data_ptr(), 256, o.data_ptr(), self.N, 256, h_ * w_, s()))
            outs.append(o)
        return outs[2], outs[1], outs[0]
