"""The SST block on MI355X: ``PPMStereo.forward_sst_block`` (/root/reference/models/core/ppmstereo.py:322-395) for the attention type the
model ships with ("self_stereo_temporal_update_time_update_space") -- SURVEY.md section 8 row f4.  On the 1/16 features of both views:
+ 2-D sine positional encoding, + time embedding, then 4 x { LoFTR "self" layer on each view, "cross" layer (the second call sees the
UPDATED first view, attention.py:230-232), TimeAttnBlock(256) on each view }.

Parameters keep the reference's names, shapes and order (``time_embed``, ``time_attn_blocks.i.*``, ``self_attn_blocks.i.layers.0.*``,
``cross_attn_blocks.i.layers.0.*``; ``tools/gen_golden.py`` loads this repo's weights into the reference's modules).  Every Linear layer is
a 1x1 implicit-GEMM launch of libppms (fp32-accurate bf16x3 MFMA), LayerNorm / per-pixel temporal attention / linear-attention sums are
the attn16.hip kernels (C = 256 instantiations); activations stay channel-last on the GPU between the two NCHW ends.
"""
from __future__ import annotations

import ctypes as C
from collections import OrderedDict
from typing import Dict, List, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from . import _lib as L
from . import packing as _packing
from .engine import ConvOp, epilogue

DIM, HEADS, DEPTH = 256, 8, 4


class _Attention(nn.Module):                       # ppmtereo_update.py:400-408 (qkv is never applied: a dead parameter)
    def __init__(self, dim):
        super().__init__()
        self.qkv = nn.Linear(dim, 3 * dim, bias=False)
        self.proj = nn.Linear(dim, dim)


class _TimeAttnBlock(nn.Module):                   # ppmtereo_update.py:593-601
    def __init__(self, dim):
        super().__init__()
        self.temporal_attn = _Attention(dim)
        self.temporal_fc = nn.Linear(dim, dim)
        self.temporal_norm1 = nn.LayerNorm(dim)


class _LoFTRLayer(nn.Module):                      # attention.py:140-162
    def __init__(self, d):
        super().__init__()
        self.q_proj = nn.Linear(d, d, bias=False)
        self.k_proj = nn.Linear(d, d, bias=False)
        self.v_proj = nn.Linear(d, d, bias=False)
        self.merge = nn.Linear(d, d, bias=False)
        self.mlp = nn.Sequential(nn.Linear(2 * d, 2 * d, bias=False), nn.ReLU(True), nn.Linear(2 * d, d, bias=False))
        self.norm1 = nn.LayerNorm(d)
        self.norm2 = nn.LayerNorm(d)


class _Transformer(nn.Module):                     # attention.py:193-205 with one layer
    def __init__(self, d):
        super().__init__()
        self.layers = nn.ModuleList([_LoFTRLayer(d)])


def position_encoding_sine(d_model: int, h: int, w: int) -> torch.Tensor:
    """PositionEncodingSine (temp_bug_fix=True), attention.py:23-64 -> (d_model, h, w); host side, once per map size."""
    import math
    y = torch.ones(h, w).cumsum(0)[None]
    x = torch.ones(h, w).cumsum(1)[None]
    div = torch.exp(torch.arange(0, d_model // 2, 2).float() * (-math.log(10000.0) / (d_model // 2)))[:, None, None]
    pe = torch.zeros(d_model, h, w)
    pe[0::4], pe[1::4] = torch.sin(x * div), torch.cos(x * div)
    pe[2::4], pe[3::4] = torch.sin(y * div), torch.cos(y * div)
    return pe


class SSTBlock(nn.Module):
    """Callable ``(fmap1_dw16, fmap2_dw16, T) -> (fmap1_dw16, fmap2_dw16)``, (BT, 256, h, w) fp32 on the GPU, batch 1."""

    def __init__(self, dim: int = DIM, num_frames: int = 5, depth: int = DEPTH):
        super().__init__()
        if dim != DIM or depth != DEPTH:
            raise NotImplementedError("ppmstereo_amd SSTBlock: dim = 256, depth = 4 (ppmstereo.py:65,95)")
        self.num_frames = num_frames
        self.time_embed = nn.Parameter(torch.zeros(1, num_frames, dim))
        self.time_attn_blocks = nn.ModuleList([_TimeAttnBlock(dim) for _ in range(depth)])
        self.self_attn_blocks = nn.ModuleList([_Transformer(dim) for _ in range(depth)])
        self.cross_attn_blocks = nn.ModuleList([_Transformer(dim) for _ in range(depth)])
        self._engines: "OrderedDict[tuple, _SstEngine]" = OrderedDict()
        self._packed = None

    def load_state_dict(self, sd, strict: bool = True, **kw):
        r = super().load_state_dict(sd, strict=strict, **kw)
        self.invalidate()
        return r

    def invalidate(self) -> None:
        self._packed = None
        self._engines.clear()

    def _pack(self, device):
        if self._packed is not None:
            return self._packed
        pk: Dict[str, tuple] = {}

        def put(name, w, b, segs):
            w4 = w.detach().to(device)[:, :, None, None]
            packed, bias, meta = _packing.pack_conv2(w4, None if b is None else b.detach().to(device), segs, segs)
            pk[name] = (packed, bias, meta)
            # every layer of the block is a 1x1 GEMM on 640 ... 3 680 pixels per frame: the thin-GEMM kernel's packs beside the implicit GEMM's
            pk[name + "@1"] = _packing.pack_gemm1(w4, None if b is None else b.detach().to(device), segs, segs, None, meta["M"])

        ln: Dict[str, tuple] = {}
        for i in range(DEPTH):
            ta = self.time_attn_blocks[i]
            # temporal_fc(proj(.)): two Linear layers with nothing in between = one (W_fc W_proj, W_fc b_proj + b_fc; formed in fp64)
            w_fc, w_pj = ta.temporal_fc.weight.detach().double(), ta.temporal_attn.proj.weight.detach().double()
            put(f"ta{i}.fc", (w_fc @ w_pj).float(), (w_fc @ ta.temporal_attn.proj.bias.detach().double() + ta.temporal_fc.bias.detach().double()).float(), [DIM])
            ln[f"ta{i}"] = (ta.temporal_norm1.weight.detach().float().to(device).contiguous(), ta.temporal_norm1.bias.detach().float().to(device).contiguous())
            for kind, blocks in (("self", self.self_attn_blocks), ("cross", self.cross_attn_blocks)):
                ly = blocks[i].layers[0]
                p = f"{kind}{i}."
                put(p + "q", ly.q_proj.weight, None, [DIM])
                put(p + "kv", torch.cat([ly.k_proj.weight, ly.v_proj.weight], 0), None, [DIM])
                put(p + "merge", ly.merge.weight, None, [DIM])
                put(p + "mlp0", ly.mlp[0].weight, None, [DIM, DIM])
                put(p + "mlp2", ly.mlp[2].weight, None, [2 * DIM])
                for n in ("norm1", "norm2"):
                    m = getattr(ly, n)
                    ln[p + n] = (m.weight.detach().float().to(device).contiguous(), m.bias.detach().float().to(device).contiguous())
        self._packed = (pk, ln)
        return self._packed

    def forward(self, f1: torch.Tensor, f2: torch.Tensor, T: int):
        L.require_gpu(f1, f2)
        if f1.shape != f2.shape or f1.dim() != 4 or f1.shape[1] != DIM or f1.shape[0] != T:
            raise RuntimeError("ppmstereo_amd SSTBlock: two (T, 256, h, w) feature maps of one clip expected (batch 1)")
        _, _, h, w = f1.shape
        key = (T, h, w, f1.device.index)
        eng = self._engines.get(key)
        if eng is None:
            te = self.time_embed.detach().float()
            if T != self.num_frames:                                  # ppmstereo.py:347-352
                te = F.interpolate(te.transpose(1, 2), size=T, mode="nearest").transpose(1, 2).contiguous()
            with torch.cuda.device(f1.device):
                eng = _SstEngine(self._pack(f1.device), te[0], T, h, w, f1.device)
            self._engines[key] = eng
            while len(self._engines) > 2:
                self._engines.popitem(last=False)
        with torch.cuda.device(f1.device):                       # launches go to the current stream OF THE TENSOR'S device
            return eng.run(f1.contiguous().float(), f2.contiguous().float())


class _SstEngine:
    def __init__(self, packed, te: torch.Tensor, T: int, h: int, w: int, device):
        pk, ln = packed
        self.lib = lib = L.load()
        self.T, self.h, self.w, self.n = T, h, w, h * w
        n, P = self.n, T * h * w
        self.P = P
        # positional encoding + time embedding as ONE (T, 256, h, w) addend (both are per-(frame, channel, pixel) constants)
        self.addend = (position_encoding_sine(DIM, h, w)[None] + te.cpu()[:, :, None, None]).to(device).contiguous()
        sp = lambda c: L.SPTensor(P, c, device)
        f32 = lambda c: torch.empty(P, c, device=device, dtype=torch.float32)
        self.xin = torch.empty(T, DIM, h, w, device=device, dtype=torch.float32)
        self.X = [sp(DIM), sp(DIM)]                 # the two views
        self.Y = [sp(DIM), sp(DIM)]                 # ping-pong partners
        self.QF, self.KVF, self.M2, self.M3 = f32(DIM), f32(2 * DIM), f32(DIM), f32(DIM)
        self.MSG, self.MSGN, self.H1, self.O1 = sp(DIM), sp(DIM), sp(2 * DIM), sp(DIM)
        self.KVWS = torch.empty(int(lib.ppms_linear_attention_workspace_floats(T, n, HEADS, DIM // HEADS)) + 64, device=device, dtype=torch.float32)
        self.steps: List = []
        none_sp = L.SP(None, None, 0, 0)
        s = L.stream_ptr

        def conv(name, segs: List[L.SPTensor], e0: L.Epilogue, e1: Optional[L.Epilogue] = None, m_split: Optional[int] = None):
            packed_w, bias, meta = pk[name]
            d = L.Conv()
            for i, t in enumerate(segs):
                d.seg[i] = t.view()
            d.nseg, d.w, d.bias = len(segs), packed_w.data_ptr(), bias.data_ptr()
            d.T, d.H, d.W = T, h, w
            d.kt = d.kh = d.kw = 1
            d.M = meta["M"]
            d.m_split = meta["M"] if m_split is None else m_split
            d.epi[0] = e0
            if e1 is not None:
                d.epi[1] = e1
            p1, b1, _ = pk[name + "@1"]
            d1 = L.Conv.from_buffer_copy(bytes(d))
            d1.w, d1.bias = p1.data_ptr(), b1.data_ptr()
            if lib.ppms_gemm1_applicable(C.byref(d1)) == 1:           # (one memory round trip deep, no K slices: gemm1.hip)
                op = ConvOp(d1, list(segs) + [p1, b1], 6, device=device)
            else:
                op = ConvOp(d, list(segs) + [packed_w, bias], 2, device=device)
            self.steps.append(op)

        def call(fn):
            self.steps.append(fn)

        def loftr(p: str, x: L.SPTensor, src: L.SPTensor, out: L.SPTensor):
            """out = x + LN2(mlp(cat[x, LN1(merge(linear_attention(q(x), k(src), v(src))))]))        attention.py:164-190, 73-100"""
            conv(p + "q", [x], epilogue(act=L.ACT_ELU1, n_valid=DIM, out_f32=self.QF, out_f32_ld=DIM))
            conv(p + "kv", [src], epilogue(act=L.ACT_ELU1, n_valid=DIM, out_f32=self.KVF, out_f32_ld=2 * DIM),
                 epilogue(scale=1.0 / n, n_valid=DIM, out_f32=self.KVF[:, DIM:], out_f32_ld=2 * DIM), m_split=DIM)
            call(lambda: L.check(lib.ppms_linear_attention(self.QF.data_ptr(), DIM, self.KVF.data_ptr(), 2 * DIM, self.KVF.data_ptr() + DIM * 4, 2 * DIM,
                                                           self.KVWS.data_ptr(), self.MSG.view(), T, n, HEADS, DIM // HEADS, s())))
            conv(p + "merge", [self.MSG], epilogue(n_valid=DIM, out_f32=self.M2, out_f32_ld=DIM))
            n1, n2 = ln[p + "norm1"], ln[p + "norm2"]
            call(lambda: L.check(lib.ppms_layernorm(self.M2.data_ptr(), DIM, n1[0].data_ptr(), n1[1].data_ptr(), none_sp, self.MSGN.view(), P, DIM, s())))
            conv(p + "mlp0", [x, self.MSGN], epilogue(act=L.ACT_RELU, n_valid=2 * DIM, out_sp=self.H1.view()))
            conv(p + "mlp2", [self.H1], epilogue(n_valid=DIM, out_f32=self.M3, out_f32_ld=DIM))
            xv, ov = x.view(), out.view()
            call(lambda: L.check(lib.ppms_layernorm(self.M3.data_ptr(), DIM, n2[0].data_ptr(), n2[1].data_ptr(), xv, ov, P, DIM, s())))

        def time_attn(i: int, x: L.SPTensor, out: L.SPTensor):
            """out = x + fc(proj(attn_T(LN(x))))                                              ppmtereo_update.py:603-618"""
            w_, b_ = ln[f"ta{i}"]
            xv = x.view()
            call(lambda: L.check(lib.ppms_time_attn(xv, w_.data_ptr(), b_.data_ptr(), self.O1.view(), T, n, HEADS, s())))
            conv(f"ta{i}.fc", [self.O1], epilogue(L.EPI_RESID, n_valid=DIM, out_sp=out.view(), aux_sp=x.view()))

        X, Y = self.X, self.Y
        for i in range(DEPTH):
            loftr(f"self{i}.", X[0], X[0], Y[0])            # feat0 = L(feat0, feat0); feat1 = L(feat1, feat1)
            loftr(f"self{i}.", X[1], X[1], Y[1])
            loftr(f"cross{i}.", Y[0], Y[1], X[0])           # feat0 = L(feat0, feat1)
            loftr(f"cross{i}.", Y[1], X[0], X[1])           # feat1 = L(feat1, UPDATED feat0)
            time_attn(i, X[0], Y[0])
            time_attn(i, X[1], Y[1])
            X, Y = Y, X
        self.final = X

    def run(self, f1: torch.Tensor, f2: torch.Tensor):
        lib, s = self.lib, L.stream_ptr
        T, h, w = self.T, self.h, self.w
        for v, f in enumerate((f1, f2)):
            L.check(lib.ppms_axpby(f.data_ptr(), self.addend.data_ptr(), self.xin.data_ptr(), 1.0, 1.0, self.addend.numel(), f.numel(), s()))
            L.check(lib.ppms_nchw_to_sp(self.xin.data_ptr(), self.X[v].view(), T, DIM, h * w, s()))
        for st in self.steps:
            st()
        outs = []
        for v in range(2):
            o = torch.empty(T, DIM, h, w, device=f1.device, dtype=torch.float32)
            L.check(lib.ppms_sp_to_nchw(self.final[v].view(), o.data_ptr(), T, DIM, h * w, s()))
            outs.append(o)
        return outs[0], outs[1]
