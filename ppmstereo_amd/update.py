"""Update-block modules with the reference's constructor / method signatures and ``state_dict`` layout
(/root/reference/models/core/ppmtereo_update.py: SequenceUpdateBlock3D :880-1003, Attention_qk :118-133,
get_temporal_positional_encoding :25-88), executing on the gfx950 kernels through ``ScaleEngine``.

The nn.Conv*/Linear/LayerNorm children below are parameter holders only (same names and shapes as the reference, so
its checkpoints load unchanged); their ``forward`` is never called.  There is no CPU path.
"""
from __future__ import annotations

import math
from collections import OrderedDict
from typing import Optional, Tuple

import torch
import torch.nn as nn

from . import _lib as L
from .engine import PackedBlock, ScaleEngine, pack_conv, temporal_pe


@torch.no_grad()
def get_temporal_positional_encoding(max_sequence_len, channels, device, is_normalize=False, scale=2 * math.pi, is_debug=False):
    """ppmtereo_update.py:25-88 -- (T, 1, 1, channels)."""
    if is_normalize and scale == 1.0:
        pe = temporal_pe(max_sequence_len, channels)
    else:
        pos = torch.arange(max_sequence_len)
        if is_normalize:
            pos = pos / pos[-1] * scale
        pos = pos.unsqueeze(1)
        div = 1.0 / (10000.0 ** (torch.arange(0, channels, 2).float() / channels))
        ang = pos * div
        pe = torch.zeros(max_sequence_len, channels)
        pe[:, 0::2], pe[:, 1::2] = torch.sin(ang), torch.cos(ang)
    return pe.view(max_sequence_len, 1, 1, channels).to(device)


class _Holder(nn.Module):
    def forward(self, *a, **k):  # pragma: no cover
        raise RuntimeError("parameter holder: the gfx950 engine runs this layer")


class _PCBlock(_Holder):
    """PCBlock4_Deep_nopool_res parameters (ppmtereo_update.py:1006-1022)."""

    def __init__(self, c_in, c_out, k_conv):
        super().__init__()
        self.conv_list = nn.ModuleList([nn.Conv2d(c_in, c_in, k, padding=k // 2, groups=c_in) for k in k_conv])
        mid = int(1.5 * c_in)
        self.ffn1 = nn.Sequential(nn.Conv2d(c_in, mid, 1), nn.GELU(), nn.Conv2d(mid, c_in, 1))
        self.pw = nn.Conv2d(c_in, c_in, 1)
        self.ffn2 = nn.Sequential(nn.Conv2d(c_in, mid, 1), nn.GELU(), nn.Conv2d(mid, c_out, 1))


class _MotionEncoder(_Holder):
    """BasicMotionEncoder_v2 parameters (ppmtereo_update.py:445-464)."""

    def __init__(self, cor_planes):
        super().__init__()
        self.convc1 = _PCBlock(cor_planes, 256, [1, 7])
        self.convc2 = nn.Conv2d(256, 192, 3, padding=1)
        self.convf1 = nn.Conv2d(2, 128, 7, padding=3)
        self.convf2 = nn.Conv2d(128, 64, 3, padding=1)
        self.final_conv = nn.Conv2d(320, 190, 3, padding=1)
        self.init_conv = nn.Sequential(nn.Conv2d(128, 64, 3, padding=1), nn.ReLU(inplace=True), nn.Conv2d(64, 64, 3, padding=1))


class _GRU(_Holder):
    """SKSepConvGRU3D parameters (ppmtereo_update.py:254-289)."""

    def __init__(self, hidden_dim, input_dim):
        super().__init__()
        c = hidden_dim + input_dim
        two = lambda: nn.Sequential(nn.Conv3d(c, hidden_dim, (1, 1, 15), padding=(0, 0, 7)), nn.GELU(),
                                    nn.Conv3d(hidden_dim, hidden_dim, (1, 1, 5), padding=(0, 0, 2)))
        self.convz1, self.convr1 = two(), two()
        self.convq1 = nn.Conv3d(c, hidden_dim, (1, 1, 5), padding=(0, 0, 2))
        for n, k, p in (("2", (1, 5, 1), (0, 2, 0)), ("3", (5, 1, 1), (2, 0, 0))):
            for g in "zrq":
                setattr(self, f"conv{g}{n}", nn.Conv3d(c, hidden_dim, k, padding=p))


class _FlowHead(_Holder):
    def __init__(self, input_dim=128, hidden_dim=256):
        super().__init__()
        self.conv1 = nn.Conv3d(input_dim, hidden_dim, 3, padding=1)
        self.conv2 = nn.Conv3d(hidden_dim, 2, 3, padding=1)


class _Attention(_Holder):
    def __init__(self, dim):
        super().__init__()
        self.qkv = nn.Linear(dim, dim * 3, bias=False)        # never applied by the reference either (:406,411-412)
        self.proj = nn.Linear(dim, dim)


class _TimeAttn(_Holder):
    def __init__(self, dim):
        super().__init__()
        self.temporal_attn = _Attention(dim)
        self.temporal_fc = nn.Linear(dim, dim)
        self.temporal_norm1 = nn.LayerNorm(dim)
        nn.init.constant_(self.temporal_fc.weight, 0)
        nn.init.constant_(self.temporal_fc.bias, 0)


class _LoFTR(_Holder):
    def __init__(self, d):
        super().__init__()
        self.q_proj, self.k_proj, self.v_proj, self.merge = (nn.Linear(d, d, bias=False) for _ in range(4))
        self.mlp = nn.Sequential(nn.Linear(2 * d, 2 * d, bias=False), nn.ReLU(), nn.Linear(2 * d, d, bias=False))
        self.norm1, self.norm2 = nn.LayerNorm(d), nn.LayerNorm(d)


class _SpaceAttn(_Holder):
    def __init__(self, dim):
        super().__init__()
        self.encoder_layer = _LoFTR(dim)


class _Aggregate(_Holder):
    def __init__(self, dim=128):
        super().__init__()
        self.to_v = nn.Conv2d(dim, dim, 1, bias=False)
        self.beta = nn.Parameter(torch.zeros(1))


class Attention_qk(nn.Module):
    """ppmtereo_update.py:118-133."""

    def __init__(self, *, num_heads=1, dim_head=128):
        super().__init__()
        if dim_head != 128:
            raise NotImplementedError("Attention_qk: dim_head must be 128")
        self.heads, self.scale = num_heads, dim_head ** -0.5
        self.to_qk = nn.Conv2d(dim_head, dim_head * 2, 1, bias=False)
        self._packed = None
        self.register_load_state_dict_post_hook(lambda m, k: setattr(m, "_packed", None))

    def packed(self, device):
        if self._packed is None or self._packed[0].device != torch.device(device):
            self._packed = pack_conv(self.to_qk.weight.detach().to(device).float(), None, [128])
        return self._packed

    def forward(self, fmap):
        raise RuntimeError("Attention_qk.forward is not on the hot path; forward_update_block projects q/k inside the engine")


class SequenceUpdateBlock3D(nn.Module):
    """ppmtereo_update.py:880-1003; use_convex_3d selects the mask_3d head (:903-908, 993-996) instead of mask_2d."""

    def __init__(self, hidden_dim, cor_planes, mask_size=8, use_convex_3d=False, attention_type=None):
        super().__init__()
        if hidden_dim != 128 or cor_planes != 36 or mask_size != 4:
            raise NotImplementedError("SequenceUpdateBlock3D: hidden_dim=128, cor_planes=36, mask_size=4 (the PPMStereo configuration)")
        self.encoder = _MotionEncoder(cor_planes)
        self.gru = _GRU(hidden_dim, 256 + hidden_dim)
        self.flow_head = _FlowHead(hidden_dim, 256)
        self.uncertainty = nn.Sequential(nn.Conv2d(hidden_dim + 128, hidden_dim, 3, padding=1), nn.ReLU(inplace=True),
                                         nn.Conv2d(hidden_dim, 1, 1), nn.Sigmoid())
        self.use_convex_3d = bool(use_convex_3d)
        if self.use_convex_3d:                               # ppmtereo_update.py:903-908
            self.mask_3d = nn.Sequential(nn.Conv3d(hidden_dim, hidden_dim + 128, 3, padding=1), nn.ReLU(inplace=True),
                                         nn.Conv3d(hidden_dim + 128, (mask_size ** 2) * 27, 1, padding=0))
        else:                                                # :910-914
            self.mask_2d = nn.Sequential(nn.Conv2d(hidden_dim, hidden_dim + 128, 3, padding=1), nn.ReLU(inplace=True),
                                         nn.Conv2d(hidden_dim + 128, (mask_size ** 2) * 9, 1))
        self.attention_type = attention_type
        if attention_type is not None:
            if "update_time" in attention_type:
                self.time_attn = _TimeAttn(384)
            if "update_space" in attention_type:
                self.space_attn = _SpaceAttn(384)
            if not ("update_time" in attention_type and "update_space" in attention_type):
                raise NotImplementedError("attention_type must contain both update_time and update_space (or be None)")
        self.aggregator = _Aggregate(128)
        self._pk: Optional[PackedBlock] = None
        # engines hold every buffer of a scale (gigabytes at 736x1280): keep the two most recently used geometries only
        # (a long video's last window may have a different T than the others)
        self._engines: "OrderedDict[Tuple, ScaleEngine]" = OrderedDict()
        self.register_load_state_dict_post_hook(lambda m, k: m.invalidate())

    MAX_ENGINES = 2

    # ------------------------------------------------------------------ engine plumbing
    def invalidate(self):
        """Call after changing parameters in place: weights are re-packed on next use."""
        self._pk = None
        self._engines = OrderedDict()

    def packed(self, device) -> PackedBlock:
        if self._pk is None or self._pk.beta.device != torch.device(device):
            sd = {k: v for k, v in self.state_dict().items()}
            with torch.cuda.device(device):
                self._pk = PackedBlock(sd, device)
            self._engines = OrderedDict()
        return self._pk

    def engine(self, T: int, h: int, w: int, device, shard=None, slot: int = 0) -> ScaleEngine:
        """slot: which batch element the engine serves (forward_update_block with b > 1 keeps one engine per element)."""
        device = torch.device(device)
        if device.index is None:
            device = torch.device("cuda", torch.cuda.current_device())
        key = (T, h, w, str(device), None if shard is None else (shard.rank, shard.world, shard.T, id(shard.group), bool(getattr(shard, "force_comm", False))), slot)
        pk = self.packed(device)
        if key not in self._engines:
            while len(self._engines) >= max(self.MAX_ENGINES, slot + 1):
                self._engines.popitem(last=False)
            with torch.cuda.device(device):
                self._engines[key] = ScaleEngine(pk, T, h, w, device, shard)
        self._engines.move_to_end(key)
        eng = self._engines[key]
        if shard is not None and eng.shard is not None:
            eng.shard = shard                      # same geometry and group: the caller's object (a new one per window is fine)
        return eng

    # ------------------------------------------------------------------ reference methods (NCHW in / NCHW out)
    def get_motion_and_value(self, flow, corr, motion_hidden_state, inp):
        """ppmtereo_update.py:945-950: (mf (N,128,h,w), mhs (N,64,h,w), value (N,128,h,w))."""
        L.require_gpu(flow, corr, inp)
        N, _, h, w = flow.shape
        with torch.cuda.device(flow.device):
            e = self.engine(N, h, w, flow.device)
            e.set_flow(flow)
            e.set_inp(inp)
            e.set_mhs(motion_hidden_state)
            e.load_nchw(corr, e.CORR.view(0, 36))
            L.check(e.lib.ppms_f32_to_sp(e.FLOW.data_ptr(), 2, e.X.view(254, 2), e.P, L.stream_ptr()))
            e.motion_and_value()
            return e.get_mf(), e.get_mhs(), e.get_value()

    def get_uncertainty(self, net):
        """ppmtereo_update.py:936-938 on cat([net, value]) (N,256,h,w) -> (N,1,h,w)."""
        L.require_gpu(net)
        N, c, h, w = net.shape
        if c != 256:
            raise RuntimeError("get_uncertainty expects cat([net, value]) with 256 channels")
        with torch.cuda.device(net.device):
            e = self.engine(N, h, w, net.device)
            e.set_net(net[:, :128])
            e.load_nchw(net[:, 128:], e.VAL.view())
            e.uncertainty()
            return e.get_unc()

    def forward(self, net, inp, motion_features, motion_features_global, t=1):
        """ppmtereo_update.py:971-1003 -> (net, mask, delta_flow)."""
        L.require_gpu(net, inp, motion_features, motion_features_global)
        N, _, h, w = net.shape
        if N != t:
            raise NotImplementedError("batch size 1 only: the frame axis of the 3-D convolutions is the whole batch")
        with torch.cuda.device(net.device):
            e = self.engine(N, h, w, net.device)
            e.set_net(net)
            e.set_inp(inp)
            e.set_mf(motion_features)
            e.set_mfg(motion_features_global)
            e.update()
            return e.get_net(), e.get_mask(), e.get_dflow()
