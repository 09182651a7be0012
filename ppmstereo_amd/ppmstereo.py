"""The hot loop with the reference's call signature, and the thin caller-side glue needed to drive it.

``forward_update_block`` keeps the signature and side effects of
``PPMStereo.forward_update_block`` (/root/reference/models/core/ppmstereo.py:426-594): it can be bound as a method
of the reference's ``PPMStereo`` or used through ``PPMStereoHotPath`` below, which also holds
the three update blocks / q-k projections under the reference's attribute names (``update_block16/08/04``,
``att``) so a reference checkpoint loads with ``strict=False``.
"""
from __future__ import annotations

import warnings
from typing import Dict, List, Optional

import torch
import torch.nn as nn

from . import _lib as L
from .corr import CorrBlock1D
from .engine import bilinear
from .update import Attention_qk, SequenceUpdateBlock3D


def interp(x: torch.Tensor, size) -> torch.Tensor:
    """models/core/utils/utils.py:10-16 (bilinear, align_corners=True)."""
    return bilinear(x, (int(size[0]), int(size[1])), True)


def convex_upsample(flow: torch.Tensor, mask: torch.Tensor, rate: int = 4) -> torch.Tensor:
    """PPMStereo.convex_upsample, ppmstereo.py:185-197 (NCHW in, NCHW out)."""
    if rate != 4:
        raise NotImplementedError("convex_upsample: rate 4 only")
    L.require_gpu(flow, mask)
    N, _, H, W = flow.shape
    lib, s = L.load(), L.stream_ptr()
    f = torch.empty(N * H * W, 2, dtype=torch.float32, device=flow.device)
    m = torch.empty(N * H * W, 144, dtype=torch.float32, device=flow.device)
    fc, mc = flow.contiguous().float(), mask.contiguous().float()
    L.check(lib.ppms_nchw_to_nhwc(fc.data_ptr(), f.data_ptr(), 2, N, 2, H * W, s))
    L.check(lib.ppms_nchw_to_nhwc(mc.data_ptr(), m.data_ptr(), 144, N, 144, H * W, s))
    out = torch.empty(N, 2, 4 * H, 4 * W, dtype=torch.float32, device=flow.device)
    L.check(lib.ppms_convex_upsample(f.data_ptr(), m.data_ptr(), 144, out.data_ptr(), N, H, W, s))
    return out


def _run_iterations(eng, iters: int, isc: int, t: int, h: int, w: int, predictions: List, uncertainties: List, emit: str = "all"):
    """The iteration loop on a prepared engine (ppmstereo.py:482-591).  emit: which iterations resize their prediction and
    uncertainty to full resolution and append them (:578-591): "all" (the reference's lists), "last" or "none" --
    PPMStereo.forward(test_mode=True) returns predictions[-1] alone (:801-804), so every other resize is dead work there."""
    flow_out = None
    for it in range(iters):
        flow_out = eng.iterate()
        if emit == "none" or (emit == "last" and it + 1 < iters):
            continue
        unc_up = bilinear(eng.UNC_local().view(t, 1, h, w), (4 * isc * h, 4 * isc * w), False)
        if isc > 1:
            flow_up = bilinear(flow_out[:, :1], (isc * 4 * h, isc * 4 * w), True, float(isc))
        else:
            flow_up = flow_out[:, :1].clone()
        predictions.append(flow_up)
        uncertainties.append(unc_up)
    return flow_out


def forward_update_block(self, image1, update_block: SequenceUpdateBlock3D, corr_fn: CorrBlock1D, flow: torch.Tensor, net: torch.Tensor,
                         inp: torch.Tensor, motion_hidden_state: Optional[torch.Tensor], attn_block: Attention_qk, predictions: List,
                         uncertainties: List, iters: int, interp_scale: float, t: int):
    """Same contract as PPMStereo.forward_update_block (ppmstereo.py:426-594): runs ``iters`` refinement
    iterations at one scale, appends one full-resolution prediction and uncertainty per iteration and returns
    (flow_out (BT,2,4h,4w), net (BT,128,h,w), motion_hidden_state (BT,64,h,w)).  ``image1`` is unused (as in the
    reference).  Batch size 1 (inference) only."""
    L.require_gpu(flow, net, inp)
    bt, c, h, w = inp.shape
    if bt != t:
        raise NotImplementedError("forward_update_block: batch size 1 only (bt == t)")
    if c != 128:
        raise RuntimeError("forward_update_block: 128 context channels expected")
    if int(interp_scale) not in (1, 2, 4):
        raise NotImplementedError("interp_scale must be 4, 2 or 1 (the reference's only live branches)")
    if t == 1:
        # the reference divides 0/0 in the temporal encoding (ppmtereo_update.py:34-36): every output is NaN
        warnings.warn("PPMStereo with a single frame produces NaN disparities (reference behaviour, T must be >= 2)")
    isc = int(interp_scale)
    with torch.cuda.device(inp.device):         # kernels go to the current stream of the tensors' device
        eng = update_block.engine(t, h, w, inp.device)
        eng.set_inp(inp)
        eng.set_net(net)
        eng.set_flow(flow)
        eng.set_mhs(motion_hidden_state)
        eng.begin(corr_fn.levels, attn_block.packed(inp.device))
        flow_out = _run_iterations(eng, iters, isc, t, h, w, predictions, uncertainties)
        return flow_out.clone(), eng.get_net(), eng.get_mhs()


class PPMStereoHotPath(nn.Module):
    """The part of PPMStereo that lives on the hot path (ppmstereo.py:82-117 modules, :426-594 loop, :696-804
    cascade), from the encoder / SST outputs on.  Attribute names follow the reference."""

    def __init__(self, max_disp: int = 192, mixed_precision: bool = False, num_frames: int = 5,
                 attention_type: Optional[str] = "self_stereo_temporal_update_time_update_space", use_3d_update_block: bool = True,
                 different_update_blocks: bool = True, use_convex_3d: bool = False, init_flow: bool = False):
        super().__init__()
        if not (use_3d_update_block and different_update_blocks) or use_convex_3d or init_flow:
            raise NotImplementedError("supported configuration: models/ppm_stereo_model.py:27-33 "
                                      "(use_3d_update_block=True, different_update_blocks=True, use_convex_3d=False, init_flow=False)")
        self.hidden_dim = self.context_dim = 128
        self.mixed_precision = mixed_precision      # the engine's precision is fixed: fp32-accurate convs, bf16 attention
        self.use_convex_3d = False
        self.num_frames = num_frames
        self.att = nn.ModuleList([Attention_qk(num_heads=1, dim_head=128) for _ in range(3)])
        self.update_block08 = SequenceUpdateBlock3D(hidden_dim=128, cor_planes=36, mask_size=4)
        self.update_block16 = SequenceUpdateBlock3D(hidden_dim=128, cor_planes=36, mask_size=4, attention_type=attention_type)
        self.update_block04 = SequenceUpdateBlock3D(hidden_dim=128, cor_planes=36, mask_size=4)

    forward_update_block = forward_update_block

    def convex_upsample(self, flow, mask, rate: int = 4):
        return convex_upsample(flow, mask, rate)

    def zero_init(self, fmap: torch.Tensor) -> torch.Tensor:
        """ppmstereo.py:231-236."""
        N, _, H, W = fmap.shape
        return torch.zeros(N, 2, H, W, dtype=torch.float32, device=fmap.device)

    def load_hot_path_weights(self, weights: Dict[str, Dict[str, torch.Tensor]]):
        """weights: {"update_block16": sd, ..., "att.0": sd, ...} (ppmstereo_amd.weights.hot_path_weights)."""
        for tag in ("update_block16", "update_block08", "update_block04"):
            getattr(self, tag).load_state_dict(weights[tag], strict=True)
        for i in range(3):
            self.att[i].load_state_dict(weights[f"att.{i}"], strict=True)
        return self

    @torch.no_grad()
    def cascade(self, feats: Dict[str, torch.Tensor], iters: int, t: int, predictions: Optional[list] = None,
                uncertainties: Optional[list] = None, shard=None, test_mode: bool = False):
        """The 1/16 -> 1/8 -> 1/4 cascade of PPMStereo.forward (ppmstereo.py:696-804), device resident: the state handed from
        scale to scale (hidden state, motion hidden state) stays in the engines' SP buffers (ppms_sp_resize_blend), only the
        2-channel flow passes through an NCHW resize.  feats: f1_s, f2_s, net_s, inp_s for s in (16, 8, 4) on the GPU
        (with ``shard``: this rank's frames only).  test_mode: only the final prediction is produced (ppmstereo.py:801-804).
        Returns (flow_up (T,1,H,W), uncertainty (T,1,H,W)) = predictions[-1], uncertainties[-1]."""
        preds = [] if predictions is None else predictions
        uncs = [] if uncertainties is None else uncertainties
        dev = feats["f1_16"].device
        tl = feats["f1_16"].shape[0]                      # frames on this rank (== t without sharding)
        if shard is None and tl != t:
            raise NotImplementedError("cascade: batch size 1 only (frames == t)")
        if t == 1:
            warnings.warn("PPMStereo with a single frame produces NaN disparities (reference behaviour, T must be >= 2)")
        lib = L.load()
        with torch.cuda.device(dev):
            prev, fo = None, None
            for s_, blk, ai, n_it, isc in ((16, self.update_block16, 0, iters // 2, 4), (8, self.update_block08, 1, iters // 2, 2),
                                          (4, self.update_block04, 2, iters, 1)):
                f1, f2 = feats[f"f1_{s_}"], feats[f"f2_{s_}"]
                h, w = f1.shape[2:]
                eng = blk.engine(tl, h, w, dev, shard)
                eng.set_inp(feats[f"inp_{s_}"])
                eng.set_net(feats[f"net_{s_}"])
                if prev is None:
                    eng.set_flow(self.zero_init(f1))                                                  # :231-236, :695
                    eng.set_mhs(None)
                else:
                    ph, pw = prev.h, prev.w
                    eng.set_flow(bilinear(fo, (h, w), True, -(h / fo.shape[2])))                      # :724-725, :760-761 (sign flip kept)
                    eng.parity, eng.have_mhs = 0, True                                                # :726-727, :763-764: mhs x2
                    L.check(lib.ppms_sp_resize_blend(prev.mhs_view(), eng.mhs_view(), tl, ph, pw, 2 * ph, 2 * pw, 0.0, 1.0, L.stream_ptr()))
                    # :729-732, :765-767: net = (net_s + interp(net_2s)) / 2
                    L.check(lib.ppms_sp_resize_blend(prev.net_view(), eng.net_view(), tl, ph, pw, 2 * ph, 2 * pw, 0.5, 0.5, L.stream_ptr()))
                eng.begin(CorrBlock1D(f1, f2).levels, self.att[ai].packed(dev))
                fo = _run_iterations(eng, n_it, isc, tl, h, w, preds, uncs, "all" if not test_mode else ("last" if s_ == 4 else "none"))
                prev = eng
            return preds[-1], uncs[-1]


def window_plan(num_ims: int, kernel_size: int = 20):
    """Sliding-window schedule of PPMStereo.forward_batch_test (ppmstereo.py:242-310): list of
    (start, stop, keep_from, keep_to) with keep_* window-local.  Trailing windows whose output the reference
    discards (:296) are not scheduled at all."""
    stride = kernel_size // 2
    if kernel_size > num_ims:
        return [(0, num_ims, 0, num_ims)]
    plan = []
    for i in range(0, num_ims, stride):
        n = min(i + kernel_size, num_ims) - i
        if plan and n >= stride:
            plan.append((i, i + n, stride // 2, n if n < kernel_size else n + (-stride // 2)))
        elif not plan:
            plan.append((i, i + n, 0, n + (-stride // 2)))
    return plan


def shard_windows(plan, rank: int, world: int):
    """Window-level sharding across GPUs (SURVEY.md section 8e level 1): windows are independent units; rank r takes
    windows r, r+world, ...  No data-path collective; the kept disparities are gathered once at the end."""
    return [w for i, w in enumerate(plan) if i % world == rank]
