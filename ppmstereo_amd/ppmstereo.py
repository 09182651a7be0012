"""The hot loop with the reference's call signature, and the thin caller-side glue needed to drive it.

``forward_update_block`` keeps the signature and side effects of
``PPMStereo.forward_update_block`` (/root/reference/models/core/ppmstereo.py:426-594): it can be bound as a method
of the reference's ``PPMStereo`` or used through ``PPMStereoHotPath`` below, which also holds
the three update blocks / q-k projections under the reference's attribute names (``update_block16/08/04``,
``att``) so a reference checkpoint loads with ``strict=False``.
"""
from __future__ import annotations

import math
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
        # the upsampled flow of an iteration is consumed by its prediction (when emitted) and, after the last iteration, by the next
        # scale's initialisation (:724-725): anywhere else the mask head + convex upsampling are dead work and are not launched
        flow_out = eng.iterate(need_up=(emit == "all" or it + 1 == iters))
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


def convex_upsample_3d(flow: torch.Tensor, mask: torch.Tensor, rate: int, T: int) -> torch.Tensor:
    """PPMStereo.convex_upsample_3d, ppmstereo.py:199-228 (NCHW in, NCHW out; one window of T frames, batch 1)."""
    if rate != 4:
        raise NotImplementedError("convex_upsample_3d: rate 4 only")
    L.require_gpu(flow, mask)
    N, _, H, W = flow.shape
    if N != T:
        raise NotImplementedError("convex_upsample_3d: batch size 1 (N == T)")
    lib, s = L.load(), L.stream_ptr()
    f = torch.empty(N * H * W, 2, dtype=torch.float32, device=flow.device)
    m = torch.empty(N * H * W, 432, dtype=torch.float32, device=flow.device)
    fc, mc = flow.contiguous().float(), mask.contiguous().float()
    L.check(lib.ppms_nchw_to_nhwc(fc.data_ptr(), f.data_ptr(), 2, N, 2, H * W, s))
    L.check(lib.ppms_nchw_to_nhwc(mc.data_ptr(), m.data_ptr(), 432, N, 432, H * W, s))
    out = torch.empty(N, 2, 4 * H, 4 * W, dtype=torch.float32, device=flow.device)
    L.check(lib.ppms_convex_upsample_3d(f.data_ptr(), m.data_ptr(), 432, out.data_ptr(), T, H, W, 0, s))
    return out


def forward_update_block(self, image1, update_block: SequenceUpdateBlock3D, corr_fn: CorrBlock1D, flow: torch.Tensor, net: torch.Tensor,
                         inp: torch.Tensor, motion_hidden_state: Optional[torch.Tensor], attn_block: Attention_qk, predictions: List,
                         uncertainties: List, iters: int, interp_scale: float, t: int):
    """Same contract as PPMStereo.forward_update_block (ppmstereo.py:426-594): runs ``iters`` refinement
    iterations at one scale, appends one full-resolution prediction and uncertainty per iteration and returns
    (flow_out (BT,2,4h,4w), net (BT,128,h,w), motion_hidden_state (BT,64,h,w)).  ``image1`` is unused (as in the
    reference).  BT = b * t: batch elements are independent except for the normaliser of the frame scores (:533), see
    ``_forward_update_block_batched``; b = 1 (inference) is the device-resident fast path."""
    L.require_gpu(flow, net, inp)
    bt, c, h, w = inp.shape
    if bt % t:
        raise RuntimeError(f"forward_update_block: {bt} frames do not divide into clips of t = {t}")
    if bt != t:
        return _forward_update_block_batched(update_block, corr_fn, flow, net, inp, motion_hidden_state, attn_block, predictions, uncertainties,
                                             iters, interp_scale, t)
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


def _forward_update_block_batched(update_block, corr_fn, flow, net, inp, motion_hidden_state, attn_block, predictions, uncertainties, iters,
                                  interp_scale, t: int):
    """forward_update_block for b > 1 clips in one call (ppmstereo.py:443-449: frame index = bi * t + ti).  Everything is per batch element
    -- the 3-D convolutions, the frame similarity, the top-5 pick and the attention all see one clip -- except ONE scalar per clip index:
    ``selected_score.mean()`` at :533 averages the picked frames' scores over the batch as well, so the key modulation s_hat of clip i is
    its score divided by the mean over all b elements.  One engine per element; the stages run element by element, the normaliser is
    fixed between the pick and the attention."""
    bt, c, h, w = inp.shape
    b = bt // t
    if c != 128 or int(interp_scale) not in (1, 2, 4):
        raise RuntimeError("forward_update_block: 128 context channels and interp_scale 4, 2 or 1 expected")
    if t == 1:
        warnings.warn("PPMStereo with a single frame produces NaN disparities (reference behaviour, T must be >= 2)")
    isc, dev = int(interp_scale), inp.device
    rows = t * h * w
    with torch.cuda.device(dev):
        engs = []
        for bi in range(b):
            sl = slice(bi * t, (bi + 1) * t)
            eng = update_block.engine(t, h, w, dev, slot=bi)
            eng.set_inp(inp[sl]), eng.set_net(net[sl]), eng.set_flow(flow[sl])
            eng.set_mhs(None if motion_hidden_state is None else motion_hidden_state[sl])
            eng.begin([lv[bi * rows:(bi + 1) * rows] for lv in corr_fn.levels], attn_block.packed(dev))      # the pyramid is per frame: row slices
            engs.append(eng)
        k = engs[0].ksel
        flow_out = None
        for _ in range(iters):
            for eng in engs:
                eng.lookup(), eng.motion_and_value(), eng.uncertainty(), eng.pick()
            picked = [eng.SCORE.gather(1, eng.SEL[:, :k].long()) for eng in engs]                            # (t, k) scores of the picked frames
            mean_all = torch.stack([x.sum(1) for x in picked]).sum(0) / float(b * k)                           # :533: the mean runs over the batch too
            outs, uncs = [], []
            for eng, x in zip(engs, picked):
                eng.SHAT[:, :k] = x / mean_all[:, None]
                eng.attend(), eng.update(need_mask=True)
                outs.append(eng.upsample().clone())
                uncs.append(eng.UNC_local().view(t, 1, h, w).clone())
            flow_out = torch.cat(outs)
            unc = torch.cat(uncs)
            uncertainties.append(bilinear(unc, (4 * isc * h, 4 * isc * w), False))
            predictions.append(bilinear(flow_out[:, :1], (isc * 4 * h, isc * 4 * w), True, float(isc)) if isc > 1 else flow_out[:, :1].clone())
        return flow_out, torch.cat([e.get_net() for e in engs]), torch.cat([e.get_mhs() for e in engs])


class ClipPipeline:
    """Software pipeline over CONSECUTIVE clips / sliding windows (independent units, ppmstereo.py:277-307): the 1/16 and 1/8 scales of
    a clip are latency bound (~70 small launches per iteration, a quarter of a clip's time for 7 % of its FLOPs) and leave most of the
    chip idle, the 1/4 scale is throughput bound.  With the scales on two HIP streams -- ``small`` (1/16, 1/8) and ``large`` (1/4) --
    the small scales of clip k + 1 run under the 1/4 scale of clip k.  Results are the same bits as the unpipelined cascade; what is
    shared between the two stages of consecutive clips (the 1/8 engine's state, read by the 1/4 scale's prologue) is guarded by events.

        pipe = ClipPipeline(device)
        for feats in clips:
            disp, unc = model.cascade(feats, iters, T, test_mode=True, pipeline=pipe)     # enqueues; the result is valid after ...
            ...
        pipe.wait()                                                                        # ... the caller's stream has waited here

    ``cascade`` returns tensors produced on ``large``; ``wait()`` makes the current stream wait for everything enqueued so far (no host
    synchronisation), ``wait(handle)`` for one clip (``handle`` = ``pipe.last``: the completion event of the clip just enqueued)."""

    def __init__(self, device=None, small_priority: int = 0):
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        self.device = device
        self.small = torch.cuda.Stream(device=device, priority=small_priority)
        self.large = torch.cuda.Stream(device=device)
        self.consumed: Optional[torch.cuda.Event] = None      # the previous clip's 1/4-scale prologue has read the 1/8 engine's state
        self.last: Optional[torch.cuda.Event] = None          # completion of the clip enqueued last
        self.serial = False                                   # True: no overlap between clips (per-launch timing runs of bench.py)
        self.done_events: List[torch.cuda.Event] = []         # (timing: one per clip when ``record_done`` is set)
        self.record_done = False

        self._results: List[torch.Tensor] = []                # tensors handed out by cascade() since the last wait()

    def wait(self, handle: Optional[torch.cuda.Event] = None):
        """Orders the current stream behind the clips enqueued so far AND tells the caching allocator that the tensors ``cascade`` returned
        (allocated on ``large``) are now used on the current stream: a caller may launch asynchronous kernels on them and drop them
        without the block being handed to the next clip's 1/4 scale while such a kernel still reads it."""
        ev = self.last if handle is None else handle
        cur = torch.cuda.current_stream(self.device)
        if ev is not None:
            cur.wait_event(ev)
        for t in self._results:
            t.record_stream(cur)
        self._results.clear()


class PPMStereoHotPath(nn.Module):
    """The part of PPMStereo that lives on the hot path (ppmstereo.py:82-117 modules, :426-594 loop, :696-804
    cascade), from the encoder / SST outputs on.  Attribute names follow the reference."""

    def __init__(self, max_disp: int = 192, mixed_precision: bool = False, num_frames: int = 5,
                 attention_type: Optional[str] = "self_stereo_temporal_update_time_update_space", use_3d_update_block: bool = True,
                 different_update_blocks: bool = True, use_convex_3d: bool = False, init_flow: bool = False):
        super().__init__()
        if not (use_3d_update_block and different_update_blocks) or init_flow:
            raise NotImplementedError("supported configurations: models/ppm_stereo_model.py:27-33 (use_3d_update_block=True, "
                                      "different_update_blocks=True, init_flow=False) with use_convex_3d False or True (train.py / test.py default)")
        self.hidden_dim = self.context_dim = 128
        self.mixed_precision = mixed_precision      # the engine's precision is fixed: fp32-accurate convs, bf16 attention
        self.use_convex_3d = bool(use_convex_3d)
        self.num_frames = num_frames
        self.att = nn.ModuleList([Attention_qk(num_heads=1, dim_head=128) for _ in range(3)])
        c3 = self.use_convex_3d
        self.update_block08 = SequenceUpdateBlock3D(hidden_dim=128, cor_planes=36, mask_size=4, use_convex_3d=c3)
        self.update_block16 = SequenceUpdateBlock3D(hidden_dim=128, cor_planes=36, mask_size=4, use_convex_3d=c3, attention_type=attention_type)
        self.update_block04 = SequenceUpdateBlock3D(hidden_dim=128, cor_planes=36, mask_size=4, use_convex_3d=c3)

    forward_update_block = forward_update_block

    def convex_upsample(self, flow, mask, rate: int = 4):
        return convex_upsample(flow, mask, rate)

    def convex_upsample_3d(self, flow, mask, rate: int, T: int):
        """PPMStereo.convex_upsample_3d, ppmstereo.py:199-228: flow (b*T,2,H,W), mask (b*T,432,H,W) -> (b*T,2,4H,4W); b = 1."""
        return convex_upsample_3d(flow, mask, rate, T)

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
                uncertainties: Optional[list] = None, shard=None, test_mode: bool = False, pipeline: Optional[ClipPipeline] = None):
        """The 1/16 -> 1/8 -> 1/4 cascade of PPMStereo.forward (ppmstereo.py:696-804), device resident: the state handed from
        scale to scale (hidden state, motion hidden state) stays in the engines' SP buffers (ppms_sp_resize_blend), only the
        2-channel flow passes through an NCHW resize.  feats: f1_s, f2_s, net_s, inp_s for s in (16, 8, 4) on the GPU
        (with ``shard``: this rank's frames only).  test_mode: only the final prediction is produced (ppmstereo.py:801-804).
        pipeline: a ``ClipPipeline`` -- the small scales and the 1/4 scale are enqueued on its two streams so that consecutive clips
        overlap (same results; see the class).
        Returns (flow_up (T,1,H,W), uncertainty (T,1,H,W)) = predictions[-1], uncertainties[-1]."""
        if iters < 2:
            raise ValueError(f"cascade: iters={iters}; the 1/16 and 1/8 scales run iters // 2 iterations each (ppmstereo.py:708,744) and need at least one")
        preds = [] if predictions is None else predictions
        uncs = [] if uncertainties is None else uncertainties
        dev = feats["f1_16"].device
        tl = feats["f1_16"].shape[0]                      # frames on this rank (== t without sharding)
        if shard is None and tl != t:
            if tl % t or pipeline is not None:
                raise RuntimeError(f"cascade: {tl} frames do not divide into clips of t = {t} (or a pipeline was given for a batch)")
            return self._cascade_batched(feats, iters, t, preds, uncs)
        if t == 1:
            warnings.warn("PPMStereo with a single frame produces NaN disparities (reference behaviour, T must be >= 2)")
        lib = L.load()
        with torch.cuda.device(dev):
            caller = torch.cuda.current_stream()
            if pipeline is not None:
                pipeline.small.wait_stream(caller)        # the inputs were produced on the caller's stream
                if pipeline.serial and pipeline.last is not None:
                    pipeline.small.wait_event(pipeline.last)
            prev, fo = None, None
            for s_, blk, ai, n_it, isc in ((16, self.update_block16, 0, iters // 2, 4), (8, self.update_block08, 1, iters // 2, 2),
                                          (4, self.update_block04, 2, iters, 1)):
                f1, f2 = feats[f"f1_{s_}"], feats[f"f2_{s_}"]
                h, w = f1.shape[2:]
                stream = caller if pipeline is None else (pipeline.large if s_ == 4 else pipeline.small)
                if pipeline is not None:
                    if s_ == 8 and pipeline.consumed is not None:
                        stream.wait_event(pipeline.consumed)      # the previous clip's 1/4 scale still reads this engine's state in its prologue
                    if s_ == 4:
                        stream.wait_stream(pipeline.small)        # this clip's 1/16 and 1/8 scales
                    for k in (f"f1_{s_}", f"f2_{s_}", f"net_{s_}", f"inp_{s_}"):
                        feats[k].record_stream(stream)            # (allocated on the caller's stream: keep the allocator from recycling them early)
                with torch.cuda.stream(stream):
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
                    if pipeline is not None and s_ == 4:
                        pipeline.consumed = torch.cuda.Event()
                        pipeline.consumed.record()                # the 1/8 engine's flow / hidden states have been read: the next clip may overwrite them
                    eng.begin(CorrBlock1D(f1, f2).levels, self.att[ai].packed(dev))
                    fo = _run_iterations(eng, n_it, isc, tl, h, w, preds, uncs, "all" if not test_mode else ("last" if s_ == 4 else "none"))
                    prev = eng
            if pipeline is not None:
                pipeline.last = torch.cuda.Event(enable_timing=pipeline.record_done)
                pipeline.last.record(pipeline.large)
                if pipeline.record_done:
                    pipeline.done_events.append(pipeline.last)
                pipeline._results += [preds[-1], uncs[-1]]         # (ClipPipeline.wait records the consuming stream on them)
                del pipeline._results[:-16]                        # a caller that never waits (bench.py) must not accumulate references
            return preds[-1], uncs[-1]


def _cascade_batched(self, feats, iters: int, t: int, preds: list, uncs: list):
    """cascade for b > 1 clips (frame index = bi * t + ti, ppmstereo.py:443-449): the reference's glue between the three
    forward_update_block calls (:696-791) on NCHW tensors -- every iteration's prediction is produced, as test_mode=False does.  The batch
    elements meet in one scalar per clip index only (the mean of the picked scores, :533), which forward_update_block reproduces; b = 1 --
    the reference's inference entry -- takes the device-resident path above instead."""
    f16 = feats["f1_16"]
    fo, net16, mhs16 = self.forward_update_block(None, self.update_block16, CorrBlock1D(f16, feats["f2_16"]), self.zero_init(f16), feats["net_16"],
                                                 feats["inp_16"], None, self.att[0], preds, uncs, iters // 2, 4, t)                 # :707-722
    nets, mhs = {16: net16}, {16: mhs16}
    for s_, blk, ai, n_it, isc in ((8, self.update_block08, 1, iters // 2, 2), (4, self.update_block04, 2, iters, 1)):
        f1, f2 = feats[f"f1_{s_}"], feats[f"f2_{s_}"]
        h, w = f1.shape[2:]
        flow = bilinear(fo, (h, w), True, -(h / fo.shape[2]))                                                                       # :724-725, :760-761
        m_up = bilinear(mhs[2 * s_], (h, w), True)                                                                                   # :726-727, :763-764
        net = (feats[f"net_{s_}"] + bilinear(nets[2 * s_], (h, w), True)) / 2.0                                                     # :729-732, :765-767
        fo, nets[s_], mhs[s_] = self.forward_update_block(None, blk, CorrBlock1D(f1, f2), flow, net, feats[f"inp_{s_}"], m_up, self.att[ai],
                                                          preds, uncs, n_it, isc, t)                                                # :743-758, :776-791
    return preds[-1], uncs[-1]


PPMStereoHotPath._cascade_batched = _cascade_batched


def position_encoding_sine(d_model: int, h: int, w: int) -> torch.Tensor:
    """PositionEncodingSine (temp_bug_fix=True), models/core/attention.py:23-64 -> (d_model, h, w); host table, same torch
    op sequence as the reference (built once per geometry)."""
    pe = torch.zeros((d_model, h, w))
    y_position = torch.ones((h, w)).cumsum(0).float().unsqueeze(0)
    x_position = torch.ones((h, w)).cumsum(1).float().unsqueeze(0)
    div_term = torch.exp(torch.arange(0, d_model // 2, 2).float() * (-math.log(10000.0) / (d_model // 2)))[:, None, None]
    pe[0::4, :, :] = torch.sin(x_position * div_term)
    pe[1::4, :, :] = torch.cos(x_position * div_term)
    pe[2::4, :, :] = torch.sin(y_position * div_term)
    pe[3::4, :, :] = torch.cos(y_position * div_term)
    return pe


class InputPadder:
    """Replicate-pads the last two dimensions up to multiples of ``divis_by`` and crops results back (the reference's helper,
    models/core/utils/utils.py:19-44).  "sintel" mode centres the image (the extra row / column of an odd pad goes to the bottom /
    right); any other mode pads the height at the bottom only."""

    def __init__(self, dims, mode: str = "sintel", divis_by: int = 8):
        self.ht, self.wd = int(dims[-2]), int(dims[-1])
        extra_h, extra_w = -self.ht % divis_by, -self.wd % divis_by
        left, top = extra_w // 2, (extra_h // 2 if mode == "sintel" else 0)
        self._pad = [left, extra_w - left, top, extra_h - top]          # F.pad order: left, right, top, bottom

    def pad(self, *inputs):
        for x in inputs:
            if x.ndim != 4:
                raise ValueError(f"InputPadder.pad: 4-D tensors expected, got {tuple(x.shape)}")
        if not any(self._pad):
            return list(inputs)                                           # (already a multiple: nothing to copy)
        return [torch.nn.functional.pad(x, self._pad, mode="replicate") for x in inputs]

    def unpad(self, x):
        if x.ndim != 4:
            raise ValueError(f"InputPadder.unpad: 4-D tensor expected, got {tuple(x.shape)}")
        left, right, top, bottom = self._pad
        return x[..., top:x.shape[-2] - bottom, left:x.shape[-1] - right]


class PPMStereo(PPMStereoHotPath):
    """``models/core/ppmstereo.py:PPMStereo`` from the encoder outputs on: same constructor arguments, ``forward`` (:601-804)
    and ``forward_batch_test`` (:238-320).  The encoders are outside the hot path: ``fnet``
    and ``cnet`` default to this package's HIP ``BasicEncoder`` / ``Feature("tiny", 256)`` (``ppmstereo_amd/encoder.py``, ``cnet.py``:
    rows f3, f5; ``False`` leaves one unset), or take any callable with the same contract:
    ``fnet([im1, im2]) -> (fmap1, fmap2)`` (BT,256,H/4,W/4), ``cnet(im1) -> (c4, c8, c16)`` with 256 channels;
    ``sst`` stands for ``forward_sst_block`` (:322-395): "auto" (default) follows the reference's ctor (:139-171) -- the HIP
    ``SSTBlock`` (``ppmstereo_amd/sst.py``, row f4) when ``attention_type`` names "self_stereo" / "temporal", its parameters
    registered under the reference's names (``time_embed``, ``time_attn_blocks``, ``self_attn_blocks``, ``cross_attn_blocks``);
    ``None`` = the ``attention_type=None`` behaviour (positional encoding only); or any callable ``(f1_16, f2_16, T)``.
    Everything between the images (minus cnet) and the returned disparity runs on the gfx950 kernels."""

    # what models/ppm_stereo_model.py:27-33 passes: the configuration of the released model, and the one this implementation serves
    WRAPPER_CONFIG = dict(mixed_precision=True, num_frames=5, attention_type="self_stereo_temporal_update_time_update_space",
                          use_3d_update_block=True, different_update_blocks=True)

    @classmethod
    def shipped(cls, **overrides):
        """PPMStereo(**WRAPPER_CONFIG, **overrides): the model as the reference's wrapper builds it (models/ppm_stereo_model.py:27-33)."""
        return cls(**{**cls.WRAPPER_CONFIG, **overrides})

    def __init__(self, max_disp: int = 192, mixed_precision: bool = False, num_frames: int = 5, attention_type: Optional[str] = None,
                 use_3d_update_block: bool = False, different_update_blocks: bool = False, use_convex_3d: bool = False, init_flow: bool = False,
                 *, fnet=None, cnet=None, sst="auto"):
        """Parameter names, order AND defaults of the reference's constructor (ppmstereo.py:45-55).  The reference's defaults select the 2-D
        update block (use_3d_update_block=False), which this implementation does not serve (and the reference's own forward cannot drive,
        SURVEY.md hazard 7): a default-constructed PPMStereo() therefore raises NotImplementedError naming the supported configuration
        instead of silently building another module tree -- pass the wrapper's arguments, or use ``PPMStereo.shipped()``."""
        if not (use_3d_update_block and different_update_blocks):
            raise NotImplementedError("PPMStereo: the gfx950 implementation serves the released configuration, models/ppm_stereo_model.py:27-33 -- "
                                      "use_3d_update_block=True, different_update_blocks=True (PPMStereo.shipped() / PPMStereo.WRAPPER_CONFIG); the "
                                      "2-D update block selected by the reference's constructor defaults is outside the hot path")
        super().__init__(max_disp, mixed_precision, num_frames, attention_type, use_3d_update_block, different_update_blocks, use_convex_3d, init_flow)
        at = attention_type
        if isinstance(sst, str) and sst == "auto":
            sst = None
            if at is not None and ("self_stereo" in at or "temporal" in at):
                if not ("self_stereo" in at and "temporal" in at):
                    raise NotImplementedError("SST block: attention types with both 'self_stereo' and 'temporal' (the shipped model) or neither")
                from .sst import SSTBlock
                blk = SSTBlock(dim=256, num_frames=self.num_frames)
                self.time_embed = blk.time_embed                       # same objects, registered under the reference's names
                self.time_attn_blocks, self.self_attn_blocks, self.cross_attn_blocks = blk.time_attn_blocks, blk.self_attn_blocks, blk.cross_attn_blocks
                object.__setattr__(self, "_sst_impl", blk)           # (not a second registration of the same parameters)
                sst = blk
        if fnet is None:                                       # the reference builds it in its ctor (ppmstereo.py:64)
            from .encoder import BasicEncoder
            fnet = BasicEncoder(output_dim=256, norm_fn="instance")
        if cnet is None:                                       # (ppmstereo.py:69; no checkpoint is read here: load_state_dict supplies it)
            from .cnet import Feature
            cnet = Feature(model_name="tiny", output_dim=256)
        self.fnet, self.cnet = (None if fnet is False else fnet), (None if cnet is False else cnet)
        object.__setattr__(self, "sst", sst)
        # state_dict in the reference's registration order (ppmstereo.py:64-171): fnet, cnet, att, the update blocks, the SST modules
        ref_order = ["fnet", "cnet", "att", "update_block08", "update_block16", "update_block04", "time_attn_blocks", "self_attn_blocks", "cross_attn_blocks"]
        mods = self._modules
        for k in [k for k in ref_order if k in mods] + [k for k in list(mods) if k not in ref_order]:
            mods[k] = mods.pop(k)                              # (re-insertion moves the key to the end)
        self.dim = 256
        self._pe_cache: Dict[tuple, torch.Tensor] = {}
        self.parallel_encoders = True                          # cnet on a side stream beside fnet (forward)
        self._enc_streams: Dict[int, torch.cuda.Stream] = {}

    def _encoder_stream(self, device) -> torch.cuda.Stream:
        idx = torch.device(device).index or 0
        if idx not in self._enc_streams:
            self._enc_streams[idx] = torch.cuda.Stream(device=device)
        return self._enc_streams[idx]

    def load_state_dict(self, sd, strict: bool = True, **kw):
        r = super().load_state_dict(sd, strict=strict, **kw)
        for m in (self.fnet, self.cnet, getattr(self, "_sst_impl", None)):       # packed weight copies are cached per module
            if hasattr(m, "invalidate"):
                m.invalidate()
        return r

    # ------------------------------------------------------------------ pre-loop glue (ppmstereo.py:620-682)
    def _pe(self, h: int, w: int, device) -> torch.Tensor:
        key = (h, w, str(device))
        if key not in self._pe_cache:
            self._pe_cache[key] = position_encoding_sine(self.dim, h, w).to(device).contiguous()
        return self._pe_cache[key]

    def pre_loop(self, fmap1: torch.Tensor, fmap2: torch.Tensor, c4: torch.Tensor, c8: torch.Tensor, c16: torch.Tensor, t: int, ctx_ready=None):
        """fmap (BT,256,h,w) at 1/4, context features at 1/4, 1/8, 1/16 -> the dict ``cascade`` consumes.
        ctx_ready (optional callable): called once, right before the context features are first read -- ``forward`` passes the wait for the
        stream cnet runs on, so that the pooling and the SST block (which need fnet's output only) do not wait for cnet."""
        L.require_gpu(fmap1, fmap2, c4, c8, c16)
        lib, st = L.load(), L.stream_ptr
        N, C, h, w = fmap1.shape
        if C != 256 or h % 4 or w % 4:
            raise RuntimeError("pre_loop: 256-channel 1/4-resolution features with h, w multiples of 4 expected")
        dev = fmap1.device
        f32 = lambda *s: torch.empty(*s, dtype=torch.float32, device=dev)
        fm = [fmap1.contiguous().float(), fmap2.contiguous().float()]
        feats: Dict[str, torch.Tensor] = {"f1_4": fm[0], "f2_4": fm[1]}

        def mix(f, c, tag):                                   # net = tanh(avg), inp = relu(avg) of the two 128-channel halves
            c = c.contiguous().float()
            net, inp = f32(N, 128, f.shape[2], f.shape[3]), f32(N, 128, f.shape[2], f.shape[3])
            L.check(lib.ppms_ctx_mix(f.data_ptr(), c.data_ptr(), net.data_ptr(), inp.data_ptr(), N, f.shape[2] * f.shape[3], st()))
            feats["net_" + tag], feats["inp_" + tag] = net, inp

        h16, w16, h8, w8 = h // 4, w // 4, h // 2, w // 2
        f16 = []
        for f in fm:                                           # :649-652 avg_pool 4x4, then the SST block
            p = f32(N, C, h16, w16)
            L.check(lib.ppms_avgpool(f.data_ptr(), p.data_ptr(), N * C, h, w, 4, st()))
            f16.append(p)
        if self.sst is not None:
            f16 = list(self.sst(f16[0], f16[1], t))
        else:                                                  # forward_sst_block with attention_type=None: + positional encoding
            pe = self._pe(h16, w16, dev)
            for p in f16:
                L.check(lib.ppms_axpby(p.data_ptr(), pe.data_ptr(), p.data_ptr(), 1.0, 1.0, pe.numel(), p.numel(), st()))
        feats["f1_16"], feats["f2_16"] = f16
        if ctx_ready is not None:
            ctx_ready()
        mix(fm[0], c4, "4")
        mix(f16[0], c16, "16")
        for i, f in enumerate(fm):                             # :666-671 (avg_pool2 + interp(1/16)) / 2
            p = f32(N, C, h8, w8)
            L.check(lib.ppms_avgpool(f.data_ptr(), p.data_ptr(), N * C, h, w, 2, st()))
            q = bilinear(f16[i], (h8, w8), True)
            L.check(lib.ppms_axpby(p.data_ptr(), q.data_ptr(), p.data_ptr(), 0.5, 0.5, p.numel(), p.numel(), st()))
            feats[f"f{i + 1}_8"] = p
        mix(feats["f1_8"], c8, "8")
        return feats

    @torch.no_grad()
    def forward(self, image1: torch.Tensor, image2: torch.Tensor, flow_init=None, iters: int = 10, test_mode: bool = False, pipeline=None):
        """PPMStereo.forward (ppmstereo.py:601-804): image (b, T, 3, H, W) in [0, 255], H, W multiples of 32 (b = 1: the device-resident
        cascade; b > 1: the reference's glue around the batched forward_update_block).
        test_mode: (flow_up, uncertainty), each (b, T, 1, H, W); else (predictions (D, b, T, 1, H, W), uncertainties).
        pipeline (test_mode only): a ``ClipPipeline`` -- the result is valid once ``pipeline.wait()`` has been called."""
        if flow_init is not None:
            raise NotImplementedError("flow_init: the reference's own path for it reads undefined state (ppmstereo.py:691-693, 763)")
        if self.fnet is None or self.cnet is None:
            raise RuntimeError("PPMStereo.forward needs the encoders: pass fnet= / cnet= (outside the hot path, SURVEY.md section 8 f3-f5)")
        b, T, c, h, w = image1.shape
        if b != 1 and pipeline is not None:
            raise NotImplementedError("PPMStereo.forward: a ClipPipeline overlaps consecutive batch-1 clips")
        with torch.cuda.device(image1.device):
            im1 = (2 * (image1 / 255.0) - 1.0).contiguous().reshape(b * T, c, h, w)
            im2 = (2 * (image2 / 255.0) - 1.0).contiguous().reshape(b * T, c, h, w)
            # fnet (both views) and cnet (left view) depend on the images only and are chains of small launches that leave most of the chip
            # idle: cnet runs on a second stream beside fnet (whole call -3.5 ms at config 2); its outputs are handed to the caller's
            # stream with an event + record_stream
            cur = torch.cuda.current_stream(image1.device)
            if self.parallel_encoders:
                side = self._encoder_stream(image1.device)
                side.wait_stream(cur)
                im1.record_stream(side)
                with torch.cuda.stream(side):
                    c4, c8, c16 = self.cnet(im1)
                fmap1, fmap2 = self.fnet([im1, im2])
                for t_ in (c4, c8, c16):
                    if torch.is_tensor(t_):
                        t_.record_stream(cur)
                ready = {"done": False}

                def ctx_ready():                               # the pooling and the SST block run before this: they need fnet's output only
                    if not ready["done"]:
                        cur.wait_stream(side)
                        ready["done"] = True
            else:
                fmap1, fmap2 = self.fnet([im1, im2])
                c4, c8, c16 = self.cnet(im1)
                ctx_ready = None
            if b == 1:
                feats = self.pre_loop(fmap1, fmap2, c4, c8, c16, T, ctx_ready)
            else:                                              # the glue in front of the loop is per clip (the SST block's time attention sees T frames)
                if ctx_ready is not None:
                    ctx_ready()
                per = [self.pre_loop(*(x[bi * T:(bi + 1) * T] for x in (fmap1, fmap2, c4, c8, c16)), T) for bi in range(b)]
                feats = {k: torch.cat([p_[k] for p_ in per]) for k in per[0]}
            preds, uncs = [], []
            self.cascade(feats, iters, T, preds, uncs, test_mode=test_mode, pipeline=pipeline if test_mode else None)
            if test_mode:
                return preds[-1].reshape(b, T, 1, h, w), uncs[-1].reshape(b, T, 1, h, w)
            return torch.stack(preds).reshape(-1, b, T, 1, h, w), torch.stack(uncs).reshape(-1, b, T, 1, h, w)

    @torch.no_grad()
    def forward_batch_test(self, batch_dict: Dict, kernel_size: int = 20, iters: int = 20, device=None, shard_ranks: bool = False):
        """PPMStereo.forward_batch_test (ppmstereo.py:238-320): batch_dict["stereo_video"] (N, 2, 3, H, W) on the host;
        per window: InputPadder(divis_by=32), one host->device copy, forward(test_mode=True), unpad, one device->host copy;
        windows of ``kernel_size`` frames every ``kernel_size // 2``, centre frames kept (:296-307).  Windows whose output the
        reference computes and then drops are not run.  Returns {"disparity", "uncertainties"}: (N, 1, H, W) CPU tensors.
        shard_ranks: under torch.distributed the windows are dealt round-robin over the ranks (independent units, no data-path
        collective) and the kept frames are gathered once at the end (``dist.gather_kept_frames``); every rank returns the video."""
        video = batch_dict["stereo_video"]
        num_ims = len(video)
        dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        disp_preds, uncertainties = [], []
        plan = window_plan(num_ims, kernel_size)
        if shard_ranks and torch.distributed.is_available() and torch.distributed.is_initialized():
            from . import dist as D
            rank, world = torch.distributed.get_rank(), torch.distributed.get_world_size()
            mine_d, mine_u, first = [], [], 0
            firsts = []
            for start, stop, keep_from, keep_to in plan:          # first kept frame of every window in video coordinates
                firsts.append(start + keep_from)
            for wi, (start, stop, keep_from, keep_to) in enumerate(plan):
                if wi % world != rank:
                    continue
                win = video[start:stop].to(dev)                              # one host -> device copy per window (see below)
                left, right = win[:, 0], win[:, 1]
                padder = InputPadder(left.shape, divis_by=32)
                left, right = padder.pad(left, right)
                d, u = self.forward(left[None], right[None], iters=iters, test_mode=True)
                d, u = padder.unpad(d[0]), padder.unpad(u[0])               # (T, 1, H0, W0)
                mine_d.append((firsts[wi], d[keep_from:keep_to].abs()[:, :1]))
                mine_u.append((firsts[wi], u[keep_from:keep_to].abs()[:, :1]))
            H0, W0 = video.shape[-2:]
            disp = D.gather_kept_frames(mine_d, num_ims, H0, W0)
            unc = D.gather_kept_frames(mine_u, num_ims, H0, W0)
            return {"disparity": disp.cpu(), "uncertainties": unc.cpu()}
        # several windows: independent units -> software pipeline (ClipPipeline): window k + 1's encoders and small scales are enqueued
        # before window k's result is collected, and run under window k's 1/4 scale
        pipe = ClipPipeline(dev) if len(plan) > 1 else None
        pending = None

        def collect(item):
            d, u, handle, padder, keep_from, keep_to = item
            if pipe is not None:
                pipe.wait(handle)
            # device -> host.  The stream is synchronised FIRST (a spinning wait): a pageable copy issued while the clip's ~900 launches are
            # still queued blocks inside the runtime on an interrupt-driven wait, and on a loaded host (the GPU boxes: load average 50-60)
            # the thread was rescheduled 50-150 ms late in every other call (whole call 50 / 100 / 195 ms alternating; with the
            # synchronisation in front the copy finds an idle stream: 49-50 ms every time, tools/whole_call_probe.py)
            du = torch.cat([padder.unpad(d[0])[:, None], padder.unpad(u[0])[:, None]])                        # one copy for both results
            torch.cuda.current_stream(du.device).synchronize()
            du = du.cpu()
            d, u = du[:du.shape[0] // 2], du[du.shape[0] // 2:]
            disp_preds.append(d[keep_from:keep_to])
            uncertainties.append(u[keep_from:keep_to])

        with torch.cuda.device(dev):
            for start, stop, keep_from, keep_to in plan:
                # host -> device: ONE copy of the window's contiguous (T, 2, 3, H, W) block; the two views are split and padded on the
                # device (slicing a view out on the host first costs a host-side copy of each view, padding there another one)
                win = video[start:stop].to(dev)
                left, right = win[:, 0], win[:, 1]
                padder = InputPadder(left.shape, divis_by=32)
                left, right = padder.pad(left, right)
                d, u = self.forward(left[None], right[None], iters=iters, test_mode=True, pipeline=pipe)
                item = (d, u, None if pipe is None else pipe.last, padder, keep_from, keep_to)
                if pending is not None:
                    collect(pending)
                pending = item
            collect(pending)
        return {"disparity": torch.cat(disp_preds).squeeze(1).abs()[:, :1], "uncertainties": torch.cat(uncertainties).squeeze(1).abs()[:, :1]}


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
