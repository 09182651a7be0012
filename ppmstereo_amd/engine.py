"""Device-resident engine for one scale of the hot loop (PPMStereo.forward_update_block,
/root/reference/models/core/ppmstereo.py:426-594).

Everything between two iterations stays in HBM in the kernels' own operand format (channel-last split-bf16 "SP"
planes, see include/ppms.h); the reference's NCHW tensors are converted once on the way in and once on the way
out.  All buffers are allocated once per (block, T, h, w) by the caller-side PyTorch allocator, every conv
descriptor is built once and uploaded once: an iteration is a fixed sequence of kernel launches on the current
stream with no host synchronisation, no allocation and no host<->device copy.

Stage methods mirror the reference calls one to one:
    lookup()                  corr_fn(flow)                               ppmstereo.py:489
    motion_and_value()        update_block.get_motion_and_value           :492   (ppmtereo_update.py:945-950)
    uncertainty()             update_block.get_uncertainty                :495   (ppmtereo_update.py:936-938)
    pick()                    QAM score / top-k / usage counter           :505-513
    attend()                  play + flash_attn_func + aggregation        :517-552
    update()                  update_block(net, inp, mf, mfg, t)          :569   (ppmtereo_update.py:971-1003)
    upsample()                convex_upsample + prediction resize         :576-591
"""
from __future__ import annotations

import contextlib
import ctypes as C
import math
from typing import Dict, List, Optional

import torch

from . import _lib as L
from . import packing as _packing

TOP_K = 5
# Which HIP kernel generation serves a convolution where more than one applies.  These are the measured-best settings (DESIGN.md
# section 5); tools/ab_switches.py maps PPMS_* environment variables onto this table for A/B runs on the GPU box -- the product itself
# reads no environment variable.  Every setting runs HIP kernels only.
TUNING = dict(
    conv5=True,           # one 8-wave workgroup per CU, 7- / 8-block tiles (conv_gemm5.hip) where it applies
    conv6=True,           # one wave per SIMD on the 16x16x32 MFMA, 16 x 13-pixel tiles (conv_gemm6.hip, round 5) where the library rates its fill >= 85 %
    conv6_stream=True,    # conv_gemm6's STREAM form for the (5,1,1) convs of the GRU's pass T (else conv_gemm5 / conv_gemm2)
    conv6_pad2x=False,    # conv_gemm6 also for convs whose couts fill only half of the padded rows (convf2: 64 of 128 -- 50 us instead of 63 + 117 us of K-sliced launch + reduce, but on the side stream it takes whole CUs from convc2: 35.5 vs 35.4 ms per clip)
    conv5_sliced=False,   # its K-sliced form on the 1/8, 1/16 maps: correct (tests) but slower than conv_gemm2's slicing there
    conv5_gemm=True,      # its GEMM mode for the 256-cout convs without a spatial sweep ((5,1,1) GRU pass, 1x1 heads)
    pwchain=True,         # fused per-pixel layer chains of the correlation encoder
    ysweep=True,          # conv_gemm2: one y-swept window per (dt, chunk) for (1, kh, 1) convs
    win2d=False,          # conv_gemm2: 2-D window for kh, kw > 1 (measured neutral to slower)
    slices=True,          # grid-level K slicing of the convs of small maps (1/16, 1/8 scales)
    hoist=True,           # iteration-invariant inp share of the GRU gates computed once per scale
    gemm1=True,           # thin-GEMM kernel (gemm1.hip) for the 1x1 convolutions / Linear layers it serves
    stream=True,          # register-streamed kernel (conv_stream.hip) where the library rates it faster (small maps: no K slices, no reduce launch);
                          # "all": wherever it serves a conv of a map of <= 16 384 pixels (A/B runs)
    stream_hint=0,        # its tile: 0 = the library chooses, 1 / 2 = 32- / 64-pixel tiles
    hid_exact=True,       # hoisted blocks: the GRU convs read [h | mf, hid] with weights (W_mf + W_mfg | beta W_mfg) instead of [h | mf, mfg]; hid is a
                          # bf16 tensor (all-zero lo plane), so the products with that plane are skipped (ppms_conv.lo_zero_from)
    convf2_unsliced=False,  # the flow encoder's 3x3 128 -> 64 conv without K slices on large maps (measured neutral: 40.3 / 40.2 ms per clip)
    conv5_m192=True,      # conv_gemm5's three-cout-block layout for the 190 / 192-cout convs (else padded to 256 rows)
    fork_min_pixels=0,    # independent branches of an iteration run on the side stream only on maps with at least this many pixels (0: always)
    conv6_grouped=True,   # the two 128 -> 128 (1,1,5) tails of convz1 / convr1 as ONE grouped conv_gemm6 launch (ppms_conv.groups = 2, M = 256 wave layout) where
                          # conv_gemm6 serves the map, instead of two M = 128 launches on two streams
    attn_p="fp16",        # format of the unnormalised probabilities P~ in the memory read-out's P~ V product (and of the V^T image the to_v conv writes):
                          # "fp16" = 11 significand bits, "bf16" = 8 (what flash-attention itself uses) at the same MFMA count.  The reference fixtures were
                          # generated with fp32 P (tools/gen_golden.py:89-95); with bf16 P~ the iters = 20 cascade ends 1.3e-3 px from them, with fp16 inside 1e-3
    conv5_pad2x=False,    # conv_gemm5 also for convs whose couts fill only half of the padded rows (convf2: 64 of 128 -- 78 us instead of 65 + 143 us
                          # of the K-sliced form, but on the side stream it then competes with convc2 for whole CUs: clip time unchanged, 40.8 ms)
)


pack_conv = _packing.pack_conv2


def temporal_pe(T: int, channels: int) -> torch.Tensor:
    """get_temporal_positional_encoding(is_normalize=True, scale=1), ppmtereo_update.py:25-49 -> (T, channels).
    Same torch op sequence as the reference (host side, once per scale); T == 1 yields NaN like the reference."""
    pos = torch.arange(T)
    pos = pos / pos[-1] * 1.0
    pos = pos.unsqueeze(1)
    div = 1.0 / (10000.0 ** (torch.arange(0, channels, 2).float() / channels))
    ang = pos * div
    pe = torch.zeros(T, channels)
    pe[:, 0::2] = torch.sin(ang)
    pe[:, 1::2] = torch.cos(ang)
    return pe


def attn_p_format() -> int:
    """TUNING["attn_p"] as ppms_mem_attn's p_format / ppms_epilogue.vt_f16 (include/ppms.h)."""
    fmt = TUNING["attn_p"]
    if fmt not in ("fp16", "bf16"):
        raise ValueError(f"TUNING['attn_p'] must be 'fp16' or 'bf16', got {fmt!r}")
    return L.ATTN_P_FP16 if fmt == "fp16" else L.ATTN_P_BF16


def softmax_scale(c: int = 128) -> float:
    """ppmstereo.py:494: c^-0.5 * log_12000(key channels = 2c)."""
    return c ** -0.5 * math.log(2 * c, 12000)


# Per-launch HIP events (ConvOp.events, Engine.enable_attn_timing) are only recorded while this is on: bench.py samples them in a
# subset of its timed steps -- every event pair is two more packets in the queue, and with ~330 of them per clip the clip gets ~5 % slower.
KERNEL_TIMING = {"on": True}


class ConvOp:
    """One implicit-GEMM launch: host descriptor (validated by the library) + its device copy."""

    def __init__(self, desc: L.Conv, keep: list, version: int = 2, wm_hint: int = 0, nslice: Optional[int] = None, ysweep: bool = False,
                 device=None):
        self.desc, self.version, self.wm_hint, self.ysweep = desc, version, wm_hint, ysweep
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        raw = bytes(desc)
        self.dev = torch.frombuffer(bytearray(raw), dtype=torch.uint8).clone().to(device)
        self.keep = keep            # tensors whose storage the descriptor points at
        self.events = None          # list collecting (start, stop) HIP events per launch when kernel timing is on
        # small maps: K-sliced launch + reduce kernel (nslice None: ask the library; own workspace per op because ops
        # of the two streams run concurrently)
        self.nslice, self.ws = 1, None
        if version == 5 and nslice is not None and nslice > 1:      # conv_gemm5's K-sliced form (same workspace layout as conv_gemm2's)
            self.nslice = int(nslice)
            self.ws = torch.empty(int(L.load().ppms_conv_gemm2_slice_workspace_bytes(C.byref(desc), self.nslice)), dtype=torch.uint8, device=device)
        if version == 2 and wm_hint == 0 and (nslice is not None or TUNING["slices"]):
            plan = L.load().ppms_conv_gemm2_ysweep_slices if ysweep else L.load().ppms_conv_gemm2_slices
            self.nslice = max(1, int(plan(C.byref(desc)))) if nslice is None else nslice
            if self.nslice > 1:
                self.ws = torch.empty(int(L.load().ppms_conv_gemm2_slice_workspace_bytes(C.byref(desc), self.nslice)), dtype=torch.uint8, device=device)

    def flops(self) -> float:
        """Algorithmic FLOPs of one launch: 2 * pixels * stored couts * input channels of the launch * taps (zero-padding
        taps included, as a FLOP counter on the reference conv would)."""
        d = self.desc
        cout = d.epi[0].n_valid + (d.epi[1].n_valid if d.m_split < d.M else 0)
        cin = d.seg[0].c if d.groups == 2 else sum(d.seg[i].c for i in range(d.nseg))       # (grouped: every cout reads its own segment only)
        return 2.0 * d.T * d.H * d.W * cout * cin * d.kt * d.kh * d.kw

    def mfma_per_product(self) -> float:
        """MFMAs the kernel issues per algorithmic bf16x3 product: 3 (hi*hi, lo*hi, hi*lo), less the hi*lo products conv_gemm5 / conv_gemm6 leave out for the
        input channels from `lo_zero_from` on (bf16-exact activations: their lo plane is all zero).  bench.py prices a launch against
        dense bf16 / this."""
        d = self.desc
        cin = sum(d.seg[i].c for i in range(d.nseg))
        lz = int(d.lo_zero_from)
        if self.version not in (5, 8) or lz <= 0 or lz >= cin or lz % (16 if self.version == 5 else 32):
            return 3.0
        if self.version == 8:
            # conv_gemm6's K loop runs the windows with a lo plane first (phase 0) and needs an EVEN number of k32-steps there (its weight registers
            # alternate between two stages): plan6 (conv_gemm6.hip) ignores lo_zero_from when (lo_zero_from / 32) * (k32-steps per window) is odd
            nsweep = d.kh * d.kw if (d.kh > 1 or d.kw > 1) else 1
            if ((lz // 32) * nsweep) & 1:
                return 3.0
        return 3.0 - (cin - lz) / cin

    def __call__(self):
        ev = self.events if KERNEL_TIMING["on"] else None
        if ev is not None:                  # bench.py: HIP events on the launch stream around this launch
            pair = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            pair[0].record()
            self._launch()
            pair[1].record()
            ev.append(pair)
        else:
            self._launch()

    def _launch(self):
        if self.version == 5 and self.nslice > 1:
            L.check(L.load().ppms_conv_gemm5_sliced(C.byref(self.desc), self.dev.data_ptr(), self.wm_hint, self.nslice, self.ws.data_ptr(), L.stream_ptr()))
        elif self.version == 5:
            L.check(L.load().ppms_conv_gemm5(C.byref(self.desc), self.dev.data_ptr(), self.wm_hint, L.stream_ptr()))
        elif self.version == 8:
            L.check(L.load().ppms_conv_gemm6(C.byref(self.desc), self.dev.data_ptr(), L.stream_ptr()))
        elif self.version == 6:
            L.check(L.load().ppms_gemm1(C.byref(self.desc), self.dev.data_ptr(), self.wm_hint, L.stream_ptr()))
        elif self.version == 7:
            L.check(L.load().ppms_conv_stream(C.byref(self.desc), self.dev.data_ptr(), self.wm_hint, L.stream_ptr()))
        elif self.ysweep:
            L.check(L.load().ppms_conv_gemm2_ysweep(C.byref(self.desc), self.dev.data_ptr(), self.nslice, L.ptr(self.ws), L.stream_ptr()))
        elif self.nslice > 1:
            L.check(L.load().ppms_conv_gemm2_sliced(C.byref(self.desc), self.dev.data_ptr(), self.nslice, self.ws.data_ptr(), L.stream_ptr()))
        else:
            L.check(L.load().ppms_conv_gemm2(C.byref(self.desc), self.dev.data_ptr(), self.wm_hint, L.stream_ptr()))


def epilogue(kind=L.EPI_STORE, act=L.ACT_NONE, scale=1.0, n_valid=0, out_sp: Optional[L.SP] = None, out_f32=None, out_f32_ld=0,
             out_vt=None, aux_sp: Optional[L.SP] = None, aux_f32=None, aux_f32_ld=0, pre_f32=None, pre_off=0, vt_f16=None) -> L.Epilogue:
    e = L.Epilogue()
    e.kind, e.act, e.scale, e.n_valid = kind, act, scale, n_valid
    if out_sp is not None:
        e.out_sp = out_sp
    e.out_f32 = None if out_f32 is None else out_f32.data_ptr()
    e.out_f32_ld = out_f32_ld
    e.out_vt = None if out_vt is None else out_vt.data_ptr()
    e.vt_f16 = attn_p_format() if vt_f16 is None else int(vt_f16)      # (only read with out_vt)
    if aux_sp is not None:
        e.aux_sp = aux_sp
    e.aux_f32 = None if aux_f32 is None else aux_f32.data_ptr()
    e.aux_f32_ld = aux_f32_ld
    if pre_f32 is not None:                  # (P, ld) fp32, this half's columns start at pre_off
        e.pre_f32, e.pre_f32_ld = pre_f32.data_ptr() + 4 * pre_off, pre_f32.shape[1]
    return e


class TimedCall:
    """A small-kernel launch (python callable) that bench.py can bracket with HIP events like a ConvOp."""

    def __init__(self, fn):
        self.fn, self.events = fn, None

    def __call__(self):
        ev = self.events if KERNEL_TIMING["on"] else None
        if ev is not None:
            pair = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            pair[0].record()
            self.fn()
            pair[1].record()
            ev.append(pair)
        else:
            self.fn()


class PwChain:
    """One fused per-pixel layer chain launch (pwchain.hip): host parameter block + device copy."""

    def __init__(self, inp: L.SP, out: L.SP, layers, pixels: int, keep: list, device=None):
        device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        cp = L.ChainParams()
        cp.inp, cp.out, cp.nlayers, cp.P = inp, out, len(layers), pixels
        # (pwchain.hip keeps two activation buffers: the residual operand = the chain input survives only up to the second layer)
        assert not any(resid for _, _, resid, _ in layers[2:]), "pwchain: a residual layer must be the first or the second of its chain"
        for i, (pack, n_valid, resid, post) in enumerate(layers):
            packed, bias, meta = pack
            assert meta["nk"] == 2 and meta["version"] == 2, "chain layers are 1x1 convs with 64 (padded) input channels"
            ly = cp.layer[i]
            ly.w, ly.bias, ly.M, ly.n_valid, ly.resid = packed.data_ptr(), bias.data_ptr(), meta["M"], n_valid, int(resid)
            ly.post_s = None if post is None else post[0].data_ptr()
            ly.post_t = None if post is None else post[1].data_ptr()
            keep += [packed, bias]
        self.pixels, self.keep, self.cp = pixels, keep, cp
        self.dev = torch.frombuffer(bytearray(bytes(cp)), dtype=torch.uint8).clone().to(device)
        self.events = None

    def __call__(self):
        ev = self.events if KERNEL_TIMING["on"] else None
        if ev is not None:
            pair = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            pair[0].record()
        L.check(L.load().ppms_pwchain(self.dev.data_ptr(), self.pixels, L.stream_ptr()))
        if ev is not None:
            pair[1].record()
            ev.append(pair)


class PackedBlock:
    """Packed weights of one SequenceUpdateBlock3D + its Attention_qk (device tensors, built once)."""

    def __init__(self, sd: Dict[str, torch.Tensor], device):
        g = lambda k: sd[k].detach().to(device=device, dtype=torch.float32)
        self.w: Dict[str, tuple] = {}

        self.w4: Dict[str, tuple] = {}             # conv_gemm5 packs (MFMA-fragment order, sweep-ordered taps, M padded to 128)
        self.w6: Dict[str, tuple] = {}             # conv_gemm6 packs (16-cout x 32-channel MFMA A-operand images, sweep-ordered taps)
        self.w1: Dict[str, tuple] = {}             # gemm1 packs of the 1x1 convolutions (MFMA A-operand images per 32 couts x 16 channels)
        self.w7: Dict[str, tuple] = {}             # conv_stream packs (the same images per tap, natural tap order) for the small maps

        def put(name, weight, bias, segs, seg_pad=None, cout_map=None, m_pad=None):
            self.w[name] = pack_conv(weight, bias, segs, seg_pad, cout_map, m_pad)
            w5 = weight if weight.dim() == 5 else weight[:, :, None]
            if TUNING["gemm1"] and tuple(w5.shape[2:]) == (1, 1, 1) and not name.endswith("_y"):
                meta2 = self.w[name][2]
                if sum(meta2["seg_padded"]) % 64 == 0:
                    self.w1[name] = _packing.pack_gemm1(weight, bias, segs, meta2["seg_padded"], cout_map, meta2["M"])
            if TUNING["stream"] and not name.endswith("_y") and sum(self.w[name][2]["seg_padded"]) % 64 == 0:
                self.w7[name] = _packing.pack_stream(weight, bias, segs, self.w[name][2]["seg_padded"], cout_map, self.w[name][2]["M"])
            sweep = w5                                                     # x sweep: natural order
            if w5.shape[3] > 1 and w5.shape[4] > 1:
                # k-step order of the 2-D window sweep: (ky, kx) flattened into the x axis
                sweep = w5.reshape(w5.shape[0], w5.shape[1], w5.shape[2], 1, w5.shape[3] * w5.shape[4]).contiguous()
                self.w[name + "_2d"] = pack_conv(sweep, bias, segs, seg_pad, cout_map, m_pad)
            elif w5.shape[3] > 1:
                sweep = w5.transpose(3, 4).contiguous()                      # y sweep: kh / kw swapped
            rows = (max(cout_map) + 1) if cout_map is not None else w5.shape[0]
            if TUNING["conv6"] and TUNING["conv6_stream"] and w5.shape[2] > 1 and w5.shape[3] == 1 and w5.shape[4] == 1 and rows > 128 and not name.endswith("_y"):
                # (kt,1,1) convs to 256 couts (the z/r conv of the GRU's pass T): conv_gemm6's STREAM form -- one k32-step per 32-channel window of a temporal
                # tap, three window buffers (1/4 scale: 131 us against conv_gemm5's 155; the 128-cout q conv stays on conv_gemm2: 85 us against 89 there)
                pads6 = list(seg_pad) if seg_pad is not None else [((c + 31) // 32) * 32 for c in segs]
                if all(p % 32 == 0 for p in pads6):
                    self.w6[name] = _packing.pack_conv6(w5, bias, segs, pads6, cout_map, 128 if rows <= 128 else (192 if rows <= 192 else 256))
            if TUNING["conv5"] and (w5.shape[3] > 1 or w5.shape[4] > 1) and not name.endswith("_y"):
                # conv_gemm5 serves 128, 192 and 256 rows (192: three 64-cout blocks dealt over the eight waves, round 4 -- convc2's 192 and
                # final_conv's 190 couts no longer run padded to 256)
                m5 = 128 if rows <= 128 else (192 if rows <= 192 and TUNING["conv5_m192"] else 256)
                self.w4[name] = _packing.pack_conv4(sweep, bias, segs, seg_pad, cout_map, m5)
                pads6 = list(seg_pad) if seg_pad is not None else [((c + 31) // 32) * 32 for c in segs]
                if TUNING["conv6"] and all(p % 32 == 0 for p in pads6):
                    self.w6[name] = _packing.pack_conv6(sweep, bias, segs, pads6, cout_map, 128 if rows <= 128 else (192 if rows <= 192 else 256))
            elif TUNING["conv5"] and TUNING["conv5_gemm"] and w5.shape[3] == 1 and w5.shape[4] == 1 and rows > 128 and not name.endswith("_y"):
                # no spatial sweep: conv_gemm5's GEMM mode (windows of 64 channels) when the segments come in such multiples.  Only the
                # 256-cout convs (GRU pass-T z/r 197 -> 164 us, mask_2d.2 62 -> 46 us at the 1/4 scale): with 128 couts the two K-groups
                # get 32-channel windows = 2 k-steps per window, too short a DMA lookahead (pass-T q 110 -> 122 us, to_v 44 -> 66 us)
                pads = list(seg_pad) if seg_pad is not None else [((c + 31) // 32) * 32 for c in segs]
                if all(p % 32 == 0 for p in pads):
                    self.w4[name] = _packing.pack_conv4(w5, bias, segs, pads, cout_map, (rows + 127) // 128 * 128)


        e = "encoder."
        put("init0", g(e + "init_conv.0.weight"), g(e + "init_conv.0.bias"), [128])
        put("init2", g(e + "init_conv.2.weight"), g(e + "init_conv.2.bias"), [64])
        c1 = e + "convc1."
        put("ffn1_0", g(c1 + "ffn1.0.weight"), g(c1 + "ffn1.0.bias"), [36], [64])
        put("ffn1_2", g(c1 + "ffn1.2.weight"), g(c1 + "ffn1.2.bias"), [54], [64])
        put("pw", g(c1 + "pw.weight"), g(c1 + "pw.bias"), [36], [64])
        put("ffn2_0", g(c1 + "ffn2.0.weight"), g(c1 + "ffn2.0.bias"), [36], [64])
        put("ffn2_2", g(c1 + "ffn2.2.weight"), g(c1 + "ffn2.2.bias"), [54], [64])
        self.dw = []
        for i, k in ((0, 1), (1, 7)):
            w = torch.zeros(64, k * k, device=device)
            b = torch.zeros(64, device=device)
            w[:36] = g(c1 + f"conv_list.{i}.weight").reshape(36, k * k)
            b[:36] = g(c1 + f"conv_list.{i}.bias")
            self.dw.append((w.contiguous(), b.contiguous(), k))
        put("convc2", g(e + "convc2.weight"), g(e + "convc2.bias"), [256])
        # convf1 (7x7 on the 2-channel flow) runs as a 1x1 GEMM over the im2col patch [tap*2 + c], 98 -> 128
        wf1 = g(e + "convf1.weight").permute(0, 2, 3, 1).reshape(128, 98, 1, 1)
        put("convf1", wf1, g(e + "convf1.bias"), [98], [128])
        put("convf2", g(e + "convf2.weight"), g(e + "convf2.bias"), [128])
        # final_conv: couts 0..125 -> motion features (rows 0..125), couts 126..189 -> motion hidden state (rows 128..191)
        put("final", g(e + "final_conv.weight"), g(e + "final_conv.bias"), [320], None, list(range(126)) + list(range(128, 192)), 192)
        put("to_v", g("aggregator.to_v.weight"), None, [128])
        put("unc0", g("uncertainty.0.weight"), g("uncertainty.0.bias"), [128, 128])
        self.unc2_w = g("uncertainty.2.weight").reshape(128).contiguous()
        self.unc2_b = float(sd["uncertainty.2.bias"].detach().float().cpu().item())
        gr = "gru."
        cat = lambda a, b: torch.cat([g(a), g(b)], 0)
        put("zr1_0", cat(gr + "convz1.0.weight", gr + "convr1.0.weight"), cat(gr + "convz1.0.bias", gr + "convr1.0.bias"), [128, 384])
        put("z1_2", g(gr + "convz1.2.weight"), g(gr + "convz1.2.bias"), [128])
        put("r1_2", g(gr + "convr1.2.weight"), g(gr + "convr1.2.bias"), [128])
        self.zr1_2_grouped = None                   # z1_2 | r1_2 as one grouped convolution (conv_gemm6 only)
        if TUNING["conv6"] and TUNING["conv6_grouped"]:
            self.zr1_2_grouped = _packing.pack_conv6_grouped([g(gr + "convz1.2.weight"), g(gr + "convr1.2.weight")], [g(gr + "convz1.2.bias"), g(gr + "convr1.2.bias")], 128)
        put("q1", g(gr + "convq1.weight"), g(gr + "convq1.bias"), [128, 384])
        for n in ("2", "3"):
            put("zr" + n, cat(gr + f"convz{n}.weight", gr + f"convr{n}.weight"), cat(gr + f"convz{n}.bias", gr + f"convr{n}.bias"), [128, 384])
            put("q" + n, g(gr + f"convq{n}.weight"), g(gr + f"convq{n}.bias"), [128, 384])
        # the (1,5,1) pass again with kh / kw swapped: k-step order of conv_gemm2's y-swept form (one halo'd window per (dt, chunk))
        put("zr2_y", cat(gr + "convz2.weight", gr + "convr2.weight").transpose(3, 4).contiguous(), cat(gr + "convz2.bias", gr + "convr2.bias"), [128, 384])
        put("q2_y", g(gr + "convq2.weight").transpose(3, 4).contiguous(), g(gr + "convq2.bias"), [128, 384])
        # The GRU input is [h | inp, mf, mfg] (ppmtereo_update.py:292-310, 985-988) and inp does not change between the
        # iterations of one scale: its share of every gate pre-activation is computed once per scale ("*_i" packs, with
        # the bias) and added in the epilogue of the per-iteration convs over [h | mf, mfg] ("*_h" packs).  Not for
        # update_block16, whose time / space attention rewrites all of x every iteration.
        self.hoist = "time_attn.temporal_fc.weight" not in sd and TUNING["hoist"]
        if self.hoist:
            for name in ("zr1_0", "q1", "zr2", "q2", "zr3", "q3"):
                if name.startswith("zr"):
                    n_ = name[2:].replace("1_0", "1.0")
                    wt, bs = cat(gr + f"convz{n_}.weight", gr + f"convr{n_}.weight"), cat(gr + f"convz{n_}.bias", gr + f"convr{n_}.bias")
                else:
                    wt, bs = g(gr + f"convq{name[1:]}.weight"), g(gr + f"convq{name[1:]}.bias")
                w_h = torch.cat([wt[:, :128], wt[:, 256:]], 1).contiguous()
                put(name + "_i", wt[:, 128:256].contiguous(), bs, [128])
                put(name + "_h", w_h, None, [128, 256])
                if name in ("zr2", "q2"):
                    put(name + "_h_y", w_h.transpose(3, 4).contiguous(), None, [128, 256])
                if TUNING["hid_exact"]:
                    # W_mf mf + W_mfg (mf + beta hid) = (W_mf + W_mfg) mf + (beta W_mfg) hid  (ppmstereo.py:552, ppmtereo_update.py:985-988):
                    # the third input block becomes the attention's bf16 read-out itself (sums formed in fp64, rounded once)
                    beta = float(sd["aggregator.beta"].detach().double().reshape(-1)[0])
                    wd = wt.double()
                    w_x = torch.cat([wd[:, :128], wd[:, 256:384] + wd[:, 384:], beta * wd[:, 384:]], 1).float().contiguous()
                    put(name + "_x", w_x, None, [128, 256])
                    if name in ("zr2", "q2"):
                        put(name + "_x_y", w_x.transpose(3, 4).contiguous(), None, [128, 256])
        put("fh1", g("flow_head.conv1.weight"), g("flow_head.conv1.bias"), [128])
        # flow_head.conv2 (256 -> 2, 3x3x3) as a 1x1 GEMM to 27*2 = 54 channels + shifted sum (ppms_tap_gather_sum)
        w2 = g("flow_head.conv2.weight")                                     # (2, 256, 3, 3, 3)
        put("fh2", w2.permute(2, 3, 4, 0, 1).reshape(54, 256, 1, 1, 1).contiguous(), None, [256])
        self.fh2_bias = g("flow_head.conv2.bias").contiguous()
        # upsampling-mask head: mask_2d (Conv2d 3x3 + 1x1 -> 144, ppmtereo_update.py:910-914) or, with use_convex_3d=True,
        # mask_3d (Conv3d 3x3x3 + 1x1x1 -> 432, :903-908)
        self.convex_3d = "mask_3d.0.weight" in sd
        mk = "mask_3d." if self.convex_3d else "mask_2d."
        put("m1", g(mk + "0.weight"), g(mk + "0.bias"), [128])
        put("m2", g(mk + "2.weight"), g(mk + "2.bias"), [256])
        self.mask_ch = 432 if self.convex_3d else 144
        self.beta = g("aggregator.beta").contiguous()
        self.attn = None
        if "time_attn.temporal_fc.weight" in sd:
            self.attn = {k: g(k) for k in sd if k.startswith("time_attn.") or k.startswith("space_attn.")}
            lin = lambda k: g(k)[:, :, None, None]                       # nn.Linear (out, in) as a 1x1 conv
            ta, sa = "time_attn.", "space_attn.encoder_layer."
            # temporal_fc(proj(o)) (ppmtereo_update.py:606-617, Attention.forward :418): two Linear layers with nothing in between are ONE
            # Linear layer -- W = W_fc W_proj, b = W_fc b_proj + b_fc, formed once in fp64 (one launch per iteration instead of two)
            w_fc, w_pj = g(ta + "temporal_fc.weight").double(), g(ta + "temporal_attn.proj.weight").double()
            w_ta = (w_fc @ w_pj).float().contiguous()
            b_ta = (w_fc @ g(ta + "temporal_attn.proj.bias").double() + g(ta + "temporal_fc.bias").double()).float().contiguous()
            put("ta_fc", w_ta[:, :, None, None], b_ta, [384])
            # q, k and v projections of the linear attention read the same x: ONE launch with two epilogue halves (rows 0..767: elu + 1 for q | k,
            # rows 768..1151: v / n)
            put("sa_qkv", torch.cat([lin(sa + "q_proj.weight"), lin(sa + "k_proj.weight"), lin(sa + "v_proj.weight")], 0), None, [384])
            put("sa_merge", lin(sa + "merge.weight"), None, [384])
            put("sa_mlp0", lin(sa + "mlp.0.weight"), None, [384, 384])
            put("sa_mlp2", lin(sa + "mlp.2.weight"), None, [768])
            self.ln = {k: (g(p_ + ".weight").contiguous(), g(p_ + ".bias").contiguous())
                       for k, p_ in (("ta", ta + "temporal_norm1"), ("n1", sa + "norm1"), ("n2", sa + "norm2"))}


class ScaleEngine:
    """All state of one forward_update_block call: buffers, descriptors, iteration stages."""

    def __init__(self, pk: PackedBlock, T: int, h: int, w: int, device, shard=None):
        """T: frames held by THIS engine.  shard (ppmstereo_amd.dist.FrameShard, optional): the window's frames are split in
        contiguous blocks over the ranks; T == shard.f, global frame ids shard.lo .. shard.hi - 1 (see dist.py for what is
        exchanged when)."""
        if shard is not None and shard.world == 1 and not getattr(shard, "force_comm", False):
            shard = None                                   # (force_comm: a single rank that runs every exchange through its process group, dist.FrameShard)
        if shard is not None and T != shard.f:
            raise ValueError(f"sharded engine: {T} local frames, the shard holds {shard.f}")
        self.shard = shard
        self.Tg = Tg = T if shard is None else shard.T          # frames of the whole window
        self.f0 = 0 if shard is None else shard.lo               # global id of local frame 0
        self.halo = 0 if shard is None else shard.HALO           # halo frames around the local block in pixel-indexed buffers
        if Tg > 64:
            raise NotImplementedError("more than 64 frames per window")
        self.pk, self.T, self.h, self.w = pk, T, h, w
        self.n = n = h * w
        self.P = P = T * h * w
        self.dev = device
        self.ksel = min(TOP_K, Tg)
        hp = self.halo * n                                        # halo pixels on each side
        sp = lambda c: L.SPTensor(P, c, device, before=hp, after=hp)
        f32 = lambda *s: torch.zeros(*s, dtype=torch.float32, device=device)
        self.CORR, self.T1, self.C1, self.C2 = sp(64), sp(64), sp(64), sp(64)
        self.COR256, self.CF, self.PATCH, self.FLO1 = sp(256), [sp(320), sp(320)], sp(128), sp(128)
        self.X, self.VAL, self.U1 = sp(384), sp(128), sp(128)
        # update_block16 runs time / space attention on a COPY of x = [inp, mf, mfg] (the reference's inp_tensor is a
        # local of SequenceUpdateBlock3D.forward, ppmtereo_update.py:976-983: inp itself must survive the iteration)
        self.XA = sp(384) if pk.attn is not None else self.X
        if pk.attn is not None:
            self.XT, self.MSG, self.MSGN, self.H1 = sp(384), sp(384), sp(384), sp(768)
            # the temporal attention needs all frames of a pixel: gathered copy of x and its output for every frame of the
            # window; the own block of O1 feeds the (per-frame) layers behind it
            self.O1 = L.SPTensor(P, 384, device, before=self.f0 * n, after=(Tg - self.f0 - T) * n)
            self.XG = self.X if shard is None else L.SPTensor(Tg * n, 384, device)
            self.QKF, self.VF, self.M2, self.M3 = f32(P, 768), f32(P, 384), f32(P, 384), f32(P, 384)
            self.KVWS = f32(int(L.load().ppms_linear_attention_workspace_floats(T, n, 8, 48)))
        self.Hb = [sp(128), sp(128), sp(128)]
        self.ZT, self.RT, self.RH, self.FH1, self.M1 = sp(128), sp(128), sp(128), sp(256), sp(256)
        self.Z, self.MASK, self.QK = f32(P, 128), f32(P, pk.mask_ch), f32(P, 256)
        self._FLOW_full, self._FH2Y_full = f32(P + 2 * hp, 2), f32(P + 2 * hp, 64)       # read across frames (3x3x3 taps): halo'd
        self.FLOW, self.FH2Y = self._FLOW_full[hp:hp + P], self._FH2Y_full[hp:hp + P]
        self.DFLOW = f32(P, 4)
        if pk.hoist:
            self.PRE = {k: torch.empty(P, m, dtype=torch.float32, device=device) for k, m in
                        (("zr1_0", 256), ("q1", 128), ("zr2", 256), ("q2", 128), ("zr3", 256), ("q3", 128))}
            self._pre_stream, self._ev_pre, self._pre_pending = torch.cuda.Stream(device=device), torch.cuda.Event(), False
        self.attn_p = attn_p_format()         # fixed per engine: the to_v descriptor (built below) and every mem_attn call agree on V^T's format
        self.VT = torch.zeros(T, 128, n, dtype=torch.bfloat16, device=device)      # (16-bit storage: bf16, or the fp16 image of bf16 values)
        self.VTG = self.VT if shard is None else torch.zeros(Tg, 128, n, dtype=torch.bfloat16, device=device)     # values of every frame
        self.KG = None if shard is None else f32(Tg * n, 128)                                                     # keys of every frame
        self.QB = torch.zeros(T, n, 128, dtype=torch.bfloat16, device=device)
        self.KB = torch.zeros(T, self.ksel, n, 128, dtype=torch.bfloat16, device=device)
        self.ATT_WS = torch.empty(int(L.load().ppms_mem_attn_workspace_bytes(T, self.ksel, n)), dtype=torch.uint8, device=device)
        self.nblk = (n + 255) // 256
        self.UNC, self.PART = f32(P), f32(T, self.nblk)
        self.PARTG = self.PART if shard is None else f32(Tg, self.nblk)
        self.SIM, self.STRIVE, self.SCORE = f32(Tg, Tg), f32(Tg, Tg), f32(Tg, Tg)
        self.SEL = torch.zeros(Tg, 5, dtype=torch.int32, device=device)
        self.SHAT = f32(Tg, 5)
        self.cells = max(1, (h // 4) * (w // 4))
        self.POOL = f32(2, T, self.cells)
        self.POOLG = self.POOL if shard is None else f32(2, Tg, self.cells)
        self.PE = temporal_pe(Tg, 128).to(device)
        self.FLOW_OUT = f32(T, 2, 4 * h, 4 * w)
        self.scale = softmax_scale(128)
        self.parity = 0             # which CF buffer holds the current motion hidden state
        self.have_mhs = False
        self.lib = L.load()
        self._ev, self._ev_i = None, 0
        self._xa_halo = None        # handle of x's +-2-frame halo while it is in flight (sharded window)
        self._x_hid = False         # True: X[256:384] holds the attention read-out hid (attend()), not mfg = mf + beta hid (set_mfg())
        # independent branches of an iteration (flow encoder || correlation encoder, r-gate || z-gate, mask head || flow
        # head) run on a second HIP stream, fork/joined with events: they fill each other's launch tails
        # the HBM-bound launches of an iteration, bracketable by bench.py like the convolutions (roofline_hbm)
        self.hbm = {"corr_lookup": TimedCall(self._lookup), "attn_prep_k": TimedCall(self._prep_k), "convex_upsample": TimedCall(self._upsample)}
        self._side = torch.cuda.Stream(device=device)
        self._ev_fork, self._ev_join = torch.cuda.Event(), torch.cuda.Event()
        self._build_descriptors()

    # ------------------------------------------------------------------ exchanges of a frame-sharded window (dist.FrameShard)
    def _halo_sp(self, tensors, k: int, async_op: bool = False):
        """+-k boundary frames of one or several SP tensors with the neighbour ranks: both bf16 planes of every tensor travel in ONE
        batch of point-to-point operations.  async_op: the exchange stays in flight (RCCL's own stream) and the returned handle is
        waited for where the halo is read."""
        if self.shard is None:
            return None
        if isinstance(tensors, L.SPTensor):
            tensors = [tensors]
        fr = self.T + 2 * self.halo
        return self.shard.halo_many([(t.data[p].view(fr, self.n, t.channels), k) for t in tensors for p in (0, 1)], async_op)

    def _halo_f32(self, full: torch.Tensor, k: int):
        if self.shard is not None:
            self.shard.halo(full.view(self.T + 2 * self.halo, self.n, full.shape[1]), k)

    # ------------------------------------------------------------------ descriptors
    def _conv(self, wname, segs: List[L.SP], k3, epi0: L.Epilogue, epi1: Optional[L.Epilogue] = None, m_split: Optional[int] = None,
              keep=(), nslice: Optional[int] = None, lo_zero_from: int = 0) -> ConvOp:
        packed, bias, meta = self.pk.w[wname] if isinstance(wname, str) else wname
        d = L.Conv()
        for i, s in enumerate(segs):
            d.seg[i] = s
        d.nseg = len(segs)
        assert [s.c for s in segs] == meta["seg_padded"], (wname, [s.c for s in segs], meta["seg_padded"])
        d.w, d.bias = packed.data_ptr(), bias.data_ptr()
        d.T, d.H, d.W = self.T, self.h, self.w
        d.t_halo = self.halo if k3[0] > 1 else 0        # temporal taps read the neighbour ranks' boundary frames from the halo slabs
        d.lo_zero_from = lo_zero_from                    # input channels from here on are bf16-exact (all-zero lo plane): a promise, see ppms.h
        d.kt, d.kh, d.kw = k3
        d.M = meta["M"]
        d.m_split = meta["M"] if m_split is None else m_split
        d.epi[0] = epi0
        if epi1 is not None:
            d.epi[1] = epi1
        version = 2
        if TUNING["gemm1"] and isinstance(wname, str) and tuple(k3) == (1, 1, 1) and wname in self.pk.w1:
            packed1, bias1, _ = self.pk.w1[wname]
            d1 = L.Conv.from_buffer_copy(bytes(d))
            d1.w, d1.bias = packed1.data_ptr(), bias1.data_ptr()
            if self.lib.ppms_gemm1_applicable(C.byref(d1)) == 1:          # (2: served, but the implicit GEMM is as fast on a map this large)
                return ConvOp(d1, [packed1, bias1, *keep], 6, device=self.dev)
        if TUNING["stream"] and isinstance(wname, str) and wname in self.pk.w7:
            # small maps (1/8, 1/16 scales): the register-streamed kernel -- no K slices, no reduce launch
            packed7, bias7, _ = self.pk.w7[wname]
            d7 = L.Conv.from_buffer_copy(bytes(d))
            d7.w, d7.bias = packed7.data_ptr(), bias7.data_ptr()
            rate = self.lib.ppms_conv_stream_applicable(C.byref(d7))
            if rate == 1 or (rate == 2 and TUNING["stream"] == "all" and self.P <= 16384):      # ("all": A/B runs only)
                return ConvOp(d7, [packed7, bias7, *keep], 7, wm_hint=TUNING["stream_hint"], device=self.dev)
        if isinstance(wname, str):
            op = self._try_fragment_kernels(wname, d, m_split, keep)
            if op is not None:
                return op
        if TUNING["ysweep"] and isinstance(wname, str) and k3[1] > 1:
            # kh > 1 on a map the large-map kernel does not take: conv_gemm2 with one window for all taps of a (dt, chunk)
            # (y-swept "_y" pack for kw == 1, 2-D window "_2d" pack otherwise), when the halo'd window fits
            key = wname + ("_y" if k3[2] == 1 else "_2d")
            if (k3[2] == 1 or TUNING["win2d"]) and key in self.pk.w and self.lib.ppms_conv_gemm2_ysweep_slices(C.byref(d)) > 0:
                packed_y, bias_y, _ = self.pk.w[key]
                d.w, d.bias = packed_y.data_ptr(), bias_y.data_ptr()
                return ConvOp(d, [packed_y, bias_y, *keep], 2, ysweep=True, device=self.dev)
        return ConvOp(d, [packed, bias, *keep], version, nslice=nslice, device=self.dev)

    def _try_fragment_kernels(self, wname: str, d: L.Conv, m_split, keep) -> Optional[ConvOp]:
        """conv_gemm6 (one wave per SIMD, 16x16x32 MFMA, pack_conv6) where the library rates its tile fill, else conv_gemm5 (weights in
        MFMA-fragment order, pack_conv4, couts padded to 128) when it serves the conv."""
        if TUNING["conv6"] and wname in self.pk.w6 and (TUNING.get("conv6_only") is None or wname in TUNING["conv6_only"]):
            packed6, bias6, meta6 = self.pk.w6[wname]
            d6 = L.Conv.from_buffer_copy(bytes(d))
            d6.w, d6.bias, d6.M = packed6.data_ptr(), bias6.data_ptr(), meta6["M"]
            if m_split is None:
                d6.m_split = meta6["M"]
            real6 = d6.epi[0].n_valid + (d6.epi[1].n_valid if d6.m_split < d6.M else 0)
            # (couts filling only half of the padded rows -- convf2's 64 of 128 -- still pay: 3x3 128 -> 64 at the 1/4 scale costs ~45 us here against
            #  63 + 117 us for conv_gemm2's K-sliced launch + its reduce launch; TUNING["conv6_pad2x"])
            if (2 * real6 > meta6["M"] or (TUNING["conv6_pad2x"] and 2 * real6 == meta6["M"])) and self.lib.ppms_conv_gemm6_applicable(C.byref(d6)) == 1:
                return ConvOp(d6, [packed6, bias6, *keep], 8, device=self.dev)
        if not TUNING["conv5"] or wname not in self.pk.w4:
            return None
        packed4, bias4, meta4 = self.pk.w4[wname]
        d4 = L.Conv.from_buffer_copy(bytes(d))
        d4.w, d4.bias, d4.M = packed4.data_ptr(), bias4.data_ptr(), meta4["M"]
        if m_split is None:
            d4.m_split = meta4["M"]
        real = d4.epi[0].n_valid + (d4.epi[1].n_valid if d4.m_split < d4.M else 0)
        if TUNING["conv5"] and (2 * real > meta4["M"] or (TUNING["conv5_pad2x"] and 2 * real == meta4["M"])) and self.lib.ppms_conv_gemm5_applicable(C.byref(d4)):
            return ConvOp(d4, [packed4, bias4, *keep], 5, device=self.dev)
        if TUNING["conv5"] and TUNING["conv5_sliced"] and 2 * real > meta4["M"]:
            ns = int(self.lib.ppms_conv_gemm5_slices(C.byref(d4)))
            if ns >= 2:
                return ConvOp(d4, [packed4, bias4, *keep], 5, nslice=ns, device=self.dev)
        return None

    def _conv_padded(self, wname, *a, **k) -> ConvOp:
        """The 190 / 192-cout convs (convc2, final_conv): conv_gemm5 / conv_gemm6 in their three-cout-block layout where they serve the map, else the
        tight conv_gemm2 pack."""
        if TUNING["conv5"] and wname in self.pk.w4:
            op = self._conv(wname, *a, **k)
            if op.version in (5, 8):
                return op
        return self._conv(wname, *a, **k)

    def _build_descriptors(self):
        E, X, H = epilogue, self.X, self.Hb
        k1, k3 = (1, 1, 1), (1, 3, 3)
        inp, mf, mfg = X.view(0, 128), X.view(128, 128), X.view(256, 128)
        self.op: Dict[str, object] = {}
        o = self.op
        self._qk_ops: Dict[int, ConvOp] = {}
        o["init0"] = self._conv("init0", [inp], k3, E(act=L.ACT_RELU, n_valid=64, out_sp=self.ZT.view(0, 64)))
        if TUNING["pwchain"]:
            w = self.pk.w
            (w1, b1, _), _ = self.pk.dw
            dw1 = (w1.reshape(64).contiguous(), b1)
            # chain A: x1 = gelu(x + ffn1(x)); x2 = gelu(x1 + dw1x1(x1))      CORR -> C2
            o["chainA"] = PwChain(self.CORR.view(), self.C2.view(), [(w["ffn1_0"], 54, False, None), (w["ffn1_2"], 36, True, dw1)], self.P, [dw1[0]],
                                 device=self.dev)
            # chain B: x4 = gelu(x3 + pw x3); cor = gelu(ffn2(x4))              C1 -> COR256
            o["chainB"] = PwChain(self.C1.view(), self.COR256.view(), [(w["pw"], 36, True, None), (w["ffn2_0"], 54, False, None),
                                                                       (w["ffn2_2"], 256, False, None)], self.P, [], device=self.dev)
            (_, _, _), (w7, b7, _) = self.pk.dw
            o["dw7"] = TimedCall(lambda: L.check(self.lib.ppms_dwconv_gelu(self.C2.view(0, 40), self.C1.view(0, 40), w7.data_ptr(), b7.data_ptr(), 7,
                                                                            self.T, self.h, self.w, L.stream_ptr())))
        o["ffn1_0"] = self._conv("ffn1_0", [self.CORR.view()], k1, E(act=L.ACT_GELU, n_valid=54, out_sp=self.T1.view()))
        o["ffn1_2"] = self._conv("ffn1_2", [self.T1.view()], k1, E(L.EPI_RESID, L.ACT_GELU, n_valid=36, out_sp=self.C1.view(), aux_sp=self.CORR.view()))
        o["pw"] = self._conv("pw", [self.C1.view()], k1, E(L.EPI_RESID, L.ACT_GELU, n_valid=36, out_sp=self.C2.view(), aux_sp=self.C1.view()))
        o["ffn2_0"] = self._conv("ffn2_0", [self.C2.view()], k1, E(act=L.ACT_GELU, n_valid=54, out_sp=self.T1.view()))
        o["ffn2_2"] = self._conv("ffn2_2", [self.T1.view()], k1, E(act=L.ACT_GELU, n_valid=256, out_sp=self.COR256.view()))
        o["convf1"] = self._conv("convf1", [self.PATCH.view()], k1, E(act=L.ACT_RELU, n_valid=128, out_sp=self.FLO1.view()))
        for par in (0, 1):
            cf, cf_next = self.CF[par], self.CF[1 - par]
            o[f"init2_{par}"] = self._conv("init2", [self.ZT.view(0, 64)], k3, E(n_valid=64, out_sp=cf.view(256, 64)))
            o[f"convc2_{par}"] = self._conv_padded("convc2", [self.COR256.view()], k3, E(act=L.ACT_RELU, n_valid=192, out_sp=cf.view(0, 192)))
            # (on large maps the library cuts its 400 workgroups in two K slices, and the slice-reduce launch of this side-stream conv crawls on
            # the CUs the main stream's convc2 leaves free -- 143 us, but off the critical path: unsliced is neither faster nor slower)
            o[f"convf2_{par}"] = self._conv("convf2", [self.FLO1.view()], k3, E(act=L.ACT_RELU, n_valid=64, out_sp=cf.view(192, 64)),
                                            nslice=1 if (TUNING["convf2_unsliced"] and self.P >= 32768) else None)
            o[f"final_{par}"] = self._conv_padded("final", [cf.view()], k3, E(act=L.ACT_RELU, n_valid=126, out_sp=mf),
                                           E(act=L.ACT_RELU, n_valid=64, out_sp=cf_next.view(256, 64)), m_split=128)
        o["to_v"] = self._conv("to_v", [mf], k1, E(n_valid=128, out_sp=self.VAL.view(), out_vt=self.VT, vt_f16=self.attn_p))
        o["unc0"] = self._conv("unc0", [H[0].view(), self.VAL.view()], k3, E(act=L.ACT_RELU, n_valid=128, out_sp=self.U1.view()))
        if self.pk.attn is not None:                 # update_block16: time / space attention on x = [inp, mf, mfg]
            o["ta_fc"] = self._conv("ta_fc", [self.O1.view()], k1, E(L.EPI_RESID, n_valid=384, out_sp=self.XT.view(), aux_sp=X.view()))
            o["sa_qkv"] = self._conv("sa_qkv", [self.XT.view()], k1, E(act=L.ACT_ELU1, n_valid=768, out_f32=self.QKF, out_f32_ld=768),
                                     E(scale=1.0 / self.n, n_valid=384, out_f32=self.VF, out_f32_ld=384), m_split=768)
            o["sa_merge"] = self._conv("sa_merge", [self.MSG.view()], k1, E(n_valid=384, out_f32=self.M2, out_f32_ld=384))
            o["sa_mlp0"] = self._conv("sa_mlp0", [self.XT.view(), self.MSGN.view()], k1, E(act=L.ACT_RELU, n_valid=768, out_sp=self.H1.view()))
            o["sa_mlp2"] = self._conv("sa_mlp2", [self.H1.view()], k1, E(n_valid=384, out_f32=self.M3, out_f32_ld=384))
        x_all = self.XA.view()
        hoist = self.pk.hoist
        if hoist:                                   # per-iteration convs see [h | mf, mfg]; the inp share comes from PRE
            x_all = X.view(128, 256)
            for k, kk in (("zr1_0", (1, 1, 15)), ("q1", (1, 1, 5)), ("zr2", (1, 5, 1)), ("q2", (1, 5, 1)), ("zr3", (5, 1, 1)), ("q3", (5, 1, 1))):
                m = self.PRE[k].shape[1]
                o["pre_" + k] = self._conv(k + "_i", [inp], kk, E(n_valid=m, out_f32=self.PRE[k], out_f32_ld=m))
        pre = lambda k, off=0: dict(pre_f32=self.PRE[k], pre_off=off) if hoist else {}
        # GRU pass along W (two-layer z / r), then H, then T: h cycles through Hb[0] -> Hb[1] -> Hb[2] -> Hb[0]
        o["z1_2"] = self._conv("z1_2", [self.ZT.view()], (1, 1, 5), E(act=L.ACT_SIGMOID, n_valid=128, out_f32=self.Z, out_f32_ld=128))
        o["r1_2"] = self._conv("r1_2", [self.RT.view()], (1, 1, 5), E(L.EPI_RH, n_valid=128, out_sp=self.RH.view(), aux_sp=H[0].view()))
        if self.pk.zr1_2_grouped is not None and TUNING["conv6_grouped"]:       # both tails in one grouped launch where conv_gemm6 rates the map (else the two launches above)
            packed_g, bias_g, meta_g = self.pk.zr1_2_grouped
            dg = L.Conv.from_buffer_copy(bytes(o["z1_2"].desc))
            dg.seg[0], dg.seg[1], dg.nseg, dg.groups = self.ZT.view(), self.RT.view(), 2, 2
            dg.w, dg.bias, dg.M, dg.m_split = packed_g.data_ptr(), bias_g.data_ptr(), 256, 128
            dg.epi[0], dg.epi[1] = o["z1_2"].desc.epi[0], o["r1_2"].desc.epi[0]
            if self.lib.ppms_conv_gemm6_applicable(C.byref(dg)) == 1:
                o["zr1_2"] = ConvOp(dg, [packed_g, bias_g, self.ZT, self.RT, self.Z, self.RH, H[0]], 8, device=self.dev)
        # the convs over x exist twice on hoisted blocks: "" reads [h | mf, mfg] (the reference's operands: update() driven with a caller's mfg),
        # "_x" reads [h | mf, hid] with the folded weights and skips the products with hid's all-zero lo plane (the loop: attend() leaves hid)
        self.hid_mode = bool(hoist and TUNING["hid_exact"])
        for tag, sfx, lz in ((("", "_h" if hoist else "", 0), ("_x", "_x", 256)) if self.hid_mode else (("", "_h" if hoist else "", 0),)):
            o["zr1_0" + tag] = self._conv("zr1_0" + sfx, [H[0].view(), x_all], (1, 1, 15), E(act=L.ACT_GELU, n_valid=128, out_sp=self.ZT.view(), **pre("zr1_0")),
                                          E(act=L.ACT_GELU, n_valid=128, out_sp=self.RT.view(), **pre("zr1_0", 128)), m_split=128, lo_zero_from=lz)
            o["q1" + tag] = self._conv("q1" + sfx, [self.RH.view(), x_all], (1, 1, 5),
                                       E(L.EPI_GRU, n_valid=128, out_sp=H[1].view(), aux_sp=H[0].view(), aux_f32=self.Z, aux_f32_ld=128, **pre("q1")), lo_zero_from=lz)
            for n, kk, src, dst in (("2", (1, 5, 1), 1, 2), ("3", (5, 1, 1), 2, 0)):
                o["zr" + n + tag] = self._conv("zr" + n + sfx, [H[src].view(), x_all], kk,
                                               E(act=L.ACT_SIGMOID, n_valid=128, out_f32=self.Z, out_f32_ld=128, **pre("zr" + n)),
                                               E(L.EPI_RH, n_valid=128, out_sp=self.RH.view(), aux_sp=H[src].view(), **pre("zr" + n, 128)), m_split=128,
                                               lo_zero_from=lz)
                o["q" + n + tag] = self._conv("q" + n + sfx, [self.RH.view(), x_all], kk,
                                              E(L.EPI_GRU, n_valid=128, out_sp=H[dst].view(), aux_sp=H[src].view(), aux_f32=self.Z, aux_f32_ld=128,
                                                **pre("q" + n)), lo_zero_from=lz)
        o["fh1"] = self._conv("fh1", [H[0].view()], (3, 3, 3), E(act=L.ACT_RELU, n_valid=256, out_sp=self.FH1.view()))
        o["fh2"] = self._conv("fh2", [self.FH1.view()], k1, E(n_valid=54, out_f32=self.FH2Y, out_f32_ld=64))
        o["m1"] = self._conv("m1", [H[0].view()], (3, 3, 3) if self.pk.convex_3d else k3, E(act=L.ACT_RELU, n_valid=256, out_sp=self.M1.view()))
        o["m2"] = self._conv("m2", [self.M1.view()], k1, E(scale=0.25, n_valid=self.pk.mask_ch, out_f32=self.MASK, out_f32_ld=self.pk.mask_ch))

    # ------------------------------------------------------------------ loading state (reference NCHW tensors)
    def _s(self):
        return L.stream_ptr()

    def load_nchw(self, x: torch.Tensor, dst: L.SP):
        x = x.contiguous().float()
        L.require_gpu(x)
        L.check(self.lib.ppms_nchw_to_sp(x.data_ptr(), dst, x.shape[0], x.shape[1], x.shape[2] * x.shape[3], self._s()))

    def store_nchw(self, src: L.SP, c: int) -> torch.Tensor:
        out = torch.empty(self.T, c, self.h, self.w, dtype=torch.float32, device=self.dev)
        L.check(self.lib.ppms_sp_to_nchw(src, out.data_ptr(), self.T, c, self.n, self._s()))
        return out

    def _drain_xa_halo(self):
        """A halo exchange of x left in flight by attend() with no update() behind it (an exception mid-iteration, a probe that calls
        attend() alone, the engine re-used for another clip): wait for it and forget it, so that the next update() exchanges the
        CURRENT x instead of consuming the stale handle.  attend() is a collective call on a sharded engine: every rank must make it."""
        if self._xa_halo is not None:
            self._xa_halo.wait()
            self._xa_halo = None

    def set_inp(self, inp: torch.Tensor):
        self._drain_xa_halo()
        self.load_nchw(inp, self.X.view(0, 128))
        self._halo_sp(self.X, 2)                    # (sharded window) inp's +-2 frames: read by the hoisted (5,1,1) gate convs
        if self.pk.hoist:                          # inp share of the GRU gate pre-activations, once per scale, on its own stream
            self._ev_fork.record()
            self._pre_stream.wait_event(self._ev_fork)
            with torch.cuda.stream(self._pre_stream):
                for k in ("zr1_0", "q1", "zr2", "q2", "zr3", "q3"):
                    self.op["pre_" + k]()
                self._ev_pre.record()
            self._pre_pending = True

    def set_net(self, net: torch.Tensor):
        self.load_nchw(net, self.Hb[0].view())

    def set_mf(self, mf: torch.Tensor):
        self.load_nchw(mf, self.X.view(128, 128))

    def set_mfg(self, mfg: torch.Tensor):
        self.load_nchw(mfg, self.X.view(256, 128))
        self._x_hid = False

    def set_flow(self, flow: torch.Tensor):
        f = flow.contiguous().float()
        L.require_gpu(f)
        L.check(self.lib.ppms_nchw_to_nhwc(f.data_ptr(), self.FLOW.data_ptr(), 2, self.T, 2, self.n, self._s()))

    def set_mhs(self, mhs: Optional[torch.Tensor]):
        self.parity = 0
        self.have_mhs = mhs is not None
        if mhs is not None:
            self.load_nchw(mhs, self.CF[self.parity].view(256, 64))

    # SP views of the state the cascade hands to the next scale (ppmstereo.py:726-732, 763-767)
    def net_view(self) -> L.SP:
        return self.Hb[0].view()

    def mhs_view(self) -> L.SP:
        return self.CF[self.parity].view(256, 64)

    def UNC_local(self) -> torch.Tensor:
        return self.UNC

    def get_net(self):
        return self.store_nchw(self.Hb[0].view(), 128)

    def get_mhs(self):
        return self.store_nchw(self.CF[self.parity].view(256, 64), 64)

    def get_mf(self):
        return self.store_nchw(self.X.view(128, 128), 128)

    def get_mfg(self):
        if self._x_hid:             # the buffer holds hid: mfg = mf + beta * hid (ppmstereo.py:552), formed on demand (tests, API parity)
            mfg = self.X.to_f32(128, 128) + self.pk.beta * self.X.to_f32(256, 128)
            return mfg.reshape(self.T, self.h, self.w, 128).permute(0, 3, 1, 2).contiguous()
        return self.store_nchw(self.X.view(256, 128), 128)

    def get_value(self):
        return self.store_nchw(self.VAL.view(), 128)

    def get_flow(self):
        out = torch.empty(self.T, 2, self.h, self.w, dtype=torch.float32, device=self.dev)
        L.check(self.lib.ppms_nhwc_to_nchw(self.FLOW.data_ptr(), 2, out.data_ptr(), self.T, 2, self.n, self._s()))
        return out

    def get_dflow(self):
        out = torch.empty(self.T, 2, self.h, self.w, dtype=torch.float32, device=self.dev)
        L.check(self.lib.ppms_nhwc_to_nchw(self.DFLOW.data_ptr(), 4, out.data_ptr(), self.T, 2, self.n, self._s()))
        return out

    def get_mask(self):
        mc = self.pk.mask_ch
        out = torch.empty(self.T, mc, self.h, self.w, dtype=torch.float32, device=self.dev)
        L.check(self.lib.ppms_nhwc_to_nchw(self.MASK.data_ptr(), mc, out.data_ptr(), self.T, mc, self.n, self._s()))
        return out

    def get_unc(self):
        return self.UNC.view(self.T, 1, self.h, self.w).clone()

    # ------------------------------------------------------------------ once per scale
    def begin(self, pyramid: List[torch.Tensor], qk_pack):
        """q/k projection, Q operand, frame similarity, usage counter (ppmstereo.py:447-475).
        pyramid: levels of CorrBlock1D; qk_pack: Attention_qk.packed(device)."""
        self._drain_xa_halo()
        self.pyr = pyramid
        self.pyr_ptrs = (C.c_void_p * 4)(*[p.data_ptr() for p in pyramid[:4]])
        key = qk_pack[0].data_ptr()
        if key not in self._qk_ops:
            self._qk_ops[key] = self._conv(qk_pack, [self.X.view(0, 128)], (1, 1, 1), epilogue(n_valid=256, out_f32=self.QK, out_f32_ld=256))
        self._qk_ops[key]()
        s = self._s()
        pe_local = self.PE.data_ptr() + self.f0 * 128 * 4
        L.check(self.lib.ppms_attn_prep_q(self.QK.data_ptr(), 256, pe_local, self.QB.data_ptr(), self.T, self.n, s))
        L.check(self.lib.ppms_qk_pool(self.QK.data_ptr(), self.QK.data_ptr() + 128 * 4, 256, self.POOL.data_ptr(), self.T, self.h, self.w, s))
        if self.shard is not None:
            # frame descriptors and keys of every frame of the window: once per scale (keys stay fp32: K' = bf16(K s + PE) must
            # round exactly as in the unsharded loop)
            g, _ = self.shard.all_gather(self.POOL.permute(1, 0, 2).contiguous())               # (Tg, 2, cells)
            self.POOLG.copy_(g.permute(1, 0, 2))
            self.shard.all_gather(self.QK[:, 128:].contiguous(), out=self.KG)
        L.check(self.lib.ppms_qk_cos(self.POOLG.data_ptr(), self.SIM.data_ptr(), self.Tg, self.cells, s))
        self.STRIVE.fill_(1.0)

    # ------------------------------------------------------------------ iteration stages
    def lookup(self):
        self.hbm["corr_lookup"]()

    def _lookup(self):
        X = self.X
        fl = X.view(254, 2)
        L.check(self.lib.ppms_corr_lookup(self.pyr_ptrs, self.FLOW.data_ptr(), 1, None, self.CORR.view().hi, self.CORR.view().lo, 64,
                                          fl.hi, fl.lo, 384, self.T, self.h, self.w, self._s()))

    def hbm_bytes(self) -> Dict[str, float]:
        """Algorithmic HBM bytes of one launch of the HBM-bound kernels of this scale, every tensor touched once:
        lookup T n (4 levels x 10 taps x 4 B in + 36 x 4 B out + 8 B flow); key modulation K' = bf16(K s + PE): the fp32 keys of the
        window's frames once (Tg n 128 x 4 B) + T k n 128 x 2 B of bf16 K' out; convex upsampling T n (144 + 2) 4 B in + T 16 n 2 x 4 B out; pyramid build 2 x 256 T n 4 B in + 1.9375 T n w 4 B out (five levels: 1 + 1/2 + 1/4 + 1/8 + 1/16)."""
        T, n, w = self.T, self.n, self.w
        return dict(corr_lookup=T * n * (4 * 10 * 4 + 36 * 4 + 8.0), attn_prep_k=self.Tg * n * 128 * 4.0 + T * self.ksel * n * 128 * 2.0,
                    convex_upsample=T * n * (self.pk.mask_ch + 2) * 4.0 + T * 16 * n * 2 * 4.0,
                    corr_build=2 * 256 * T * n * 4.0 + 1.9375 * T * n * w * 4.0)

    def _fork(self):
        if self.P < TUNING["fork_min_pixels"]:
            return contextlib.nullcontext()
        self._ev_fork.record()
        self._side.wait_event(self._ev_fork)
        return torch.cuda.stream(self._side)

    def _join(self):
        if self.P < TUNING["fork_min_pixels"]:
            return
        self._ev_join.record(self._side)
        torch.cuda.current_stream().wait_event(self._ev_join)

    def motion_and_value(self):
        o, s, par = self.op, self._s(), self.parity
        with self._fork():                        # flow branch: convf1 (7x7 via im2col) -> convf2
            L.check(self.lib.ppms_flow_patch7(self.FLOW.data_ptr(), self.PATCH.view(), self.T, self.h, self.w, self._s()))
            o["convf1"]()
            o[f"convf2_{par}"]()
        if not self.have_mhs:                     # init_conv(inp), ppmtereo_update.py:469-471
            o["init0"]()
            o[f"init2_{par}"]()
            self.have_mhs = True
        (w1, b1, _), (w7, b7, _) = self.pk.dw
        if "chainA" in o:                          # fused per-pixel chains around the depthwise 7x7 (pwchain.hip)
            o["chainA"]()
            o["dw7"]()
            o["chainB"]()
        else:
            o["ffn1_0"]()
            o["ffn1_2"]()
            L.check(self.lib.ppms_dwconv_gelu(self.C1.view(0, 40), self.C2.view(0, 40), w1.data_ptr(), b1.data_ptr(), 1, self.T, self.h, self.w, s))
            L.check(self.lib.ppms_dwconv_gelu(self.C2.view(0, 40), self.C1.view(0, 40), w7.data_ptr(), b7.data_ptr(), 7, self.T, self.h, self.w, s))
            o["pw"]()
            o["ffn2_0"]()
            o["ffn2_2"]()
        o[f"convc2_{par}"]()
        self._join()
        o[f"final_{par}"]()
        self.parity = 1 - par                     # the new motion hidden state went to the other CF buffer
        o["to_v"]()

    def uncertainty(self):
        self.op["unc0"]()
        L.check(self.lib.ppms_unc_tail(self.U1.view(), self.pk.unc2_w.data_ptr(), self.pk.unc2_b, self.UNC.data_ptr(), self.PART.data_ptr(),
                                       self.T, self.n, self._s()))

    def pick(self):
        if self.shard is not None:
            # ONE direct exchange per iteration for the memory read-out: the values of every frame (bf16, as the reference casts them;
            # new every iteration) and the frames' confidence sums -- every rank then scores all T x T frame pairs from the same numbers
            self.shard.gather_many([(self.VT, self.VTG), (self.PART, self.PARTG)])
        L.check(self.lib.ppms_qam_select(self.SIM.data_ptr(), self.STRIVE.data_ptr(), self.PARTG.data_ptr(), self.nblk, self.n, self.SEL.data_ptr(),
                                         self.SHAT.data_ptr(), self.SCORE.data_ptr(), self.Tg, self._s()))

    def attend(self, out_bf16: Optional[torch.Tensor] = None):
        s = self._s()
        sel = self.SEL.data_ptr() + self.f0 * 5 * 4                   # rows of this engine's clips
        shat = self.SHAT.data_ptr() + self.f0 * 5 * 4
        if self.shard is None:
            key, key_ld = self.QK.data_ptr() + 128 * 4, 256
        else:
            key, key_ld = self.KG.data_ptr(), 128                      # (the values of every frame arrived with the confidences: pick())
        self._prep_k_args = (key, key_ld, sel, shat)
        self.hbm["attn_prep_k"]()
        ev = None
        if KERNEL_TIMING["on"] and self._ev is not None and self._ev_i < len(self._ev):
            ev = self._ev[self._ev_i]
            self._ev_i += 1
            ev[0].record()
        # hoisted blocks: X[256:384] receives hid itself (bf16-exact), the GRU convs use the folded weights ("_x" ops)
        mf_in = L.SP(None, None, 0, 0) if self.hid_mode else self.X.view(128, 128)
        L.check(self.lib.ppms_mem_attn(self.QB.data_ptr(), self.KB.data_ptr(), self.VTG.data_ptr(), sel, self.ksel, self.scale,
                                       self.pk.beta.data_ptr(), mf_in, self.X.view(256, 128), L.ptr(out_bf16), self.T, self.n,
                                       self.ATT_WS.data_ptr(), 0, self.attn_p, s))
        self._x_hid = self.hid_mode
        if ev is not None:
            ev[1].record()
        if self.shard is not None and self.pk.attn is None:
            self._drain_xa_halo()                                     # (attend() twice without update(): the older exchange is obsolete)
            # x = [inp | mf, mfg] is final here: its +-2 frames (read by the temporal GRU pass at the END of update()) start travelling
            # now and arrive under the W and H passes
            self._xa_halo = self._halo_sp(self.XA, 2, async_op=True)

    def conv_ops(self, version: Optional[int] = None):
        """name -> ConvOp of every implicit-GEMM launch of an iteration (version: only that kernel generation's -- 2 conv_gemm2, 5 conv_gemm5, 6 gemm1, 7 conv_stream, 8 conv_gemm6)."""
        return {k: v for k, v in self.op.items() if isinstance(v, ConvOp) and (version is None or v.version == version)}

    def conv_family_ops(self):
        """name -> launch object of EVERY convolution-family launch of this scale: implicit-GEMM convs (with their slice-reduce halves),
        the once-per-scale hoisted-share and q/k projections, the fused per-pixel chains and the depthwise 7x7."""
        fam = {k: v for k, v in self.op.items() if isinstance(v, (ConvOp, PwChain, TimedCall))}
        fam.update({f"to_qk_{i}": v for i, v in enumerate(self._qk_ops.values())})
        return fam

    def attn_redo_count(self):
        """(tiles, flagged): how many (clip, split, 256-query block) tiles the LAST mem_attn call of this engine ran on the 64-query kernel and how
        many of them that kernel handed to its fix-up pass (a score more than 2^16 -- fp16 P~ -- or 2^60 -- bf16 -- above the query's softmax
        reference).  Diagnostics (tools/parity_ab.py, bench.py); synchronises.  (0, 0) when the 64-query kernel does not serve this geometry."""
        if self.n % 64:
            return 0, 0
        flags = self.ATT_WS.view(torch.int32)[self.T * self.ksel * self.n * 130:]
        g64 = (self.n + 255) // 256
        used = self.T * self.ksel * g64                       # an upper bound on the tiles of the call (two-frame splits use fewer)
        f = flags[:used * 2:2]
        torch.cuda.synchronize()
        return int(used), int((f != 0).sum().item())

    def enable_attn_timing(self, launches: int):
        """HIP events (on the stream the kernel is launched on) around the next `launches` mem_attn launches."""
        self._ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(launches)]
        self._ev_i = 0

    def attn_times_ms(self):
        torch.cuda.synchronize()
        return [a.elapsed_time(b) for a, b in self._ev[:self._ev_i]]

    def block16_attention(self):
        """TimeAttnBlock + SpaceAttnBlock on x = [inp, mf, mfg] (update_block16 only, ppmtereo_update.py:593-631,
        980-983): LayerNorm / temporal attention / linear-attention kernels (attn16.hip) + seven 1x1 GEMMs."""
        o, s, lib, ln = self.op, self._s(), self.lib, self.pk.ln
        none_sp = L.SP(None, None, 0, 0)
        # TimeAttnBlock: x + fc(proj(attn_T(LN(x))))                                      ppmtereo_update.py:603-618
        if self.shard is not None:                  # attention over the T frames of a pixel: x of every frame of the window (both planes, one exchange)
            self.shard.gather_many([(self.X.own()[pl].view(self.T, self.n, 384), self.XG.data[pl].view(self.Tg, self.n, 384)) for pl in (0, 1)])
        L.check(lib.ppms_time_attn(self.XG.view(all_rows=True), ln["ta"][0].data_ptr(), ln["ta"][1].data_ptr(), self.O1.view(all_rows=True),
                                   self.Tg, self.n, 8, s))
        o["ta_fc"]()                                  # x + fc(proj(.)) as one premultiplied layer
        # SpaceAttnBlock = LoFTR encoder layer with linear attention, x = source                 attention.py:164-190
        o["sa_qkv"]()
        L.check(lib.ppms_linear_attention(self.QKF.data_ptr(), 768, self.QKF.data_ptr() + 384 * 4, 768, self.VF.data_ptr(), 384,
                                          self.KVWS.data_ptr(), self.MSG.view(), self.T, self.n, 8, 48, s))
        o["sa_merge"]()
        L.check(lib.ppms_layernorm(self.M2.data_ptr(), 384, ln["n1"][0].data_ptr(), ln["n1"][1].data_ptr(), none_sp, self.MSGN.view(), self.P, 384, s))
        o["sa_mlp0"]()
        o["sa_mlp2"]()
        L.check(lib.ppms_layernorm(self.M3.data_ptr(), 384, ln["n2"][0].data_ptr(), ln["n2"][1].data_ptr(), self.XT.view(), self.XA.view(), self.P, 384, s))

    def update(self, need_mask: bool = True):
        """need_mask False: the mask head is skipped -- its only consumer is the convex upsampling of THIS iteration's flow
        (ppmstereo.py:573-576), which test_mode callers need from the last iteration of a scale only."""
        o = self.op
        if self.pk.attn is not None:
            self.block16_attention()
            if self.shard is not None:              # (update_block16 rewrites x: its halo starts here, behind the space attention)
                self._xa_halo = self._halo_sp(self.XA, 2, async_op=True)
        if self.pk.hoist and self._pre_pending:
            torch.cuda.current_stream().wait_event(self._ev_pre)
            self._pre_pending = False
        tag = "_x" if self._x_hid else ""         # which operands X[256:384] holds: hid (the loop) or mfg (a caller's)
        o["zr1_0" + tag]()
        if "zr1_2" in o:
            o["zr1_2"]()
        else:
            with self._fork():
                o["r1_2"]()
            o["z1_2"]()
            self._join()
        for k in ("q1", "zr2", "q2"):
            o[k + tag]()
        # GRU pass along T, (5,1,1) convs (ppmtereo_update.py:305-310): +-2 frames of [h | mf, mfg], then of r*h
        if self.shard is not None:
            self._halo_sp(self.Hb[2], 2)            # h after the H pass: needed at once
            if self._xa_halo is not None:           # x's halo has been in flight since attend() / the block16 attention
                self._xa_halo.wait()
                self._xa_halo = None
            else:                                   # (update() driven without attend(): the module-level forward())
                self._halo_sp(self.XA, 2)
        o["zr3" + tag]()
        self._halo_sp(self.RH, 2)
        o["q3" + tag]()
        self._halo_sp(self.Hb[0], 1)                # FlowHead3D / mask_3d: 3x3x3 convs of the new hidden state
        if need_mask:
            with self._fork():                    # mask head || flow head
                o["m1"]()
                o["m2"]()
        o["fh1"]()
        o["fh2"]()
        self._halo_f32(self._FH2Y_full, 1)          # the second 3x3x3 conv gathers the 54 pre-gather channels over +-1 frame
        # delta_flow = the gathered taps, and flow = flow + delta_flow (ppmstereo.py:571) in the same launch
        L.check(self.lib.ppms_tap_gather_sum(self.FH2Y.data_ptr(), 64, self.pk.fh2_bias.data_ptr(), self.DFLOW.data_ptr(), 4, self.FLOW.data_ptr(), 2,
                                             2, 3, 3, 3, self.T, self.h, self.w, self.halo, self._s()))
        if need_mask:
            self._join()

    def upsample(self) -> torch.Tensor:
        self.hbm["convex_upsample"]()
        return self.FLOW_OUT

    def _prep_k(self):
        key, key_ld, sel, shat = self._prep_k_args
        L.check(self.lib.ppms_attn_prep_k(key, key_ld, self.PE.data_ptr(), sel, shat, self.KB.data_ptr(), self.T, self.ksel, self.n, self._s()))

    def _upsample(self):
        if self.pk.convex_3d:                       # ppmstereo.py:573-576
            self._halo_f32(self._FLOW_full, 1)      # 27 spatio-temporal neighbours: +-1 frame of the flow
            L.check(self.lib.ppms_convex_upsample_3d(self.FLOW.data_ptr(), self.MASK.data_ptr(), self.pk.mask_ch, self.FLOW_OUT.data_ptr(), self.T,
                                                     self.h, self.w, self.halo, self._s()))
        else:
            L.check(self.lib.ppms_convex_upsample(self.FLOW.data_ptr(), self.MASK.data_ptr(), self.pk.mask_ch, self.FLOW_OUT.data_ptr(), self.T,
                                                  self.h, self.w, self._s()))

    def iterate(self, need_up: bool = True):
        """One refinement iteration (ppmstereo.py:482-576).  need_up False: the iteration's upsampled prediction is not wanted (test_mode
        returns the last one only, :801-804), so the mask head and the convex upsampling -- which feed nothing else -- are not run and
        None is returned; the recurrent state (flow, hidden states, memory) is identical either way."""
        self.lookup()
        self.motion_and_value()
        self.uncertainty()
        self.pick()
        self.attend()
        self.update(need_mask=need_up)
        return self.upsample() if need_up else None


def bilinear(x: torch.Tensor, size, align_corners: bool, mul: float = 1.0) -> torch.Tensor:
    """F.interpolate(mode="bilinear") replacement on NCHW fp32 (utils.py:10-16, ppmstereo.py:578,726)."""
    x = x.contiguous().float()
    L.require_gpu(x)
    N, Cc, H, W = x.shape
    out = torch.empty(N, Cc, size[0], size[1], dtype=torch.float32, device=x.device)
    L.check(L.load().ppms_bilinear(x.data_ptr(), out.data_ptr(), N, Cc, H, W, size[0], size[1], int(align_corners), mul, L.stream_ptr()))
    return out
