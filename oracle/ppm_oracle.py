"""CPU ORACLE for the PPMStereo hot path -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

A from-scratch restatement (torch CPU ops, fp32) of the reference algorithm on the path
``PPMStereo.forward_update_block`` and its callees.  Only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import this module; the product path
(``ppmstereo_amd``) never does and fails loudly when its HIP library is missing.

Parity status: PINNED.  ``tools/gen_golden.py`` imports the reference itself in the build container
(with the four import shims of SURVEY.md Appendix A), runs it on seeded inputs and procedural
weights and commits the resulting vectors under ``tests/golden/``; ``tests/test_oracle_golden.py``
checks every function here against them.  The one third-party piece, ``flash_attn.flash_attn_func``
(un-vendored, un-pinned; call site /root/reference/models/core/ppmstereo.py:550), is restated from its
published semantics: softmax(Q K^T * scale) V with fp32 softmax/accumulate, bf16 in, bf16 out.

Every function cites the reference file:line it follows (paths relative to /root/reference).
Weights are passed as dicts keyed by the reference ``state_dict`` names (see
``ppmstereo_amd/weights.py``).
"""
from __future__ import annotations

import math
from typing import Dict, List, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
TOP_K = 5  # models/core/ppmstereo.py:445


# --------------------------------------------------------------------------------------
# correlation pyramid  (models/core/corr.py)
# --------------------------------------------------------------------------------------
def coords_grid(batch: int, ht: int, wd: int) -> Tensor:
    """corr.py:47-52 -- (B,2,H,W): channel 0 = x index, channel 1 = y index."""
    ys, xs = torch.meshgrid(torch.arange(ht), torch.arange(wd), indexing="ij")
    return torch.stack([xs, ys], 0).float()[None].repeat(batch, 1, 1, 1)


def corr_volume(fmap1: Tensor, fmap2: Tensor) -> Tensor:
    """corr.py:96-104 -- vol[b,y,x1,x2] = sum_c f1[b,c,y,x1] f2[b,c,y,x2] / sqrt(C)."""
    B, D, H, W1 = fmap1.shape
    vol = torch.einsum("bcyi,bcyj->byij", fmap1.float(), fmap2.float())
    return vol / math.sqrt(D)


def corr_pyramid(fmap1: Tensor, fmap2: Tensor, num_levels: int = 4) -> List[Tensor]:
    """corr.py:56-72 -- num_levels+1 entries (B*H*W1, 1, 1, W2_l); the last one is never read."""
    vol = corr_volume(fmap1, fmap2)
    B, H, W1, W2 = vol.shape
    lvl = vol.reshape(B * H * W1, 1, 1, W2)
    pyr = [lvl]
    for _ in range(num_levels):
        w = lvl.shape[-1] // 2
        lvl = 0.5 * (lvl[..., 0:2 * w:2] + lvl[..., 1:2 * w:2])       # avg_pool2d([1,2]) floors odd widths
        pyr.append(lvl)
    return pyr


def corr_lookup(pyr: List[Tensor], flow: Tensor, num_levels: int = 4, radius: int = 4) -> Tensor:
    """corr.py:74-94 + bilinear_sampler :10-27 (grid_sample, align_corners=True, zero padding).

    p = (x + flow_x)/2^l + (kk - r); out[b, 9l+kk, y, x] = (1-a) L_l[.., floor p] + a L_l[.., floor p + 1],
    taps outside [0, W_l-1] contribute 0.  The y component of the flow is ignored (:77).
    """
    B, _, H, W = flow.shape
    x = coords_grid(B, H, W)[:, 0] + flow[:, 0]                       # (B,H,W)
    x = x.reshape(B * H * W, 1)
    outs = []
    dx = torch.arange(-radius, radius + 1, dtype=torch.float32)[None]  # (1,9)
    for l in range(num_levels):
        L = pyr[l].reshape(B * H * W, -1)
        Wl = L.shape[1]
        # the reference normalises p -> g = 2p/(W_l-1) - 1 (corr.py:14) and grid_sample maps it back with
        # ((g+1)/2)*(W_l-1); the same fp32 op sequence is kept so the round trip rounds identically
        g = 2 * (dx + x / 2 ** l) / (Wl - 1) - 1
        p = ((g + 1) / 2) * (Wl - 1)
        p0 = torch.floor(p)
        a = p - p0
        i0 = p0.long()
        i1 = i0 + 1
        v0 = torch.gather(L, 1, i0.clamp(0, Wl - 1)) * ((i0 >= 0) & (i0 <= Wl - 1))
        v1 = torch.gather(L, 1, i1.clamp(0, Wl - 1)) * ((i1 >= 0) & (i1 <= Wl - 1))
        outs.append((1 - a) * v0 + a * v1)
    out = torch.cat(outs, 1).reshape(B, H, W, -1)
    return out.permute(0, 3, 1, 2).contiguous().float()


# --------------------------------------------------------------------------------------
# temporal PE, frame similarity, QAM pick, play (models/core/ppmstereo.py, ppmtereo_update.py)
# --------------------------------------------------------------------------------------
def temporal_pe(T: int, channels: int) -> Tensor:
    """ppmtereo_update.py:25-49 with is_normalize=True, scale=1 (call site ppmstereo.py:453-458).
    Returns (T, channels).  T == 1 gives 0/0 = NaN exactly like the reference."""
    pos = torch.arange(T)
    pos = pos / pos[-1] * 1.0
    pos = pos.unsqueeze(1)
    div = 1.0 / (10000.0 ** (torch.arange(0, channels, 2).float() / channels))
    ang = pos * div
    pe = torch.zeros(T, channels)
    pe[:, 0::2] = torch.sin(ang)
    pe[:, 1::2] = torch.cos(ang)
    return pe


def qk_similarity(q: Tensor, k: Tensor) -> Tensor:
    """ppmstereo.py:397-423.  q,k: (T,C,h,w) of one batch element -> sim (T,T) with
    sim[i,j] = cos(kbar_i, qbar_j), kbar = mean_c(AdaptiveMaxPool2d(h//4,w//4)(k))."""
    T, C, h, w = q.shape
    q_ = F.adaptive_max_pool2d(q, (h // 4, w // 4)).mean(1).reshape(T, -1)
    k_ = F.adaptive_max_pool2d(k, (h // 4, w // 4)).mean(1).reshape(T, -1)
    return F.cosine_similarity(q_.unsqueeze(0), k_.unsqueeze(1), dim=-1)      # [i,j] = cos(k_i, q_j)


def qam_select(sim: Tensor, strive: Tensor, conf: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """ppmstereo.py:501-513.  sim, strive: (T,T); conf: (T,) = mean_hw(uncertainty).
    Returns (frame_score (T,T), mask (T,T) bool, strive_new)."""
    T = sim.shape[0]
    pen = torch.exp(-strive / (strive.sum(1, keepdim=True) + T))
    score = pen * sim + conf[None, :]
    idx = torch.argsort(score, dim=-1, descending=True)[:, :TOP_K]
    mask = torch.zeros_like(score, dtype=torch.bool).scatter_(1, idx, True)
    strive = strive.clone()
    strive[mask] += 1
    return score, mask, strive


def play_inputs(q: Tensor, key: Tensor, pe: Tensor, value: Tensor, score: Tensor, mask: Tensor, clip: int,
                mean: Optional[Tensor] = None):
    """ppmstereo.py:517-548.  q,key,value: (T,C,h,w); pe: (T,C).  Builds the (n,C)/(k*n,C) operands of clip
    ``clip``: Q = q_i + PE_i; frames J ascending; s_hat = score[i,J]/mean; K' = K_J*s_hat + PE_J; V = value_J.
    ``mean`` overrides the normaliser (the reference's .mean() at :533 also runs over batch elements)."""
    T, C, h, w = q.shape
    J = torch.nonzero(mask[clip]).flatten()
    s = score[clip, J]
    s_hat = s / (s.mean() if mean is None else mean)
    Q = (q[clip] + pe[clip][:, None, None]).reshape(C, -1).t()
    K = key[J] * s_hat[:, None, None, None] + pe[J][:, :, None, None]          # (k,C,h,w)
    K = K.permute(0, 2, 3, 1).reshape(-1, C)
    V = value[J].permute(0, 2, 3, 1).reshape(-1, C)
    return Q.contiguous(), K.contiguous(), V.contiguous(), J, s_hat


def softmax_scale(c: int = 128) -> float:
    """ppmstereo.py:494 -- c^-0.5 * log_12000(2c)."""
    return c ** -0.5 * math.log(2 * c, 12000)


def flash_attn_math(Q: Tensor, K: Tensor, V: Tensor, scale: float) -> Tensor:
    """flash_attn.flash_attn_func semantics at ppmstereo.py:550: bf16 operands, fp32 softmax and
    accumulation, bf16 result; returned as float (the call site's ``.float()``)."""
    Qb, Kb, Vb = (t.to(torch.bfloat16).float() for t in (Q, K, V))
    out = torch.empty(Q.shape[0], V.shape[1])
    step = 2048
    for s in range(0, Q.shape[0], step):
        P = torch.softmax((Qb[s:s + step] @ Kb.t()) * scale, dim=-1)
        out[s:s + step] = P @ Vb
    return out.to(torch.bfloat16).float()


# --------------------------------------------------------------------------------------
# update block (models/core/ppmtereo_update.py)
# --------------------------------------------------------------------------------------
def _c2(W, pre, x, pad=0, groups=1):
    return F.conv2d(x, W[pre + ".weight"], W.get(pre + ".bias"), padding=pad, groups=groups)


def _c3(W, pre, x, pad):
    return F.conv3d(x, W[pre + ".weight"], W.get(pre + ".bias"), padding=pad)


def pcblock(W: Dict[str, Tensor], pre: str, x: Tensor) -> Tensor:
    """PCBlock4_Deep_nopool_res.forward, ppmtereo_update.py:1024-1030 (k_conv=[1,7])."""
    y = _c2(W, pre + ".ffn1.2", F.gelu(_c2(W, pre + ".ffn1.0", x)))
    x = F.gelu(x + y)
    c = x.shape[1]
    x = F.gelu(x + _c2(W, pre + ".conv_list.0", x, 0, groups=c))
    x = F.gelu(x + _c2(W, pre + ".conv_list.1", x, 3, groups=c))
    x = F.gelu(x + _c2(W, pre + ".pw", x))
    return _c2(W, pre + ".ffn2.2", F.gelu(_c2(W, pre + ".ffn2.0", x)))


def motion_encoder(W, flow: Tensor, corr: Tensor, mhs: Optional[Tensor], inp: Tensor):
    """BasicMotionEncoder_v2.forward, ppmtereo_update.py:466-482.  Returns (mf (N,128), mhs (N,64))."""
    p = "encoder."
    if mhs is None:
        mhs = _c2(W, p + "init_conv.2", F.relu(_c2(W, p + "init_conv.0", inp, 1)), 1)
    cor = F.gelu(pcblock(W, p + "convc1", corr))
    cor = F.relu(_c2(W, p + "convc2", cor, 1))
    flo = F.relu(_c2(W, p + "convf1", flow, 3))
    flo = F.relu(_c2(W, p + "convf2", flo, 1))
    out = F.relu(_c2(W, p + "final_conv", torch.cat([cor, flo, mhs], 1), 1))
    out, mhs = torch.split(out, [126, 64], 1)
    return torch.cat([out, flow], 1), mhs


def get_motion_and_value(W, flow, corr, mhs, inp):
    """SequenceUpdateBlock3D.get_motion_and_value, ppmtereo_update.py:945-950."""
    mf, mhs = motion_encoder(W, flow, corr, mhs, inp)
    value = F.conv2d(mf, W["aggregator.to_v.weight"])
    return mf, mhs, value


def get_uncertainty(W, x: Tensor) -> Tensor:
    """ppmtereo_update.py:889-893,936-938 -- sigmoid(conv1x1(relu(conv3x3(x))))."""
    return torch.sigmoid(_c2(W, "uncertainty.2", F.relu(_c2(W, "uncertainty.0", x, 1))))


def gru3d(W, h: Tensor, x: Tensor) -> Tensor:
    """SKSepConvGRU3D.forward, ppmtereo_update.py:291-312.  h (b,128,t,H,W), x (b,384,t,H,W)."""
    g = "gru."
    hx = torch.cat([h, x], 1)
    z = torch.sigmoid(_c3(W, g + "convz1.2", F.gelu(_c3(W, g + "convz1.0", hx, (0, 0, 7))), (0, 0, 2)))
    r = torch.sigmoid(_c3(W, g + "convr1.2", F.gelu(_c3(W, g + "convr1.0", hx, (0, 0, 7))), (0, 0, 2)))
    q = torch.tanh(_c3(W, g + "convq1", torch.cat([r * h, x], 1), (0, 0, 2)))
    h = (1 - z) * h + z * q
    for n, pad in (("2", (0, 2, 0)), ("3", (2, 0, 0))):
        hx = torch.cat([h, x], 1)
        z = torch.sigmoid(_c3(W, g + "convz" + n, hx, pad))
        r = torch.sigmoid(_c3(W, g + "convr" + n, hx, pad))
        q = torch.tanh(_c3(W, g + "convq" + n, torch.cat([r * h, x], 1), pad))
        h = (1 - z) * h + z * q
    return h


def flow_head3d(W, x: Tensor) -> Tensor:
    """FlowHead3D.forward, ppmtereo_update.py:670-678."""
    return _c3(W, "flow_head.conv2", F.relu(_c3(W, "flow_head.conv1", x, 1)), 1)


def layer_norm(x, w, b):
    return F.layer_norm(x, (x.shape[-1],), w, b, 1e-5)


def time_attn(W, x: Tensor, T: int, pre: str = "time_attn.") -> Tensor:
    """TimeAttnBlock.forward ppmtereo_update.py:603-618 with Attention.forward :409-420
    (q = k = v = LayerNorm(x) split in 8 heads; the qkv Linear is never applied)."""
    BT, C, h, w = x.shape
    b = BT // T
    tok = x.reshape(b, T, C, h, w).permute(0, 3, 4, 1, 2).reshape(b * h * w, T, C)
    y = layer_norm(tok, W[pre + "temporal_norm1.weight"], W[pre + "temporal_norm1.bias"])
    nh = 8
    qkv = y.reshape(-1, T, nh, C // nh).permute(0, 2, 1, 3)
    att = torch.softmax((qkv @ qkv.transpose(-2, -1)) * (C // nh) ** -0.5, dim=-1)
    o = (att @ qkv).transpose(1, 2).reshape(-1, T, C)
    o = F.linear(o, W[pre + "temporal_attn.proj.weight"], W[pre + "temporal_attn.proj.bias"])
    o = F.linear(o, W[pre + "temporal_fc.weight"], W[pre + "temporal_fc.bias"])
    tok = tok + o
    return tok.reshape(b, h, w, T, C).permute(0, 3, 4, 1, 2).reshape(BT, C, h, w)


def loftr_layer(W, pre: str, x: Tensor, src: Tensor, nhead: int = 8) -> Tensor:
    """LoFTREncoderLayer.forward + LinearAttention.forward, models/core/attention.py:164-190,73-100."""
    N, L, C = x.shape
    d = C // nhead
    q = F.linear(x, W[pre + "q_proj.weight"]).view(N, -1, nhead, d)
    k = F.linear(src, W[pre + "k_proj.weight"]).view(N, -1, nhead, d)
    v = F.linear(src, W[pre + "v_proj.weight"]).view(N, -1, nhead, d)
    Q = F.elu(q) + 1
    K = F.elu(k) + 1
    S = v.shape[1]
    v = v / S
    KV = torch.einsum("nshd,nshv->nhdv", K, v)
    Z = 1 / (torch.einsum("nlhd,nhd->nlh", Q, K.sum(1)) + 1e-6)
    msg = torch.einsum("nlhd,nhdv,nlh->nlhv", Q, KV, Z) * S
    msg = F.linear(msg.reshape(N, -1, C), W[pre + "merge.weight"])
    msg = layer_norm(msg, W[pre + "norm1.weight"], W[pre + "norm1.bias"])
    msg = F.linear(F.relu(F.linear(torch.cat([x, msg], 2), W[pre + "mlp.0.weight"])), W[pre + "mlp.2.weight"])
    msg = layer_norm(msg, W[pre + "norm2.weight"], W[pre + "norm2.bias"])
    return x + msg


def space_attn(W, x: Tensor) -> Tensor:
    """SpaceAttnBlock.forward ppmtereo_update.py:626-631."""
    BT, C, h, w = x.shape
    tok = x.reshape(BT, C, h * w).transpose(1, 2)
    tok = loftr_layer(W, "space_attn.encoder_layer.", tok, tok)
    return tok.transpose(1, 2).reshape(BT, C, h, w)


def update_block_forward(W, net, inp, mf, mfg, t: int, with_attention: bool):
    """SequenceUpdateBlock3D.forward, ppmtereo_update.py:971-1003.  The mask head follows the weights: "mask_3d.*"
    present = use_convex_3d=True (:993-996: 3x3x3 conv 128->256, ReLU, 1x1x1 conv 256->432 on the (b,c,t,h,w) state)."""
    x = torch.cat([inp, mf, mfg], 1)
    if with_attention:
        x = time_attn(W, x, t)
        x = space_attn(W, x)
    BT, _, h, w = net.shape
    b = BT // t
    to5 = lambda a: a.reshape(b, t, -1, h, w).permute(0, 2, 1, 3, 4)
    to4 = lambda a: a.permute(0, 2, 1, 3, 4).reshape(BT, -1, h, w)
    net5 = gru3d(W, to5(net), to5(x))
    dflow = to4(flow_head3d(W, net5))
    net = to4(net5)
    if "mask_3d.0.weight" in W:
        mask = to4(0.25 * _c3(W, "mask_3d.2", F.relu(_c3(W, "mask_3d.0", net5, 1)), 0))
    else:
        mask = 0.25 * _c2(W, "mask_2d.2", F.relu(_c2(W, "mask_2d.0", net, 1)))
    return net, mask, dflow


def convex_upsample(flow: Tensor, mask: Tensor, rate: int = 4) -> Tensor:
    """PPMStereo.convex_upsample, ppmstereo.py:185-197 -- closed form
    out[n,c,4y+i,4x+j] = sum_k softmax_k(mask[n,16k+4i+j,y,x]) * 4 flow[n,c,y+k//3-1,x+k%3-1]."""
    N, _, H, W = flow.shape
    m = torch.softmax(mask.view(N, 1, 9, rate, rate, H, W), dim=2)
    fp = F.pad(rate * flow, (1, 1, 1, 1))
    nb = torch.stack([fp[:, :, dy:dy + H, dx:dx + W] for dy in range(3) for dx in range(3)], 2)  # (N,2,9,H,W)
    up = (m * nb.view(N, 2, 9, 1, 1, H, W)).sum(2)                     # (N,2,r,r,H,W)
    return up.permute(0, 1, 4, 2, 5, 3).reshape(N, 2, rate * H, rate * W)


def convex_upsample_3d(flow: Tensor, mask: Tensor, rate: int, T: int) -> Tensor:
    """PPMStereo.convex_upsample_3d, ppmstereo.py:199-228, with unfoldNd.UnfoldNd([3,3,3], padding=1) (absent third-party
    module, un-pinned) restated from its published semantics (N-d generalisation of nn.Unfold: channel c*27 + k,
    k = (kt,kh,kw) row-major, zero padding) -- closed form
    out[(b t),c,4y+i,4x+j] = sum_{k<27} softmax_k(mask[(b t),16k+4i+j,y,x]) * 4 flow[(b t+kt-1),c,y+ky-1,x+kx-1]."""
    BT, _, H, W = flow.shape
    b = BT // T
    f5 = (rate * flow).reshape(b, T, 2, H, W).permute(0, 2, 1, 3, 4)               # (b,2,T,H,W)
    m = torch.softmax(mask.reshape(b, T, 27, rate, rate, H, W), dim=2)              # (b,T,27,r,r,H,W)
    fp = F.pad(f5, (1, 1, 1, 1, 1, 1))
    nb = torch.stack([fp[:, :, a:a + T, y:y + H, x:x + W] for a in range(3) for y in range(3) for x in range(3)], 2)   # (b,2,27,T,H,W)
    up = (m.permute(0, 2, 3, 4, 1, 5, 6)[:, None] * nb[:, :, :, None, None]).sum(2)                                     # (b,2,r,r,T,H,W)
    up = up.permute(0, 1, 4, 5, 2, 6, 3).reshape(b, 2, T, rate * H, rate * W)
    return up.permute(0, 2, 1, 3, 4).reshape(BT, 2, rate * H, rate * W)


def interp(x: Tensor, size) -> Tensor:
    """models/core/utils/utils.py:10-16."""
    return F.interpolate(x, size=size, mode="bilinear", align_corners=True)


# --------------------------------------------------------------------------------------
# the hot loop
# --------------------------------------------------------------------------------------
def forward_update_block(Wb, Watt, pyr, flow, net, inp, mhs, iters: int, interp_scale: int, t: int,
                         with_attention: bool, predictions: list, uncertainties: list, trace: Optional[list] = None):
    """PPMStereo.forward_update_block, ppmstereo.py:426-594 (live branches only: interp_scale in {4,2,1}).

    Wb: update-block weights, Watt: {"to_qk.weight"}, pyr: output of corr_pyramid.  b == 1 layouts
    generalise to b > 1 exactly as the reference does (frames of all batch elements share QAM means).
    """
    BT, c, h, w = inp.shape
    b = BT // t
    qk = F.conv2d(inp, Watt["to_qk.weight"])
    query, key = qk[:, :c], qk[:, c:]                                   # (BT,c,h,w) each; frame index = b*t + ti
    pe = temporal_pe(t, c)
    scale = softmax_scale(c)
    sims = [qk_similarity(query[bi * t:(bi + 1) * t], key[bi * t:(bi + 1) * t]) for bi in range(b)]
    strive = [torch.ones_like(s) for s in sims]
    beta = Wb["aggregator.beta"]
    flow_out = None
    for itr in range(iters):
        out_corrs = corr_lookup(pyr, flow)
        mf, mhs, value = get_motion_and_value(Wb, flow, out_corrs, mhs, inp)
        unc = get_uncertainty(Wb, torch.cat([net, value], 1))
        conf = unc.reshape(b, t, -1).mean(-1)
        scores, masks = [], []
        for bi in range(b):
            sc, mk, strive[bi] = qam_select(sims[bi], strive[bi], conf[bi])
            scores.append(sc)
            masks.append(mk)
        # selected_score.mean() (ppmstereo.py:533) averages over batch elements too
        mfg = torch.empty_like(mf)
        for clip in range(t):
            mean_all = torch.cat([scores[bi][clip][masks[bi][clip]] for bi in range(b)]).mean()
            for bi in range(b):
                sl = slice(bi * t, (bi + 1) * t)
                Q, K, V, J, _ = play_inputs(query[sl], key[sl], pe, value[sl], scores[bi], masks[bi], clip, mean_all)
                hid = flash_attn_math(Q, K, V, scale)                   # (n,c)
                hid = hid.t().reshape(c, h, w)
                mfg[bi * t + clip] = mf[bi * t + clip] + beta * hid
        net, up_mask, dflow = update_block_forward(Wb, net, inp, mf, mfg, t, with_attention)
        flow = flow + dflow
        if up_mask.shape[1] == 16 * 27:                                     # use_convex_3d=True (ppmstereo.py:573-574)
            flow_out = convex_upsample_3d(flow, up_mask, 4, t)
        else:
            flow_out = convex_upsample(flow, up_mask, 4)
        unc_up = F.interpolate(unc, scale_factor=4 * interp_scale, mode="bilinear")
        flow_up = flow_out
        if interp_scale > 1:
            flow_up = interp_scale * interp(flow_out, (interp_scale * flow_out.shape[2], interp_scale * flow_out.shape[3]))
        predictions.append(flow_up[:, :1])
        uncertainties.append(unc_up)
        if trace is not None:
            trace.append(dict(corr=out_corrs, mf=mf, value=value, unc=unc, mfg=mfg, net=net, mask=up_mask,
                              dflow=dflow, flow=flow, flow_out=flow_out,
                              score=torch.stack(scores), sel=torch.stack(masks)))
    return flow_out, net, mhs


def cascade(W, feats, iters: int, t: int, predictions: Optional[list] = None, uncertainties: Optional[list] = None):
    """The three-scale cascade of PPMStereo.forward, ppmstereo.py:696-804 (from encoder outputs on).

    feats: dict with fmap1/fmap2 at scales 16, 8, 4 ("f1_16", "f2_16", ...) and net/inp ("net_16", "inp_16", ...),
    i.e. everything the encoders + SST block hand to the loop.  Returns (flow_up (BT,1,H,W), unc_up (BT,1,H,W)).
    """
    preds = [] if predictions is None else predictions
    uncs = [] if uncertainties is None else uncertainties
    f16 = feats["f1_16"]
    flow16 = torch.zeros(f16.shape[0], 2, f16.shape[2], f16.shape[3])
    fo, net16, mhs16 = forward_update_block(W["update_block16"], W["att.0"], corr_pyramid(feats["f1_16"], feats["f2_16"]),
                                            flow16, feats["net_16"], feats["inp_16"], None, iters // 2, 4, t, True, preds, uncs)
    h8, w8 = feats["f1_8"].shape[2:]
    flow8 = -(h8 / fo.shape[2]) * interp(fo, (h8, w8))
    mhs8 = F.interpolate(mhs16, scale_factor=2, mode="bilinear", align_corners=True)
    net8 = (feats["net_8"] + interp(net16, (2 * net16.shape[2], 2 * net16.shape[3]))) / 2.0
    fo, net8, mhs8 = forward_update_block(W["update_block08"], W["att.1"], corr_pyramid(feats["f1_8"], feats["f2_8"]),
                                          flow8, net8, feats["inp_8"], mhs8, iters // 2, 2, t, False, preds, uncs)
    h4, w4 = feats["f1_4"].shape[2:]
    flow4 = -(h4 / fo.shape[2]) * interp(fo, (h4, w4))
    mhs4 = F.interpolate(mhs8, scale_factor=2, mode="bilinear", align_corners=True)
    net4 = (feats["net_4"] + interp(net8, (2 * net8.shape[2], 2 * net8.shape[3]))) / 2.0
    forward_update_block(W["update_block04"], W["att.2"], corr_pyramid(feats["f1_4"], feats["f2_4"]),
                         flow4, net4, feats["inp_4"], mhs4, iters, 1, t, False, preds, uncs)
    return preds[-1], uncs[-1]


def window_plan(num_ims: int, kernel_size: int = 20):
    """Index-only restatement of forward_batch_test's sliding window, ppmstereo.py:242-310.
    Returns [(start, stop, keep_from, keep_to)] in window-local indices for every window whose output is kept."""
    stride = kernel_size // 2
    if kernel_size > num_ims:
        return [(0, num_ims, 0, num_ims)]
    plan = []
    for i in range(0, num_ims, stride):
        n = min(i + kernel_size, num_ims) - i
        if plan and n >= stride:
            if n < kernel_size:
                plan.append((i, i + n, stride // 2, n))
            else:
                plan.append((i, i + n, stride // 2, n + (-stride // 2)))
        elif not plan:
            plan.append((i, i + n, 0, n + (-stride // 2)))
    return plan


# --------------------------------------------------------------------------------------
# caller-side glue needed to drive / check the path end to end (not on the hot path itself)
# --------------------------------------------------------------------------------------
def position_encoding_sine(d_model: int, h: int, w: int) -> Tensor:
    """PositionEncodingSine (temp_bug_fix=True), models/core/attention.py:23-64 -> (d_model,h,w)."""
    y = torch.ones(h, w).cumsum(0)[None]
    x = torch.ones(h, w).cumsum(1)[None]
    div = torch.exp(torch.arange(0, d_model // 2, 2).float() * (-math.log(10000.0) / (d_model // 2)))[:, None, None]
    pe = torch.zeros(d_model, h, w)
    pe[0::4], pe[1::4] = torch.sin(x * div), torch.cos(x * div)
    pe[2::4], pe[3::4] = torch.sin(y * div), torch.cos(y * div)
    return pe


def sst_block(W: Dict[str, Tensor], f1: Tensor, f2: Tensor, T: int, depth: int = 4, num_frames: int = 5):
    """PPMStereo.forward_sst_block with attention_type "self_stereo_temporal_update_time_update_space", ppmstereo.py:322-395, on the 1/16
    features (BT, 256, h, w) of both views: + PositionEncodingSine, + time_embed (nearest-interpolated along T when T != num_frames,
    :347-352), then depth x { "self" LoFTR layer on each view, "cross" layer (the second call sees the UPDATED first view,
    attention.py:230-232), TimeAttnBlock(256) on each view }.  W: keys of ppmstereo_amd.weights.sst_param_shapes()."""
    BT, C, h, w = f1.shape
    pe = position_encoding_sine(C, h, w)
    te = W["time_embed"]
    if T != num_frames:
        te = F.interpolate(te.transpose(1, 2), size=T, mode="nearest").transpose(1, 2).contiguous()
    b = BT // T
    add_te = lambda x: (x.reshape(b, T, C, h, w) + te[0][None, :, :, None, None]).reshape(BT, C, h, w)
    x0, x1 = add_te(f1 + pe), add_te(f2 + pe)
    tok = lambda x: x.reshape(BT, C, h * w).transpose(1, 2)
    img = lambda t: t.transpose(1, 2).reshape(BT, C, h, w)
    for i in range(depth):
        p = f"self_attn_blocks.{i}.layers.0."
        t0, t1 = tok(x0), tok(x1)
        t0, t1 = loftr_layer(W, p, t0, t0), loftr_layer(W, p, t1, t1)
        p = f"cross_attn_blocks.{i}.layers.0."
        t0 = loftr_layer(W, p, t0, t1)
        t1 = loftr_layer(W, p, t1, t0)
        x0, x1 = img(t0), img(t1)
        p = f"time_attn_blocks.{i}."
        x0, x1 = time_attn(W, x0, T, p), time_attn(W, x1, T, p)
    return x0, x1


def pre_loop_glue(fmap1: Tensor, fmap2: Tensor, c4: Tensor, c8: Tensor, c16: Tensor, sst_fn=None, hdim: int = 128):
    """PPMStereo.forward between the encoders and the loop, ppmstereo.py:620-682.
    sst_fn(f1_16, f2_16) stands for forward_sst_block (:322-395); None = positional encoding only
    (what the reference does with attention_type=None)."""
    feats = {}
    net, inp = torch.split(fmap1, [hdim, hdim], 1)
    feats["net_4"] = torch.tanh((net + c4[:, :hdim]) / 2.0)
    feats["inp_4"] = F.relu((inp + c4[:, hdim:]) / 2.0)
    h, w = fmap1.shape[2:]
    f1_16, f2_16 = F.avg_pool2d(fmap1, 4, 4), F.avg_pool2d(fmap2, 4, 4)
    if sst_fn is None:
        pe = position_encoding_sine(fmap1.shape[1], f1_16.shape[2], f1_16.shape[3])
        f1_16, f2_16 = f1_16 + pe, f2_16 + pe
    else:
        f1_16, f2_16 = sst_fn(f1_16, f2_16)
    n16, i16 = torch.split(f1_16, [hdim, hdim], 1)
    feats["net_16"] = torch.tanh((n16 + c16[:, :hdim]) / 2.0)
    feats["inp_16"] = F.relu((i16 + c16[:, hdim:]) / 2.0)
    f1_8 = (F.avg_pool2d(fmap1, 2, 2) + interp(f1_16, (h // 2, w // 2))) / 2.0
    f2_8 = (F.avg_pool2d(fmap2, 2, 2) + interp(f2_16, (h // 2, w // 2))) / 2.0
    n8, i8 = torch.split(f1_8, [hdim, hdim], 1)
    feats["net_8"] = torch.tanh((n8 + c8[:, :hdim]) / 2.0)
    feats["inp_8"] = F.relu((i8 + c8[:, hdim:]) / 2.0)
    feats.update(f1_16=f1_16, f2_16=f2_16, f1_8=f1_8, f2_8=f2_8, f1_4=fmap1, f2_4=fmap2)
    return feats


# ---------------------------------------------------------------------------------------------------------------------
# fnet: BasicEncoder(output_dim=256, norm_fn="instance") -- the producer of fmap1 / fmap2
def _instance_norm(x: Tensor) -> Tensor:
    """nn.InstanceNorm2d(planes, affine=False): per (sample, channel) over H x W, biased variance, eps 1e-5 (models/core/extractor.py:326-329, 364)."""
    return F.instance_norm(x, eps=1e-5)


def residual_block(W: Dict[str, Tensor], pre: str, x: Tensor, stride: int) -> Tensor:
    """ResidualBlock.forward, models/core/extractor.py:337-345: the 1x1 projection + norm on the skip is applied ALWAYS (also at
    stride 1 with equal planes), its norm is norm3 (:339-341)."""
    y = F.relu(_instance_norm(F.conv2d(x, W[pre + "conv1.weight"], W[pre + "conv1.bias"], stride=stride, padding=1)))
    y = F.relu(_instance_norm(F.conv2d(y, W[pre + "conv2.weight"], W[pre + "conv2.bias"], padding=1)))
    x = _instance_norm(F.conv2d(x, W[pre + "downsample.0.weight"], W[pre + "downsample.0.bias"], stride=stride))
    return F.relu(x + y)


def basic_encoder(W: Dict[str, Tensor], x):
    """BasicEncoder.forward, models/core/extractor.py:391-423 with the ctor of :349-389 (conv1 7x7 s2 p3, layer1 64 s1, layer2 96 s2,
    layer3 128 s1, conv2 1x1 -> 256; dropout 0).  x: (N, 3, H, W) or a pair of such (batch concatenation :398-401, split back :419-420)."""
    is_list = isinstance(x, (tuple, list))
    if is_list:
        x = torch.cat(list(x), dim=0)
    x = F.relu(_instance_norm(F.conv2d(x, W["conv1.weight"], W["conv1.bias"], stride=2, padding=3)))
    for layer, stride in ((1, 1), (2, 2), (3, 1)):
        x = residual_block(W, f"layer{layer}.0.", x, stride)
        x = residual_block(W, f"layer{layer}.1.", x, 1)
    x = F.conv2d(x, W["conv2.weight"], W["conv2.bias"])
    if is_list:
        return torch.split(x, x.shape[0] // 2, dim=0)
    return x


# ---------------------------------------------------------------------------------------------------------------------
# cnet: Feature("tiny", 256) = frozen ConvNeXt-V2-tiny + FPN decoder
CNET_DIMS, CNET_DEPTHS = (96, 192, 384, 768), (3, 3, 9, 3)


def _ln_cf(x: Tensor, w: Tensor, b: Tensor) -> Tensor:
    """LayerNorm(data_format="channels_first", eps=1e-6), models/core/convnext.py:29-34 (normalises over C of an NCHW tensor)."""
    u = x.mean(1, keepdim=True)
    s = (x - u).pow(2).mean(1, keepdim=True)
    return w[:, None, None] * ((x - u) / torch.sqrt(s + 1e-6)) + b[:, None, None]


def convnext_block(W: Dict[str, Tensor], p: str, x: Tensor) -> Tensor:
    """Block.forward, models/core/convnext.py:67-79 (drop_path = identity): dw7x7 -> LN (channels_last, eps 1e-6) -> Linear 4x ->
    GELU (exact) -> GRN (:44-47) -> Linear -> + input."""
    C = x.shape[1]
    y = F.conv2d(x, W[p + "dwconv.weight"], W[p + "dwconv.bias"], padding=3, groups=C).permute(0, 2, 3, 1)
    y = F.layer_norm(y, (C,), W[p + "norm.weight"], W[p + "norm.bias"], 1e-6)
    y = F.gelu(F.linear(y, W[p + "pwconv1.weight"], W[p + "pwconv1.bias"]))
    gx = torch.norm(y, p=2, dim=(1, 2), keepdim=True)
    nx = gx / (gx.mean(dim=-1, keepdim=True) + 1e-6)
    y = W[p + "grn.gamma"] * (y * nx) + W[p + "grn.beta"] + y
    y = F.linear(y, W[p + "pwconv2.weight"], W[p + "pwconv2.bias"])
    return x + y.permute(0, 3, 1, 2)


def feature_cnet(W: Dict[str, Tensor], x: Tensor):
    """Feature.forward, models/core/convnext.py:256-264, with ConvNeXtV2.forward_features (:133-139): stem conv 4x4 s4 + LN, three
    LN + conv 2x2 s2 downsamplers, stages of 3/3/9/3 blocks -> x4, x8, x16, x32; decoder (:225-253): up* = nearest x2 -> conv3x3 ->
    InstanceNorm -> ReLU, decode* = conv1x1 -> InstanceNorm -> ReLU -> conv3x3 on cat[skip, up].  x: (N, 3, H, W), H, W multiples of 32."""
    d = "convnext.downsample_layers."
    feats = []
    y = _ln_cf(F.conv2d(x, W[d + "0.0.weight"], W[d + "0.0.bias"], stride=4), W[d + "0.1.weight"], W[d + "0.1.bias"])
    for i in range(4):
        if i > 0:
            y = F.conv2d(_ln_cf(y, W[d + f"{i}.0.weight"], W[d + f"{i}.0.bias"]), W[d + f"{i}.1.weight"], W[d + f"{i}.1.bias"], stride=2)
        for j in range(CNET_DEPTHS[i]):
            y = convnext_block(W, f"convnext.stages.{i}.{j}.", y)
        feats.append(y)
    x4, x8, x16, x32 = feats

    def up(tag, t):
        t = F.interpolate(t, scale_factor=2.0, mode="nearest")
        return F.relu(_instance_norm(F.conv2d(t, W[tag + ".1.weight"], W[tag + ".1.bias"], padding=1)))

    def dec(tag, t):
        t = F.relu(_instance_norm(F.conv2d(t, W[tag + ".0.weight"], W[tag + ".0.bias"])))
        return F.conv2d(t, W[tag + ".3.weight"], W[tag + ".3.bias"], padding=1)

    x16 = dec("decode_16x", torch.cat([x16, up("upconv_16", x32)], 1))
    x8 = dec("decode_8x", torch.cat([x8, up("upconv_8", x16)], 1))
    x4 = dec("decode_4x", torch.cat([x4, up("upconv_4", x8)], 1))
    return x4, x8, x16


class InputPadder:
    """models/core/utils/utils.py:19-44 (mode "sintel"): replicate-pad H, W up to multiples of divis_by, split evenly."""

    def __init__(self, dims, divis_by: int = 8):
        self.ht, self.wd = dims[-2:]
        pad_ht = (((self.ht // divis_by) + 1) * divis_by - self.ht) % divis_by
        pad_wd = (((self.wd // divis_by) + 1) * divis_by - self.wd) % divis_by
        self._pad = [pad_wd // 2, pad_wd - pad_wd // 2, pad_ht // 2, pad_ht - pad_ht // 2]

    def pad(self, *inputs):
        return [F.pad(x, self._pad, mode="replicate") for x in inputs]

    def unpad(self, x):
        ht, wd = x.shape[-2:]
        return x[..., self._pad[2]:ht - self._pad[3], self._pad[0]:wd - self._pad[1]]


def forward(W, fnet, cnet, image1: Tensor, image2: Tensor, iters: int, sst_fn=None):
    """PPMStereo.forward(test_mode=True), ppmstereo.py:601-804, batch 1: image (1,T,3,H,W) in [0,255].  fnet / cnet stand
    for the encoders (extractor.py / convnext.py, outside the path): fnet([im1, im2]) -> (fmap1, fmap2) (T,256,H/4,W/4),
    cnet(im1) -> context at 1/4, 1/8, 1/16.  Returns (flow_up (1,T,1,H,W), uncertainty (1,T,1,H,W))."""
    T = image1.shape[1]
    im1 = (2 * (image1 / 255.0) - 1.0)[0].contiguous()
    im2 = (2 * (image2 / 255.0) - 1.0)[0].contiguous()
    fmap1, fmap2 = fnet([im1, im2])
    c4, c8, c16 = cnet(im1)
    feats = pre_loop_glue(fmap1, fmap2, c4, c8, c16, sst_fn)
    disp, unc = cascade(W, feats, iters, T)
    return disp[None], unc[None]


def forward_batch_test(W, fnet, cnet, video: Tensor, kernel_size: int = 20, iters: int = 20, sst_fn=None) -> Dict[str, Tensor]:
    """PPMStereo.forward_batch_test, ppmstereo.py:238-320: video (N,2,3,H,W); InputPadder(divis_by=32) per window, windows
    of kernel_size frames every kernel_size//2, centre frames kept (:296-307), outputs .abs()[:, :1] (:309-310)."""
    stride = kernel_size // 2
    num_ims = len(video)

    def run(lo, hi):
        left, right = video[lo:hi, 0], video[lo:hi, 1]
        padder = InputPadder(left.shape, divis_by=32)
        left, right = padder.pad(left, right)
        d, u = forward(W, fnet, cnet, left[None], right[None], iters, sst_fn)
        return padder.unpad(d[0])[:, None], padder.unpad(u[0])[:, None]

    if kernel_size > num_ims:
        d, u = run(0, num_ims)
        return {"disparity": d.squeeze(1).abs()[:, :1], "uncertainties": u.squeeze(1).abs()[:, :1]}
    disp_preds, uncertainties = [], []
    for i in range(0, num_ims, stride):
        d, u = run(i, min(i + kernel_size, num_ims))       # (the reference also runs the trailing windows it then discards)
        if len(disp_preds) > 0 and len(d) >= stride:
            if len(d) < kernel_size:
                disp_preds.append(d[stride // 2:])
                uncertainties.append(u[stride // 2:])
            else:
                disp_preds.append(d[stride // 2: -stride // 2])
                uncertainties.append(u[stride // 2: -stride // 2])
        elif len(disp_preds) == 0:
            disp_preds.append(d[: -stride // 2])
            uncertainties.append(u[: -stride // 2])
    return {"disparity": torch.cat(disp_preds).squeeze(1).abs()[:, :1], "uncertainties": torch.cat(uncertainties).squeeze(1).abs()[:, :1]}
