"""The frame-sharded loop (ppmstereo_amd.dist.FrameShard) written with the CPU oracle's math for the per-rank compute: the same
exchange schedule the gfx950 engine runs (ppmstereo_amd/engine.py), so that `sharded == unsharded` can be shown on CPU ranks
over gloo.  Test infrastructure."""
import torch
import torch.nn.functional as F

from oracle import ppm_oracle as O

HALO = 2
DROP = None          # fault injection for the test of the test: name of an exchange to skip ("rh")


def _ext(shard, x, k):
    """(f,C,h,w) local frames -> (f + 2*HALO, C, h, w) with the +-k neighbour frames exchanged (zeros beyond the window ends)."""
    f = x.shape[0]
    buf = torch.zeros((f + 2 * HALO,) + tuple(x.shape[1:]), dtype=x.dtype)
    buf[HALO:HALO + f] = x
    if k > 0:
        shard.halo(buf, k)
    return buf


def _ext_many(shard, xs, k):
    """Several tensors through ONE batched halo exchange (the engine sends both bf16 planes of a tensor, or two tensors, in one batch)."""
    bufs = []
    for x in xs:
        f = x.shape[0]
        buf = torch.zeros((f + 2 * HALO,) + tuple(x.shape[1:]), dtype=x.dtype)
        buf[HALO:HALO + f] = x
        bufs.append(buf)
    shard.halo_many([(b, k) for b in bufs])
    return bufs


def _conv_t(W, name, x_ext, kt):
    """temporal conv (kt,1,1) of the reference on the halo'd block: zero padding beyond the halo, crop to the own frames."""
    x5 = x_ext.permute(1, 0, 2, 3)[None]                                   # (1,C,f+2H,h,w)
    y = F.conv3d(x5, W[name + ".weight"], W[name + ".bias"], padding=(kt // 2, 0, 0))
    return y[0].permute(1, 0, 2, 3)[HALO:-HALO]


def gru3d_sharded(shard, W, h, x):
    """SKSepConvGRU3D.forward (ppmtereo_update.py:291-312) on this rank's frames; passes W and H are per frame, pass T exchanges
    +-2 frames of [h | x] and then of r*h."""
    g = "gru."
    to5 = lambda a: a.permute(1, 0, 2, 3)[None]
    to4 = lambda a: a[0].permute(1, 0, 2, 3)
    c3 = lambda name, a, pad: to4(F.conv3d(to5(a), W[name + ".weight"], W[name + ".bias"], padding=pad))
    hx = torch.cat([h, x], 1)
    z = torch.sigmoid(c3(g + "convz1.2", F.gelu(c3(g + "convz1.0", hx, (0, 0, 7))), (0, 0, 2)))
    r = torch.sigmoid(c3(g + "convr1.2", F.gelu(c3(g + "convr1.0", hx, (0, 0, 7))), (0, 0, 2)))
    q = torch.tanh(c3(g + "convq1", torch.cat([r * h, x], 1), (0, 0, 2)))
    h = (1 - z) * h + z * q
    hx = torch.cat([h, x], 1)
    z = torch.sigmoid(c3(g + "convz2", hx, (0, 2, 0)))
    r = torch.sigmoid(c3(g + "convr2", hx, (0, 2, 0)))
    q = torch.tanh(c3(g + "convq2", torch.cat([r * h, x], 1), (0, 2, 0)))
    h = (1 - z) * h + z * q
    x_ext, h_ext = _ext_many(shard, [x, h], 2)                               # exchange 1: [h | x], one batch
    hx = torch.cat([h_ext, x_ext], 1)
    z = torch.sigmoid(_conv_t(W, g + "convz3", hx, 5))
    r = torch.sigmoid(_conv_t(W, g + "convr3", hx, 5))
    rh_ext = _ext(shard, r * h, 2 if DROP != "rh" else 0)                   # exchange 2: r * h
    q = torch.tanh(_conv_t(W, g + "convq3", torch.cat([rh_ext, x_ext], 1), 5))
    return (1 - z) * h + z * q


def _conv333(W, name, x_ext):
    y = F.conv3d(x_ext.permute(1, 0, 2, 3)[None], W[name + ".weight"], W[name + ".bias"], padding=(1, 1, 1))
    return y[0].permute(1, 0, 2, 3)


def update_block_sharded(shard, W, net, inp, mf, mfg, with_attention):
    """SequenceUpdateBlock3D.forward (ppmtereo_update.py:971-1003) on this rank's frames."""
    x = torch.cat([inp, mf, mfg], 1)
    if with_attention:                                                      # TimeAttnBlock: all T frames of a pixel
        xg = torch.empty((shard.T,) + tuple(x.shape[1:]))
        shard.gather_many([(x, xg)])                                        # direct all-gather (one grouped P2P batch)
        x = O.time_attn(W, xg, shard.T)[shard.lo:shard.hi]
        x = O.space_attn(W, x)                                              # per frame
    net = gru3d_sharded(shard, W, net, x)
    n_ext = _ext(shard, net, 1)                                             # exchange 3: new hidden state +-1
    hid = F.relu(_conv333(W, "flow_head.conv1", n_ext))                     # valid on frames [HALO-1+..]: computed on the halo'd block
    # the second 3x3x3 conv needs the 256-channel hidden layer of the +-1 frames: exchanged (the engine exchanges the 54 pre-gather
    # channels instead, which is the same sum reordered per tap)
    hid_own = hid[HALO:-HALO]
    h_ext = _ext(shard, hid_own, 1)                                         # exchange 4
    dflow = _conv333(W, "flow_head.conv2", h_ext)[HALO:-HALO]
    if "mask_3d.0.weight" in W:
        m1 = F.relu(_conv333(W, "mask_3d.0", n_ext))[HALO:-HALO]
        mask = 0.25 * F.conv3d(m1.permute(1, 0, 2, 3)[None], W["mask_3d.2.weight"], W["mask_3d.2.bias"])[0].permute(1, 0, 2, 3)
    else:
        mask = 0.25 * O._c2(W, "mask_2d.2", F.relu(O._c2(W, "mask_2d.0", net, 1)))
    return net, mask, dflow


def forward_update_block_sharded(shard, Wb, Watt, pyr, flow, net, inp, mhs, iters, interp_scale, with_attention, predictions, uncertainties):
    """PPMStereo.forward_update_block (ppmstereo.py:426-594) for this rank's contiguous block of frames of a T-frame window.
    pyr / flow / net / inp / mhs: local frames.  Exchanges: see ppmstereo_amd/dist.py."""
    T, lo, hi = shard.T, shard.lo, shard.hi
    f, c, h, w = inp.shape
    qk = F.conv2d(inp, Watt["to_qk.weight"])
    query, key = qk[:, :c], qk[:, c:]
    pe = O.temporal_pe(T, c)
    scale = O.softmax_scale(c)
    # frame descriptors of every frame (once per scale), then the same T x T similarity on every rank
    q_ = F.adaptive_max_pool2d(query, (h // 4, w // 4)).mean(1).reshape(f, -1)
    k_ = F.adaptive_max_pool2d(key, (h // 4, w // 4)).mean(1).reshape(f, -1)
    pooled, _ = shard.all_gather(torch.stack([q_, k_], 1))                  # (T, 2, cells)
    sim = F.cosine_similarity(pooled[:, 0].unsqueeze(0), pooled[:, 1].unsqueeze(1), dim=-1)
    strive = torch.ones_like(sim)
    key_all, _ = shard.all_gather(key)                                      # keys of every frame, fp32, once per scale
    beta = Wb["aggregator.beta"]
    flow_out = None
    for _ in range(iters):
        out_corrs = O.corr_lookup(pyr, flow)
        mf, mhs, value = O.get_motion_and_value(Wb, flow, out_corrs, mhs, inp)
        unc = O.get_uncertainty(Wb, torch.cat([net, value], 1))
        # ONE direct exchange: the T confidences and the values of every frame (the reference casts V to bf16, :550)
        conf, value_all = torch.empty(T), torch.empty((T,) + tuple(value.shape[1:]), dtype=torch.bfloat16)
        shard.gather_many([(value.to(torch.bfloat16), value_all), (unc.reshape(f, -1).mean(-1), conf)])
        score, mask, strive = O.qam_select(sim, strive, conf)
        value_all = value_all.float()
        mfg = torch.empty_like(mf)
        for li in range(f):
            clip = lo + li
            J = torch.nonzero(mask[clip]).flatten()
            s = score[clip, J]
            s_hat = s / s.mean()
            Q = (query[li] + pe[clip][:, None, None]).reshape(c, -1).t().contiguous()
            K = (key_all[J] * s_hat[:, None, None, None] + pe[J][:, :, None, None]).permute(0, 2, 3, 1).reshape(-1, c).contiguous()
            V = value_all[J].permute(0, 2, 3, 1).reshape(-1, c).contiguous()
            hid = O.flash_attn_math(Q, K, V, scale).t().reshape(c, h, w)
            mfg[li] = mf[li] + beta * hid
        net, up_mask, dflow = update_block_sharded(shard, Wb, net, inp, mf, mfg, with_attention)
        flow = flow + dflow
        if up_mask.shape[1] == 16 * 27:
            fl_ext = _ext(shard, flow, 1)                                   # convex_upsample_3d: +-1 frame of the flow
            m_ext = torch.zeros((f + 2 * HALO,) + tuple(up_mask.shape[1:]))
            m_ext[HALO:HALO + f] = up_mask
            flow_out = O.convex_upsample_3d(fl_ext, m_ext, 4, f + 2 * HALO)[HALO:-HALO]
        else:
            flow_out = O.convex_upsample(flow, up_mask, 4)
        unc_up = F.interpolate(unc, scale_factor=4 * interp_scale, mode="bilinear")
        flow_up = flow_out
        if interp_scale > 1:
            flow_up = interp_scale * O.interp(flow_out, (interp_scale * flow_out.shape[2], interp_scale * flow_out.shape[3]))
        predictions.append(flow_up[:, :1])
        uncertainties.append(unc_up)
    return flow_out, net, mhs
