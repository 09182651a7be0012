"""TimeAttnBlock + SpaceAttnBlock of update_block16 on the channel-last activation x = [inp, mf, mfg]
(/root/reference/models/core/ppmtereo_update.py:593-631 with Attention :400-420 and the LoFTR linear-attention
layer /root/reference/models/core/attention.py:73-100,164-190).

These two blocks exist only at the 1/16 scale (640 pixels per frame at 320x512; 3.6 MFLOP/px, 0.05 % of the
loop's FLOPs) and are made of LayerNorms, (T x T) / (48 x 48) per-head products and five 384-wide Linear layers.
They run as fp32 torch-ROCm tensor ops on the device-resident tensor; everything
else on the path is hand-written HIP.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F


def _ln(x, w, b):
    return F.layer_norm(x, (x.shape[-1],), w, b, 1e-5)


def time_space_attention(W: Dict[str, torch.Tensor], x: torch.Tensor, T: int, h: int, w: int, nhead: int = 8) -> torch.Tensor:
    """x: (T*h*w, C) fp32, pixel-major per frame.  Returns the same layout."""
    P, Cc = x.shape
    n = h * w
    # ---- time attention: tokens = the T frames of one pixel; q = k = v = LN(x) split in heads (no qkv projection)
    tok = x.view(T, n, Cc).transpose(0, 1)                                   # (n, T, C)
    y = _ln(tok, W["time_attn.temporal_norm1.weight"], W["time_attn.temporal_norm1.bias"])
    d = Cc // nhead
    qkv = y.reshape(n, T, nhead, d).permute(0, 2, 1, 3)                        # (n, heads, T, d)
    att = torch.softmax((qkv @ qkv.transpose(-2, -1)) * d ** -0.5, dim=-1)
    o = (att @ qkv).transpose(1, 2).reshape(n, T, Cc)
    o = F.linear(o, W["time_attn.temporal_attn.proj.weight"], W["time_attn.temporal_attn.proj.bias"])
    o = F.linear(o, W["time_attn.temporal_fc.weight"], W["time_attn.temporal_fc.bias"])
    tok = tok + o
    x = tok.transpose(0, 1)                                                    # (T, n, C)
    # ---- space attention: LoFTR encoder layer with linear attention over the n pixels of each frame
    p = "space_attn.encoder_layer."
    q = F.linear(x, W[p + "q_proj.weight"]).view(T, n, nhead, d)
    k = F.linear(x, W[p + "k_proj.weight"]).view(T, n, nhead, d)
    v = F.linear(x, W[p + "v_proj.weight"]).view(T, n, nhead, d)
    Q = F.elu(q) + 1
    K = F.elu(k) + 1
    v = v / n
    KV = torch.einsum("nshd,nshv->nhdv", K, v)
    Z = 1 / (torch.einsum("nlhd,nhd->nlh", Q, K.sum(1)) + 1e-6)
    msg = torch.einsum("nlhd,nhdv,nlh->nlhv", Q, KV, Z) * n
    msg = F.linear(msg.reshape(T, n, Cc), W[p + "merge.weight"])
    msg = _ln(msg, W[p + "norm1.weight"], W[p + "norm1.bias"])
    msg = F.linear(F.relu(F.linear(torch.cat([x, msg], 2), W[p + "mlp.0.weight"])), W[p + "mlp.2.weight"])
    msg = _ln(msg, W[p + "norm2.weight"], W[p + "norm2.bias"])
    return (x + msg).reshape(P, Cc).contiguous()
