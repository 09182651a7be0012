"""Synthetic, RNG-free inputs for the hot path (identical on every box).

The loop consumes what the encoders + SST block produce in the reference
(/root/reference/models/core/ppmstereo.py:618-682): feature maps at 1/4, 1/8, 1/16 and the
tanh / relu'd hidden and context tensors.  Encoders are out of scope, so
benches and tests synthesise those tensors directly with the right shapes and statistics; the right
feature map is the left one shifted along the epipolar line plus noise, so that the correlation volume has
real peaks.
"""
from __future__ import annotations

from typing import Dict

import torch

from .weights import hash_normal


def synth_scale_inputs(T: int, h: int, w: int, seed: int, with_mhs: bool = True, c: int = 256,
                       shift: int = 3, frame_contrast: float = 0.0) -> Dict[str, torch.Tensor]:
    """Inputs of one ``forward_update_block`` call at one scale: fmap1/fmap2 (T,c,h,w), net/inp (T,128,h,w),
    flow (T,2,h,w), mhs (T,64,h,w) or None.

    frame_contrast > 0 multiplies the context features of frame t by a per-(frame, 4x4 block) gain, so that the frames'
    pooled q / k descriptors (ppmstereo.py:397-423) really differ: with i.i.d. noise all T x T frame similarities are
    equal to ~1e-5 and the QAM top-k pick (ppmstereo.py:505-513) is decided by rounding noise, which no two
    implementations share (matters once T >> top-k)."""
    f1 = hash_normal((T, c, h, w), seed * 16 + 1)
    noise = hash_normal((T, c, h, w), seed * 16 + 2)
    f2 = 0.8 * torch.roll(f1, shifts=-shift, dims=3) + 0.6 * noise
    net = torch.tanh(hash_normal((T, 128, h, w), seed * 16 + 3))
    inp = torch.relu(hash_normal((T, 128, h, w), seed * 16 + 4))
    if frame_contrast > 0.0:
        gain = (1.0 + frame_contrast * hash_normal((T, 1, (h + 3) // 4, (w + 3) // 4), seed * 16 + 7)).clamp_min(0.05)
        inp = inp * gain.repeat_interleave(4, 2).repeat_interleave(4, 3)[:, :, :h, :w]
    flow = hash_normal((T, 2, h, w), seed * 16 + 5, std=1.5)
    flow[:, 0] -= float(shift)
    mhs = torch.relu(hash_normal((T, 64, h, w), seed * 16 + 6)) if with_mhs else None
    return dict(fmap1=f1, fmap2=f2, net=net, inp=inp, flow=flow, mhs=mhs)


# T = 40 >> top-k parity cases (tools/gen_golden.py, tests): inputs whose QAM pick is well conditioned -- smallest gap between
# the 5th and 6th frame score over all clips and iterations 1.6e-4 / 4.9e-4 (seeds found by search with the CPU restatement under tests)
T40_CASES = {"fub04_T40": dict(seed=813, frame_contrast=1.0), "fub16_T40": dict(seed=903, frame_contrast=1.0)}


def synth_cascade_feats(T: int, H: int, W: int, seed: int = 7) -> Dict[str, torch.Tensor]:
    """Everything the loop needs at the three scales for a padded H x W clip (H, W multiples of 32):
    keys f1_s, f2_s, net_s, inp_s for s in (16, 8, 4)."""
    feats = {}
    for i, s in enumerate((16, 8, 4)):
        d = synth_scale_inputs(T, H // s, W // s, seed * 8 + i, with_mhs=False, shift=max(1, 12 // s))
        feats[f"f1_{s}"], feats[f"f2_{s}"] = d["fmap1"], d["fmap2"]
        feats[f"net_{s}"], feats[f"inp_{s}"] = d["net"], d["inp"]
    return feats
