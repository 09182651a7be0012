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


def synthetic_disparity(T: int, H: int, W: int) -> torch.Tensor:
    """The smooth disparity field of ``stereo_video``: d(x, y, t) = 8 + 6 sin(2 pi x / W + 0.3 t) cos(2 pi y / H)
    pixels, (T, H, W)."""
    import math
    t = torch.arange(T, dtype=torch.float32)[:, None, None]
    y = torch.arange(H, dtype=torch.float32)[None, :, None]
    x = torch.arange(W, dtype=torch.float32)[None, None, :]
    return 8.0 + 6.0 * torch.sin(2 * math.pi * x / W + 0.3 * t) * torch.cos(2 * math.pi * y / H)


def stereo_video(T: int, H: int, W: int, seed: int = 7, noise: float = 0.02) -> torch.Tensor:
    """SURVEY.md section 8(d)'s image-level input: float32 (T, 2, 3, H, W) in [0, 255], as the reference's data loader hands frames to
    ``forward_batch_test`` (datasets/dynamic_stereo_datasets.py:578-590).  Left view: integer-valued uniform hash noise (a counter hash
    of (seed, index), not torch's RNG: identical on every box); right view: the left one resampled along the epipolar line at
    x + d(x, y, t) with the smooth ``synthetic_disparity`` (a left pixel x shows at x - d in the right image; linear interpolation,
    replicated border) plus ``noise`` * 255 of hash noise -- so the correlation volume has real peaks at a known disparity."""
    from .weights import hash_uniform
    left = hash_uniform((T, 3, H, W), seed * 8 + 1, 0.0, 256.0).floor().clamp_(0.0, 255.0)
    d = synthetic_disparity(T, H, W)
    pos = (torch.arange(W, dtype=torch.float32)[None, None, :] + d).clamp_(0.0, W - 1.0)          # (T, H, W) sample position in the left view
    x0 = pos.floor().long().clamp_(0, W - 2)
    a = (pos - x0.float())[:, None]
    idx0 = x0[:, None].expand(T, 3, H, W)
    right = (1.0 - a) * torch.gather(left, 3, idx0) + a * torch.gather(left, 3, idx0 + 1)
    right = (right + noise * 255.0 * hash_uniform((T, 3, H, W), seed * 8 + 2)).clamp_(0.0, 255.0)
    return torch.stack([left, right], 1).contiguous()
