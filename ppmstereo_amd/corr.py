"""CorrBlock1D with the reference's call signature (/root/reference/models/core/corr.py:47-104), backed by the
gfx950 kernels of ppmstereo_amd/csrc/corr.hip.  Drop-in for ``from models.core.corr import CorrBlock1D``."""
from __future__ import annotations

import ctypes as C
from typing import List

import torch

from . import _lib as L


BUILD_EVENTS = None        # a list while bench.py times the pyramid builds: ((B, H, W), (start, stop)) per launch


def coords_grid(batch: int, ht: int, wd: int, device) -> torch.Tensor:
    """corr.py:47-52 -- (B,2,H,W), channel 0 = x, channel 1 = y."""
    ys, xs = torch.meshgrid(torch.arange(ht, device=device), torch.arange(wd, device=device), indexing="ij")
    return torch.stack([xs, ys], dim=0).float()[None].repeat(batch, 1, 1, 1)


def build_pyramid(fmap1: torch.Tensor, fmap2: torch.Tensor) -> List[torch.Tensor]:
    """Five levels (B*H*W, W>>l) fp32, one allocation; ppms_corr_build (corr.py:56-72,96-104)."""
    L.require_gpu(fmap1, fmap2)
    if fmap1.shape != fmap2.shape or fmap1.dim() != 4:
        raise RuntimeError(f"CorrBlock1D: fmap shapes {tuple(fmap1.shape)} / {tuple(fmap2.shape)}")
    f1, f2 = fmap1.contiguous().float(), fmap2.contiguous().float()
    B, Cc, H, W = f1.shape
    rows = B * H * W
    widths = [W >> l for l in range(5)]
    if widths[4] < 1:
        # same failure point as the reference (avg_pool2d of an empty row, corr.py:71; SURVEY.md hazard 2)
        raise RuntimeError(f"CorrBlock1D: width {W} is too small for a 4-level pyramid (needs >= 16 at this scale)")
    store = torch.empty(rows * sum(widths), dtype=torch.float32, device=f1.device)
    levels, off = [], 0
    for wl in widths:
        levels.append(store[off:off + rows * wl].view(rows, wl))
        off += rows * wl
    ptrs = (C.c_void_p * 5)(*[t.data_ptr() for t in levels])
    ev = BUILD_EVENTS
    if ev is not None:                      # bench.py: HIP events on the launch stream around the pyramid build (roofline_hbm)
        pair = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
        pair[0].record()
    L.check(L.load().ppms_corr_build(f1.data_ptr(), f2.data_ptr(), ptrs, B, Cc, H, W, L.stream_ptr()))
    if ev is not None:
        pair[1].record()
        ev.append(((B, H, W), pair))
    return levels


class CorrBlock1D:
    def __init__(self, fmap1: torch.Tensor, fmap2: torch.Tensor, num_levels: int = 4, radius: int = 4):
        if num_levels != 4 or radius != 4:
            raise NotImplementedError("CorrBlock1D: the gfx950 kernels implement num_levels=4, radius=4 (the only configuration PPMStereo uses)")
        self.num_levels, self.radius = num_levels, radius
        B, _, H, W = fmap1.shape
        self.shape = (B, H, W)
        self._device, self._coords = fmap1.device, None      # .coords is built on first access: the kernels take x from the pixel index,
        self.levels = build_pyramid(fmap1, fmap2)             # and a grid per CorrBlock1D is 5 small launches per scale of every clip
        # reference attribute: list of (B*H*W1, 1, 1, W2_l), num_levels + 1 entries (the last is never read)
        self.corr_pyramid = [lv.view(lv.shape[0], 1, 1, lv.shape[1]) for lv in self.levels]

    @property
    def coords(self) -> torch.Tensor:
        """reference attribute (corr.py:62): (B, 2, H, W) pixel coordinates"""
        if self._coords is None:
            self._coords = coords_grid(*self.shape, self._device)
        return self._coords

    def __call__(self, flow: torch.Tensor) -> torch.Tensor:
        B, H, W = self.shape
        L.require_gpu(flow)
        if tuple(flow.shape) != (B, 2, H, W):
            raise RuntimeError(f"CorrBlock1D: flow shape {tuple(flow.shape)} != {(B, 2, H, W)}")
        f = flow.contiguous().float()
        out = torch.empty(B, 36, H, W, dtype=torch.float32, device=f.device)
        ptrs = (C.c_void_p * 4)(*[t.data_ptr() for t in self.levels[:4]])
        L.check(L.load().ppms_corr_lookup(ptrs, f.data_ptr(), 0, out.data_ptr(), None, None, 0, None, None, 0, B, H, W, L.stream_ptr()))
        return out

    @staticmethod
    def corr(fmap1: torch.Tensor, fmap2: torch.Tensor) -> torch.Tensor:
        """corr.py:96-104 -- (B, H, W1, 1, W2)."""
        B, _, H, W = fmap1.shape
        return build_pyramid(fmap1, fmap2)[0].view(B, H, W, 1, W)
