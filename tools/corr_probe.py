"""Pyramid-build probe (GPU box): python tools/corr_probe.py [T H W]  -- avg launch time of ppms_corr_build, the line-resident kernel and
(through a 4-byte misaligned copy of the features, which the fast path refuses) the general kernel, with algorithmic GB/s."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppmstereo_amd import _lib as L  # noqa: E402
from ppmstereo_amd.weights import hash_normal  # noqa: E402

T, H, W = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (5, 80, 128)
dev = "cuda:0"
lib = L.load()
f1, f2 = hash_normal((T, 256, H, W), 1).to(dev), hash_normal((T, 256, H, W), 2).to(dev)
rows = T * H * W
widths = [W >> l for l in range(5)]
store = torch.empty(rows * sum(widths), device=dev)
lv, off = [], 0
for wl in widths:
    lv.append(store[off:off + rows * wl])
    off += rows * wl
ptrs = (C.c_void_p * 5)(*[t.data_ptr() for t in lv])
by = 2 * 256 * rows * 4 + 1.9375 * rows * W * 4


def run(a, b, reps=30):
    for _ in range(3):
        L.check(lib.ppms_corr_build(a.data_ptr(), b.data_ptr(), ptrs, T, 256, H, W, L.stream_ptr()))
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for s, e in ev:
        s.record()
        L.check(lib.ppms_corr_build(a.data_ptr(), b.data_ptr(), ptrs, T, 256, H, W, L.stream_ptr()))
        e.record()
    torch.cuda.synchronize()
    ms = sorted(s.elapsed_time(e) for s, e in ev)
    return ms[len(ms) // 2]


fast = run(f1, f2)
ref = store.clone()
pad1, pad2 = torch.empty(f1.numel() + 1, device=dev), torch.empty(f2.numel() + 1, device=dev)
m1, m2 = pad1[1:].view_as(f1), pad2[1:].view_as(f2)
m1.copy_(f1), m2.copy_(f2)
slow = run(m1, m2)
same = bool(torch.equal(ref, store))
if not same:
    off = 0
    for l, wl in enumerate(widths):
        a, b_ = ref[off:off + rows * wl].view(rows, wl), store[off:off + rows * wl].view(rows, wl)
        bad = (a != b_)
        print(f"  level {l}: {int(bad.sum())} of {a.numel()} differ, max |d| {(a - b_).abs().max().item():.3e}; first bad (row, col): "
              f"{bad.nonzero()[:6].tolist()}  fast={a[bad][:4].tolist()} general={b_[bad][:4].tolist()}")
        off += rows * wl
print(f"T={T} {H}x{W}: line-resident {fast * 1e3:.1f} us = {by / fast / 1e6:.0f} GB/s; general kernel {slow * 1e3:.1f} us = {by / slow / 1e6:.0f} GB/s; "
      f"bit-identical pyramids: {same}")
