"""Times ppms_corr_build at the three scales of BASELINE config 2 (T=5, 320x512) and checks the result against torch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppmstereo_amd.corr import CorrBlock1D
from ppmstereo_amd.weights import hash_normal
dev = torch.device("cuda:0")
for sc in (16, 8, 4):
    T, C, h, w = 5, 256, 320 // sc, 512 // sc
    f1, f2 = hash_normal((T, C, h, w), 11).to(dev), hash_normal((T, C, h, w), 12).to(dev)
    for _ in range(3): lv = CorrBlock1D(f1, f2).levels
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); lv = CorrBlock1D(f1, f2).levels; b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
    ts.sort()
    ref = torch.einsum("tchx,tchy->thxy", f1, f2) / C ** 0.5
    err = float((lv[0].view(T, h, w, w) - ref).abs().max())
    mb = (2 * C * T * h * w * 4 + 1.9375 * T * h * w * w * 4) / 1e6
    print(f"1/{sc}: corr_build median {ts[10]:.1f} us (min {ts[0]:.1f}); {mb:.0f} MB -> {mb / ts[10] * 1e-3 * 1e3:.2f} GB/ms = {mb / ts[10]:.2f} TB/s incl. launch; max |err| vs einsum {err:.2e}")
