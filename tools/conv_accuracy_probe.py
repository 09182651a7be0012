"""How far are the bf16x3 convolutions from the IDEAL bf16x3 result?  (GPU box)

For GRU-sized convolutions the probe compares, against an fp64 reference of the fp32 convolution:
  ideal   : hi*hi + hi*lo + lo*hi evaluated exactly (fp64 sums of the bf16 products) -- what the split scheme can give at best;
  hip vN  : the library's kernels (conv_gemm2 / conv_gemm5).
Reported per case: rms error / rms output, and the BIAS of the error along the sign of the output (mean(err * sign(ref)) / rms):
a truncating (round-toward-zero) accumulation inside the matrix pipe shows up as a negative bias, round-to-nearest as ~0.

    python tools/conv_accuracy_probe.py
"""
import os
import sys

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from ppmstereo_amd import _lib as L            # noqa: E402
from ppmstereo_amd.weights import hash_normal  # noqa: E402
from test_gpu_ops import _run_conv             # noqa: E402


def split(x):
    hi = x.to(torch.bfloat16).float()
    lo = (x - hi).to(torch.bfloat16).float()
    return hi.double(), lo.double()


def conv64(xs, w, b, k3, T, H, W):
    x = torch.cat(xs, 1)
    x5 = x.reshape(1, T, H, W, -1).permute(0, 4, 1, 2, 3)
    y = F.conv3d(x5, w, b, padding=tuple(k // 2 for k in k3))
    return y.permute(0, 2, 3, 4, 1).reshape(T * H * W, -1)


def main():
    torch.set_num_threads(16)
    cases = [("zr1_0 (1,1,15) 512->256", 2, 8, 64, [128, 384], 256, (1, 1, 15), (2, 5)),
             ("q1 (1,1,5) 512->128", 2, 8, 64, [128, 384], 128, (1, 1, 5), (2, 5)),
             ("3x3 128->256", 2, 16, 64, [128], 256, (1, 3, 3), (2, 5)),
             ("1x1 384->384", 5, 4, 16, [384], 384, (1, 1, 1), (2,)),
             ("(5,1,1) 512->256", 5, 8, 32, [128, 384], 256, (5, 1, 1), (2,))]
    for name, T, H, W, segs, cout, k3, versions in cases:
        P = T * H * W
        xs = [torch.relu(hash_normal((P, c), 100 + i)) if i else torch.tanh(hash_normal((P, c), 100 + i)) for i, c in enumerate(segs)]
        cin = sum(segs)
        wt = (hash_normal((cout, cin, *k3), 200) / (cin * k3[0] * k3[1] * k3[2]) ** 0.5)
        bs = hash_normal((cout,), 201) * 0.1
        ref = conv64([x.double() for x in xs], wt.double(), bs.double(), k3, T, H, W)
        rms = ref.pow(2).mean().sqrt().item()
        parts = [split(x) for x in xs]
        wh, wl = split(wt)
        xh, xl = [p[0] for p in parts], [p[1] for p in parts]
        ideal = conv64(xh, wh, bs.double(), k3, T, H, W) + conv64(xh, wl, None, k3, T, H, W) + conv64(xl, wh, None, k3, T, H, W)
        f32 = conv64(xs, wt, bs, k3, T, H, W).double()

        def stat(tag, y):
            e = y.double() - ref
            print(f"  {tag:14s} rms err / rms out = {e.pow(2).mean().sqrt().item() / rms:.3e}   max / rms = {e.abs().max().item() / rms:.3e}   "
                  f"bias along sign(out) / rms = {(e * torch.sign(ref)).mean().item() / rms:+.3e}")

        print(f"{name}: T={T} {H}x{W}, rms out {rms:.3f}")
        stat("torch fp32 CPU", f32)
        stat("ideal bf16x3", ideal)
        for v in versions:
            seg_pad = [((c + 15) // 16) * 16 for c in segs] if v == 5 else None
            try:
                got = _run_conv(L, xs, wt, bs, k3, T, H, W, version=v, seg_pad=seg_pad)
                stat(f"hip conv{v}", got)
            except Exception as ex:      # noqa: BLE001
                print(f"  hip conv{v}: not applicable here ({str(ex)[:80]})")


if __name__ == "__main__":
    main()
