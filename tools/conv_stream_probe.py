"""Per-launch time of the update block's convolutions on the small maps (1/16 and 1/8 scales of config 2; 1/16 of config 3): the LDS-staged
implicit GEMM as the library plans it (conv_gemm2: K slices + slice-reduce launch, y sweep where it applies) against the register-streamed
kernel (conv_stream.hip) with 32- and 64-pixel tiles.  Back-to-back launches of one conv (weights L2-warm), HIP events around the batch.
GPU box:  python tools/conv_stream_probe.py [scale ...]      (scales: 16 8 c3_16; default all)
Sweep of the ring depth / waves per workgroup (a build with the extra instantiations):
          PPMS_BUILD_DEFINES=-DPPMS_STREAM_PROBE python tools/conv_stream_probe.py --sweep 16 8"""
import ctypes as C
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ppmstereo_amd import _lib as L                                 # noqa: E402
from ppmstereo_amd.engine import ConvOp, epilogue                   # noqa: E402
from ppmstereo_amd.packing import pack_conv2, pack_stream           # noqa: E402
from ppmstereo_amd.weights import hash_normal                      # noqa: E402

DEV = torch.device("cuda:0")

# name, segs (padded channels), couts, taps -- the per-iteration convolutions of SequenceUpdateBlock3D (hoisted form: x = [mf, mfg])
CONVS = [
    ("convf2", [128], 64, (1, 3, 3)), ("convc2", [256], 192, (1, 3, 3)), ("final", [320], 190, (1, 3, 3)), ("unc0", [128, 128], 128, (1, 3, 3)),
    ("zr1_0", [128, 256], 256, (1, 1, 15)), ("z1_2", [128], 128, (1, 1, 5)), ("q1", [128, 256], 128, (1, 1, 5)),
    ("zr2", [128, 256], 256, (1, 5, 1)), ("q2", [128, 256], 128, (1, 5, 1)), ("zr3", [128, 256], 256, (5, 1, 1)), ("q3", [128, 256], 128, (5, 1, 1)),
    ("fh1", [128], 256, (3, 3, 3)), ("m1", [128], 256, (1, 3, 3)),
]
# update_block16 does not hoist (its attention rewrites x every iteration): x = [inp, mf, mfg], plus the K = 768 Linear layers
CONVS16 = [(n, ([128, 384] if s == [128, 256] else s), m, k) for n, s, m, k in CONVS] + [("sa_mlp0", [384, 384], 768, (1, 1, 1)), ("sa_mlp2", [768], 384, (1, 1, 1))]
SCALES = {"16": (5, 20, 32, CONVS16), "8": (5, 40, 64, CONVS), "c3_16": (5, 46, 80, CONVS16)}


def bench(op, n=40):
    for _ in range(5):
        op()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        op()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


def hint_of(kg, pb, d):
    return (kg << 8) | (pb << 4) | d


SWEEP = [(4, 1, 6), (4, 1, 8), (4, 1, 10), (8, 1, 4), (8, 1, 6), (8, 1, 8), (4, 2, 3), (4, 2, 4), (4, 2, 6), (8, 2, 3), (8, 2, 4), (8, 2, 6)]


def main():
    lib = L.load()
    args = [a for a in sys.argv[1:] if a != "--sweep"]
    sweep = "--sweep" in sys.argv[1:]
    variants = [(7, hint_of(*v)) for v in SWEEP] if sweep else [(7, 1), (7, 2)]
    for key in (args or list(SCALES)):
        T, H, W, convs = SCALES[key]
        P = T * H * W
        tot = {"conv2": 0.0, "s1": 0.0, "s2": 0.0, "best": 0.0}
        print(f"--- scale {key}: T={T} {H}x{W} = {P} pixels")
        for name, segs, cout, k3 in convs:
            xs = []
            for i, c in enumerate(segs):
                t = L.SPTensor(P, c, DEV)
                t.set_f32(hash_normal((P, c), 10 + i).to(DEV))
                xs.append(t)
            cin = sum(segs)
            w = (hash_normal((cout, cin, *k3), 2) / math.sqrt(cin * k3[0] * k3[1] * k3[2])).to(DEV)
            bias = hash_normal((cout,), 3).to(DEV)
            M = (cout + 63) // 64 * 64
            out = L.SPTensor(P, M, DEV)
            gflop = 2.0 * P * cout * cin * k3[0] * k3[1] * k3[2] * 1e-9
            res, times = [], {}
            for ver, hint in [(2, 0)] + variants:
                wp = w
                ys = ver == 2 and k3[1] > 1 and k3[2] == 1
                if ys:
                    wp = w.transpose(3, 4).contiguous()
                packed, b, meta = (pack_conv2 if ver == 2 else pack_stream)(wp, bias, segs, segs, None, M)
                d = L.Conv()
                for i, t in enumerate(xs):
                    d.seg[i] = t.view()
                d.nseg, d.w, d.bias = len(xs), packed.data_ptr(), b.data_ptr()
                d.T, d.H, d.W = T, H, W
                d.kt, d.kh, d.kw = k3
                d.M = d.m_split = M
                d.epi[0] = epilogue(act=L.ACT_RELU, n_valid=cout, out_sp=out.view())
                if ver == 2:
                    ys = ys and lib.ppms_conv_gemm2_ysweep_slices(C.byref(d)) > 0
                    if not ys and wp is not w:
                        packed, b, meta = pack_conv2(w, bias, segs, segs, None, M)
                        d.w, d.bias = packed.data_ptr(), b.data_ptr()
                    op = ConvOp(d, [packed, b], 2, ysweep=ys)
                    t_us = bench(op)
                    times["conv2"] = t_us
                    res.append(f"conv2{'y' if ys else ' '} x{op.nslice}: {t_us:6.1f}")
                else:
                    if hint > 2 and (cin // 16) % (hint >> 8):
                        continue
                    op = ConvOp(d, [packed, b], 7, hint)
                    t_us = bench(op)
                    times[f"s{hint}"] = t_us
                    tag = f"pb={hint}" if hint <= 2 else "kg%d pb%d d%d" % (hint >> 8, (hint >> 4) & 15, hint & 15)
                    res.append(f"{tag}: {t_us:6.1f}" + (f" ({gflop / t_us * 1e3:4.0f} TF/s)" if not sweep else ""))
            best = min(v for k, v in times.items() if k != "conv2")
            for k in ("conv2", "s1", "s2"):
                tot[k] += times.get(k, 0.0)
            tot["best"] += best
            print(f"{name:8s} K={cin * k3[0] * k3[1] * k3[2]:5d} M={M:4d} {gflop:6.2f} GF  " + "   ".join(res))
        print(f"sum: conv2 (+ reduce) {tot['conv2']:.0f} us   stream pb=1 {tot['s1']:.0f}   pb=2 {tot['s2']:.0f}   best stream variant per conv {tot['best']:.0f}")


if __name__ == "__main__":
    main()
