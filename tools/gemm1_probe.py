"""Per-launch time of the 1x1 convolutions at config 2's sizes: implicit GEMM (conv_gemm2, as planned by the library, + its slice reduce)
against the thin-GEMM kernel with 1, 2 or 4 cout blocks per workgroup.  GPU box:  python tools/gemm1_probe.py"""
import ctypes as C
import math
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ppmstereo_amd import _lib as L                                 # noqa: E402
from ppmstereo_amd.engine import ConvOp, epilogue                   # noqa: E402
from ppmstereo_amd.packing import pack_conv2, pack_gemm1            # noqa: E402
from ppmstereo_amd.weights import hash_normal                      # noqa: E402

DEV = torch.device("cuda:0")


def bench(op, n=60):
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


def main():
    for name, T, H, W, K, M, nv, vt in (("to_v 1/4", 5, 80, 128, 128, 128, 128, True), ("convf1 1/4", 5, 80, 128, 128, 128, 128, False),
                                         ("fh2 1/4", 5, 80, 128, 256, 64, 54, False), ("m2 1/4", 5, 80, 128, 256, 192, 144, False),
                                         ("to_v 1/8", 5, 40, 64, 128, 128, 128, True), ("to_v 1/16", 5, 20, 32, 128, 128, 128, True),
                                         ("ta_proj 1/16", 5, 20, 32, 384, 384, 384, False), ("sa_qkv 1/16", 5, 20, 32, 384, 1152, 1152, False),
                                         ("cnet pw 96->384 1/4", 5, 80, 128, 96, 384, 384, False)):
        P = T * H * W
        x = L.SPTensor(P, K, DEV)
        x.set_f32(hash_normal((P, K), 1).to(DEV))
        w = (hash_normal((nv, K, 1, 1), 2) / math.sqrt(K)).to(DEV)
        out = L.SPTensor(P, M, DEV)
        vtt = torch.zeros(T, nv, H * W, dtype=torch.bfloat16, device=DEV) if vt else None
        res = []
        for ver, hint in ((2, 0), (6, 1), (6, 2), (6, 4)):
            if ver == 6 and K % 64:
                continue
            packed, b, meta = (pack_conv2 if ver == 2 else pack_gemm1)(w, None, [K], None, None, M)
            d = L.Conv()
            d.seg[0] = x.view()
            d.nseg, d.w, d.bias = 1, packed.data_ptr(), b.data_ptr()
            d.T, d.H, d.W, d.kt, d.kh, d.kw = T, H, W, 1, 1, 1
            d.M = d.m_split = M
            d.epi[0] = epilogue(n_valid=nv, out_sp=out.view(), out_vt=vtt)
            if ver == 6 and (M // 32) % hint:
                continue
            try:
                op = ConvOp(d, [packed, b], ver, hint)
                res.append(f"{'conv2' if ver == 2 else 'gemm1 cb=%d' % hint}: {bench(op):6.1f} us" + (f" (nslice {op.nslice})" if ver == 2 else ""))
            except Exception as ex:      # noqa: BLE001
                res.append(f"v{ver}/{hint}: {str(ex)[:40]}")
        print(f"{name:22s} P={P:6d} K={K:3d} M={M:4d}  " + "   ".join(res))


if __name__ == "__main__":
    main()
