"""Concurrency stress probe: runs small kernels of the library on the current stream while a conv kernel (or a torch GEMM) runs
on another stream, and counts runs whose result differs from the kernel running alone.  This is how the packed-fp32 problem
was found (ppmstereo_amd/build.py): with v_pk_*_f32 enabled, ppms_bilinear lost one of its four taps in lanes 48-63 of some
waves whenever a conv kernel ran concurrently; with packed fp32 formation disabled every line prints 0 mismatches."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F
from ppmstereo_amd import weights as Wm, engine as E, _lib as L
from ppmstereo_amd.ppmstereo import PPMStereoHotPath
from ppmstereo_amd.weights import hash_normal
dev = torch.device("cuda:0")
T, h, w = 5, 80, 128
m = PPMStereoHotPath().load_hot_path_weights(Wm.hot_path_weights()).to(dev).eval()
eng = m.update_block04.engine(T, h, w, dev)
for t in (eng.X, eng.Hb[0], eng.M1, eng.FH1):
    t.set_f32(0.3 * hash_normal((t.pixels, t.channels), 1).to(dev))
side = torch.cuda.Stream()
A = torch.randn(8192, 8192, device=dev); B = torch.randn(8192, 8192, device=dev)
def heavy_conv(): eng.op["m1"]()
def heavy_mm(): torch.mm(A, B)
Ab, Bb = A.bfloat16(), B.bfloat16()
def heavy_mm_bf16(): torch.mm(Ab, Bb)
def heavy_attn():
    eng_a = m.update_block04.engine(T, h, w, dev)
    eng_a.attend()
def run(heavy, main_fn, n=30):
    ref = main_fn().clone(); torch.cuda.synchronize()
    bad, worst = 0, None
    for _ in range(n):
        ev = torch.cuda.Event(); ev.record(); side.wait_event(ev)
        with torch.cuda.stream(side):
            if heavy: heavy()
        out = main_fn()
        torch.cuda.synchronize()
        if not torch.equal(out, ref):
            bad += 1
            if worst is None:
                d = (out != ref).nonzero()
                worst = (len(d), d[0].tolist(), out[out != ref][:4].tolist(), ref[out != ref][:4].tolist())
    return bad, worst
ones = torch.ones(T, 1, h, w, device=dev)
rnd = torch.sigmoid(hash_normal((T, 1, h, w), 7)).to(dev)
flow = hash_normal((eng.P, 2), 8).to(dev); mask = hash_normal((eng.P, 144), 9).to(dev)
def cvx():
    out = torch.empty(T, 2, 4 * h, 4 * w, device=dev)
    L.check(L.load().ppms_convex_upsample(flow.data_ptr(), mask.data_ptr(), 144, out.data_ptr(), T, h, w, L.stream_ptr()))
    return out
def cvt():
    out = torch.empty(T, 144, h, w, device=dev)
    L.check(L.load().ppms_nhwc_to_nchw(mask.data_ptr(), 144, out.data_ptr(), T, 144, h * w, L.stream_ptr()))
    return out
tests = {"bilinear(ones,x4)": lambda: E.bilinear(ones, (4 * h, 4 * w), False), "bilinear(rnd,x4)": lambda: E.bilinear(rnd, (4 * h, 4 * w), False),
         "bilinear(rnd,x4,align)": lambda: E.bilinear(rnd, (4 * h, 4 * w), True), "convex_upsample": cvx, "nhwc_to_nchw": cvt,
         "torch.interpolate": lambda: F.interpolate(rnd, size=(4 * h, 4 * w), mode="bilinear", align_corners=False)}
for name, fn in tests.items():
    print(f"{name:24s} alone {run(None, fn)[0]:2d}   with conv {run(heavy_conv, fn)}   with torch.mm fp32 {run(heavy_mm, fn)[0]}   with torch.mm bf16 {run(heavy_mm_bf16, fn)[0]}")
