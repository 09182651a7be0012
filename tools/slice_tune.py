"""Per-conv sweep of conv_gemm2's grid-level K slices at the small scales: for every implicit-GEMM op of the 1/16- and 1/8-scale engines
(BASELINE config 2) times the launch (conv + reduce) for nslice in the legal set and prints it beside the library's own choice."""
import ctypes as C
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppmstereo_amd import _lib as L
from ppmstereo_amd import weights as Wm
from ppmstereo_amd.engine import ConvOp
from ppmstereo_amd.ppmstereo import PPMStereoHotPath
from ppmstereo_amd.synth import synth_cascade_feats

dev = torch.device("cuda:0")
T, H, W = 5, 320, 512
model = PPMStereoHotPath().load_hot_path_weights(Wm.hot_path_weights()).to(dev).eval()
feats = {k: v.to(dev) for k, v in synth_cascade_feats(T, H, W).items()}
model.cascade(feats, 2, T)
lib = L.load()


def timed(op, reps=30):
    for _ in range(3):
        op()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        op()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3

tot_now = tot_best = 0.0
for sc, blk in ((16, model.update_block16), (8, model.update_block08)):
    eng = blk.engine(T, H // sc, W // sc, dev)
    for name, op in eng.conv_ops().items():
        if op.version != 2 or op.ysweep or op.wm_hint != 0:
            continue
        d = op.desc
        nchunk = sum(d.seg[i].c for i in range(d.nseg)) // 32
        rs = d.kh * nchunk
        cur = op.nslice
        res = {}
        for s in (1, 2, 3, 4, 6, 8, 12, 16):
            if rs % s or (d.kt * rs * d.kw) // s < 2:
                continue
            try:
                o2 = ConvOp(d, op.keep, 2, nslice=s, device=dev)
                res[s] = timed(o2)
            except RuntimeError:
                pass
        if not res:
            continue
        best = min(res, key=res.get)
        tot_now += res.get(cur, float("nan"))
        tot_best += res[best]
        flag = "" if best == cur or res[best] > 0.93 * res.get(cur, 1e9) else "  <-- "
        print(f"1/{sc} {name:12s} M={d.M:3d} k=({d.kt},{d.kh},{d.kw}) rows/tap={rs:3d} cur={cur:2d}:{res.get(cur, float('nan')):6.1f}us best={best:2d}:{res[best]:6.1f}us  all={ {k: round(v, 1) for k, v in res.items()} }{flag}")
print(f"sum of the swept ops: current {tot_now:.0f} us, best-per-op {tot_best:.0f} us (per iteration of both scales)")
