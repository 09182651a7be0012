"""Every conv op of one scale's engine, timed stand-alone with HIP events: name, kernel generation, K slices, us, algorithmic TFLOP/s.
usage: tools/conv_table.py [scale=4] [reps=20]     (BASELINE config 2 geometry: T=5, 320x512)"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ppmstereo_amd import weights as Wm
from ppmstereo_amd.ppmstereo import PPMStereoHotPath
from ppmstereo_amd.weights import hash_normal

dev = torch.device("cuda:0")
sc = int(sys.argv[1]) if len(sys.argv) > 1 else 4
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
T, H, W = 5, 320, 512
m = PPMStereoHotPath().load_hot_path_weights(Wm.hot_path_weights()).to(dev).eval()
eng = {16: m.update_block16, 8: m.update_block08, 4: m.update_block04}[sc].engine(T, H // sc, W // sc, dev)
for t in (eng.X, eng.Hb[0], eng.Hb[1], eng.Hb[2], eng.RH, eng.ZT, eng.RT, eng.FH1, eng.M1, eng.COR256, eng.CF[0], eng.FLO1, eng.VAL):
    t.set_f32(0.3 * hash_normal((t.pixels, t.channels), 1).to(dev))
tot = 0.0
for name, op in eng.conv_ops().items():
    for _ in range(3):
        op()
    torch.cuda.synchronize()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        op()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    med = ts[len(ts) // 2]
    d = op.desc
    segs = [d.seg[i].c for i in range(d.nseg)]
    tot += med
    print(f"{name:12s} v{op.version} nslice={op.nslice} M={d.M:4d} cin={segs} k=({d.kt},{d.kh},{d.kw}) med={med*1e3:7.1f} us {op.flops()/1e9:7.2f} GF {op.flops()/med/1e9:7.1f} TF")
print(f"sum of medians {tot*1e3:.1f} us")
