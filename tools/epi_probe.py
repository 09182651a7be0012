"""Cost of the fused conv epilogues at BASELINE config 2's 1/4 scale: times a GRU conv as built by the engine and with
its epilogue reduced step by step (no hoisted share, plain SP store).  usage: tools/epi_probe.py [op,...] [reps]"""
import copy
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from ppmstereo_amd import _lib as L
from ppmstereo_amd import weights as Wm
from ppmstereo_amd.engine import ConvOp
from ppmstereo_amd.ppmstereo import PPMStereoHotPath
from ppmstereo_amd.weights import hash_normal

dev = torch.device("cuda:0")
ops = (sys.argv[1] if len(sys.argv) > 1 else "zr1_0,q1,zr2,q2,zr3,q3").split(",")
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
T, h, w = (int(x) for x in os.environ.get("PROBE_SHAPE", "5,80,128").split(","))
m = PPMStereoHotPath().load_hot_path_weights(Wm.hot_path_weights()).to(dev).eval()
eng = m.update_block04.engine(T, h, w, dev)
for t in (eng.X, eng.Hb[0], eng.Hb[1], eng.Hb[2], eng.RH, eng.ZT, eng.RT):
    t.set_f32(0.3 * hash_normal((t.pixels, t.channels), 1).to(dev))


def timeit(op):
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
    return ts[len(ts) // 2] * 1e3


def variant(op, fn):
    d = L.Conv.from_buffer_copy(bytes(op.desc))
    for i in range(2):
        fn(d.epi[i])
    return ConvOp(d, op.keep, op.version, op.wm_hint)


def no_pre(e):
    e.pre_f32 = None


def plain(e):
    e.pre_f32 = None
    e.kind, e.act = L.EPI_STORE, L.ACT_NONE
    if not e.out_sp.hi:
        e.out_sp = eng.ZT.view()
    e.out_f32 = None


for name in ops:
    op = eng.op[name]
    print(f"{name:7s} v{op.version}  engine={timeit(op):7.1f} us   no-pre={timeit(variant(op, no_pre)):7.1f} us   plain-store={timeit(variant(op, plain)):7.1f} us")
