"""Per-launch times of the fnet / cnet plans at config 2's sizes (GPU box): python tools/enc_probe.py [fnet|cnet]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppmstereo_amd import weights as Wm  # noqa: E402
from ppmstereo_amd.engine import ConvOp  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "fnet"
dev = "cuda:0"
T, H, W = 5, 320, 512
i1, i2 = Wm.hash_uniform((T, 3, H, W), 611).to(dev), Wm.hash_uniform((T, 3, H, W), 612).to(dev)
if which == "fnet":
    from ppmstereo_amd.encoder import BasicEncoder
    m = BasicEncoder(256, "instance")
    m.load_state_dict(Wm.fnet_weights())
    m = m.to(dev).eval()
    run = lambda: m([i1, i2])
    run()
    eng = list(m._engines.values())[0]
    ops = [op for _, op in eng.ops]
else:
    from ppmstereo_amd.cnet import Feature
    m = Feature("tiny", 256)
    m.load_state_dict(Wm.cnet_weights())
    m = m.to(dev).eval()
    run = lambda: m(i1)
    run()
    eng = list(m._engines.values())[0]
    ops = eng.steps
for _ in range(3):
    run()
torch.cuda.synchronize()
reps = 5
tot = [0.0] * len(ops)
for _ in range(reps):
    ev = []
    for op in ops:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        op()
        b.record()
        ev.append((a, b))
    torch.cuda.synchronize()
    for i, (a, b) in enumerate(ev):
        tot[i] += a.elapsed_time(b) / reps
conv_ms = other_ms = 0.0
for op, ms in zip(ops, tot):
    if isinstance(op, ConvOp):
        d = op.desc
        cin = sum(d.seg[i].c for i in range(d.nseg))
        gf = op.flops() / 1e9
        print(f"conv v{op.version} {d.T}x{d.H}x{d.W} cin={cin} M={d.M} k={d.kh}x{d.kw} ns={op.nslice}: {ms * 1e3:7.1f} us  {gf:6.1f} GF  {gf / ms:6.0f} TFLOP/s" if ms > 0 else "")
        conv_ms += ms
    else:
        print(f"call: {ms * 1e3:7.1f} us")
        other_ms += ms
t0 = torch.cuda.Event(enable_timing=True); t1 = torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(5):
    run()
t1.record()
torch.cuda.synchronize()
print(f"{which}: convs {conv_ms:.3f} ms, other launches {other_ms:.3f} ms (event-bracketed sum), whole call {t0.elapsed_time(t1) / 5:.3f} ms")
