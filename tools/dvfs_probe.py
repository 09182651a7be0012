"""Is the large conv power-limited?  Times the GRU (1,1,15) conv of the 1/4 scale (T=5, 80x128) on its real operands and on all-zero
operands (same instruction stream; zeros draw less power, so a power-limited kernel runs faster on them: MI355X_MICROARCH.md,
DVFS give-back).  usage: python tools/dvfs_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppmstereo_amd import weights as Wm
from ppmstereo_amd.ppmstereo import PPMStereoHotPath
from ppmstereo_amd.synth import synth_cascade_feats

dev = torch.device("cuda:0")
T, H, W = 5, 320, 512
model = PPMStereoHotPath().load_hot_path_weights(Wm.hot_path_weights()).to(dev).eval()
feats = {k: v.to(dev) for k, v in synth_cascade_feats(T, H, W).items()}
model.cascade(feats, 2, T)
eng = model.update_block04.engine(T, H // 4, W // 4, dev)


def time_op(name, reps=30):
    op = eng.op[name]
    for _ in range(5):
        op()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
    for a, b in ev:
        a.record()
        op()
        b.record()
    torch.cuda.synchronize()
    ts = sorted(a.elapsed_time(b) for a, b in ev)
    return ts[len(ts) // 2], op.flops()


for name in ("zr1_0", "fh1", "q1", "zr3"):
    t, fl = time_op(name)
    print(f"{name}: v{eng.op[name].version} random {t * 1e3:.1f} us = {fl / t / 1e9:.0f} TFLOP/s", end="")
    keep = [(x, x.clone()) for x in (eng.X.data, eng.Hb[0].data, eng.Hb[2].data, eng.RH.data)]
    for x, _ in keep:
        x.zero_()
    t0, _ = time_op(name)
    for x, c in keep:
        x.copy_(c)
    print(f" | zero activations {t0 * 1e3:.1f} us ({t / t0:.2f}x)")
