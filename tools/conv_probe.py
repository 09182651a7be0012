"""Times individual engine ops at BASELINE config 2's 1/4 scale (T=5, 80x128) with HIP events.
usage: tools/conv_probe.py op1,op2,... [reps]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ab_switches  # noqa: F401,E402  (PPMS_CONV6=0 etc.: A/B of the kernel generations)
from ppmstereo_amd import weights as Wm
from ppmstereo_amd.ppmstereo import PPMStereoHotPath
from ppmstereo_amd.weights import hash_normal

dev = torch.device("cuda:0")
ops = sys.argv[1].split(",")
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 20
T, h, w = (int(x) for x in os.environ.get("PROBE_SHAPE", "5,80,128").split(","))
m = PPMStereoHotPath().load_hot_path_weights(Wm.hot_path_weights()).to(dev).eval()
eng = m.update_block04.engine(T, h, w, dev)
for t in (eng.X, eng.Hb[0], eng.Hb[1], eng.Hb[2], eng.RH, eng.ZT, eng.RT, eng.FH1, eng.M1, eng.COR256, eng.CF[0], eng.FLO1, eng.VAL):
    t.set_f32(0.3 * hash_normal((t.pixels, t.channels), 1).to(dev))
flops = {"zr1_0": 2 * 256 * 512 * 15, "q1": 2 * 128 * 512 * 5, "zr2": 2 * 256 * 512 * 5, "q2": 2 * 128 * 512 * 5, "zr3": 2 * 256 * 512 * 5, "q3": 2 * 128 * 512 * 5,
         "fh1": 2 * 256 * 128 * 27, "fh2": 2 * 2 * 256 * 27, "m1": 2 * 256 * 128 * 9, "m2": 2 * 144 * 256, "unc0": 2 * 128 * 256 * 9, "final_0": 2 * 190 * 320 * 9,
         "convc2_0": 2 * 192 * 256 * 9, "z1_2": 2 * 128 * 128 * 5, "r1_2": 2 * 128 * 128 * 5, "to_v": 2 * 128 * 128, "convf2_0": 2 * 64 * 128 * 9}
lz = int(os.environ.get("PROBE_LOZERO", "0"))       # e.g. 256: the mfg third of x = [mf, mfg] holds bf16-exact values (ppms_conv.lo_zero_from)
if lz:
    eng.X.own()[1, :, 256:] = 0
for name in ops:
    op = eng.op[name]
    if lz and name in ("zr1_0", "q1", "zr2", "q2", "zr3", "q3"):
        op.desc.lo_zero_from = lz
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
    fl = flops.get(name, 0) * eng.P
    print(f"{name:10s} v{getattr(op, 'version', '-')} dbg={os.environ.get('PPMS_DBG','0'):>3s} wm={os.environ.get('PPMS_WM','-')} med={med*1e3:8.1f} us  min={ts[0]*1e3:8.1f} us  {fl/med/1e9:7.1f} TFLOP/s(alg)")
