"""Times ppms_dwconv_gelu (7x7 depthwise + GELU residual, 40 channels of the 64-channel SP tensor) at the 1/4 scale of BASELINE config 2."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppmstereo_amd import weights as Wm, _lib as L
from ppmstereo_amd.ppmstereo import PPMStereoHotPath
from ppmstereo_amd.weights import hash_normal
dev = torch.device("cuda:0")
m = PPMStereoHotPath().load_hot_path_weights(Wm.hot_path_weights()).to(dev).eval()
T_, h_, w_ = (int(v) for v in os.environ.get("PROBE_SHAPE", "5,80,128").split(","))
eng = m.update_block04.engine(T_, h_, w_, dev)
for t in (eng.C1, eng.C2):
    t.set_f32(0.3 * hash_normal((t.pixels, t.channels), 1).to(dev))
w7 = hash_normal((40, 49), 3).to(dev).contiguous(); b7 = hash_normal((40,), 4).to(dev)
lib, s = L.load(), L.stream_ptr()
def run():
    L.check(lib.ppms_dwconv_gelu(eng.C2.view(0, 40), eng.C1.view(0, 40), w7.data_ptr(), b7.data_ptr(), 7, eng.T, eng.h, eng.w, s))
for _ in range(5): run()
torch.cuda.synchronize()
ts = []
for _ in range(20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); run(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
ts.sort(); print("dwconv_gelu 7x7: median %.1f us min %.1f" % (ts[10], ts[0]))
