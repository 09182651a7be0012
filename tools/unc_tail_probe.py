"""Times ppms_unc_tail (128 -> 1 projection + sigmoid + per-block confidence sums) at the 1/4 scale of BASELINE config 2."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppmstereo_amd import weights as Wm, _lib as L
from ppmstereo_amd.ppmstereo import PPMStereoHotPath
from ppmstereo_amd.weights import hash_normal
dev = torch.device("cuda:0")
m = PPMStereoHotPath().load_hot_path_weights(Wm.hot_path_weights()).to(dev).eval()
eng = m.update_block04.engine(5, 80, 128, dev)
eng.U1.set_f32(0.3 * hash_normal((eng.U1.pixels, eng.U1.channels), 1).to(dev))
lib = L.load()
def run():
    L.check(lib.ppms_unc_tail(eng.U1.view(), eng.pk.unc2_w.data_ptr(), eng.pk.unc2_b, eng.UNC.data_ptr(), eng.PART.data_ptr(), eng.T, eng.n, L.stream_ptr()))
for _ in range(5): run()
torch.cuda.synchronize()
ts = []
for _ in range(20):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); run(); b.record(); torch.cuda.synchronize(); ts.append(a.elapsed_time(b) * 1e3)
ts.sort(); print("unc_tail: median %.1f us min %.1f; checksum %.6f %.6f" % (ts[10], ts[0], float(eng.UNC.sum()), float(eng.PART.sum())))
