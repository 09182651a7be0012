"""How long does the host need to ENQUEUE one clip (no sync) vs the GPU time?  If enqueue >= GPU time the loop is
launch-bound and hipGraph capture would pay."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppmstereo_amd import weights as Wm
from ppmstereo_amd.ppmstereo import PPMStereoHotPath
from ppmstereo_amd.synth import synth_cascade_feats
dev = torch.device("cuda:0")
m = PPMStereoHotPath().load_hot_path_weights(Wm.hot_path_weights()).to(dev).eval()
feats = {k: v.to(dev) for k, v in synth_cascade_feats(5, 320, 512).items()}
for _ in range(2):
    m.cascade(feats, 10, 5)
torch.cuda.synchronize()
for _ in range(3):
    t0 = time.perf_counter()
    m.cascade(feats, 10, 5)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"enqueue {1e3*(t1-t0):.1f} ms, total {1e3*(t2-t0):.1f} ms")
