import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import torch
import ab_switches  # noqa
from ppmstereo_amd import weights as Wm
from ppmstereo_amd.ppmstereo import PPMStereoHotPath
from ppmstereo_amd.synth import synth_cascade_feats
dev = "cuda:0"
model = PPMStereoHotPath().load_hot_path_weights(Wm.hot_path_weights()).to(dev).eval()
T, H, Wd = 5, 736, 1280
feats = {k: v.to(dev) for k, v in synth_cascade_feats(T, H, Wd).items()}
for iters in (2, 20):
    p, u = [], []
    d1, c1 = model.cascade(feats, iters, T, p, u)
    torch.cuda.synchronize()
    print("iters", iters, "finite:", bool(torch.isfinite(d1).all()), [bool(torch.isfinite(x).all()) for x in p])
