"""Experiment: capture one cascade call (config 2) in a HIP graph (torch.cuda.CUDAGraph) and compare replay with eager."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppmstereo_amd import weights as Wm
from ppmstereo_amd.ppmstereo import PPMStereoHotPath
from ppmstereo_amd.synth import synth_cascade_feats

dev = torch.device("cuda:0")
T, H, W, iters = 5, 320, 512, 10
m = PPMStereoHotPath().load_hot_path_weights(Wm.hot_path_weights()).to(dev).eval()
feats = {k: v.to(dev) for k, v in synth_cascade_feats(T, H, W).items()}
for _ in range(3):
    ref, _ = m.cascade(feats, iters, T)
torch.cuda.synchronize()
ref = ref.clone()

def timeit(fn, n=10):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3

print("eager  ms/clip", round(timeit(lambda: m.cascade(feats, iters, T)), 3))
g = torch.cuda.CUDAGraph()
s = torch.cuda.Stream()
s.wait_stream(torch.cuda.current_stream())
with torch.cuda.stream(s):
    m.cascade(feats, iters, T)
torch.cuda.current_stream().wait_stream(s)
torch.cuda.synchronize()
with torch.cuda.graph(g):
    out, _ = m.cascade(feats, iters, T)
torch.cuda.synchronize()
g.replay(); torch.cuda.synchronize()
print("graph max|diff| vs eager:", (out - ref).abs().max().item())
print("graph  ms/clip", round(timeit(g.replay), 3))
