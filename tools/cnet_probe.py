"""Runs the HIP cnet on BASELINE config 2's left images (5 images of 320x512) a few times: target for rocprofv3 --kernel-trace --stats.
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/cnet -- /usr/bin/python3 tools/cnet_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppmstereo_amd import weights as Wm
from ppmstereo_amd.cnet import Feature
dev = "cuda:0"
m = Feature("tiny", 256)
m.load_state_dict(Wm.cnet_weights())
m = m.to(dev).eval()
T, H, W = 5, 320, 512
img = Wm.hash_uniform((T, 3, H, W), 911).to(dev)
m(img)
torch.cuda.synchronize()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
t0 = time.perf_counter()
for _ in range(reps):
    m(img)
torch.cuda.synchronize()
print(f"cnet {T} images {H}x{W}: {(time.perf_counter() - t0) / reps * 1e3:.2f} ms per call")
