"""Runs the HIP fnet on BASELINE config 2's images (10 images of 320x512) a few times: target for rocprofv3 --kernel-trace --stats.
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/fnet -- /usr/bin/python3 tools/fnet_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppmstereo_amd import weights as Wm
from ppmstereo_amd.encoder import BasicEncoder
dev = "cuda:0"
m = BasicEncoder(256, "instance")
m.load_state_dict(Wm.fnet_weights())
m = m.to(dev).eval()
T, H, W = 5, 320, 512
i1, i2 = Wm.hash_uniform((T, 3, H, W), 611).to(dev), Wm.hash_uniform((T, 3, H, W), 612).to(dev)
m([i1, i2])
torch.cuda.synchronize()
t0 = time.perf_counter()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
for _ in range(reps):
    m([i1, i2])
torch.cuda.synchronize()
print(f"fnet {T}+{T} images {H}x{W}: {(time.perf_counter() - t0) / reps * 1e3:.2f} ms per call")
