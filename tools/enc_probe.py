"""The producers of the path's inputs alone (fnet on 2T images, cnet on T images, SST block on the 1/16 features) at config 2's sizes, for a
kernel trace:   cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d <dir> -o enc -- /usr/bin/python3 <repo>/tools/enc_probe.py [fnet|cnet|sst]"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from ppmstereo_amd import weights as Wm                 # noqa: E402
from ppmstereo_amd.cnet import Feature                  # noqa: E402
from ppmstereo_amd.encoder import BasicEncoder          # noqa: E402
from ppmstereo_amd.sst import SSTBlock                  # noqa: E402

which = sys.argv[1] if len(sys.argv) > 1 else "all"
dev = torch.device("cuda:0")
T, H, W = 5, 320, 512
i1, i2 = Wm.hash_uniform((T, 3, H, W), 611).to(dev), Wm.hash_uniform((T, 3, H, W), 612).to(dev)
f16a, f16b = Wm.hash_normal((T, 256, H // 16, W // 16), 5).to(dev), Wm.hash_normal((T, 256, H // 16, W // 16), 6).to(dev)
runs = {}
if which in ("all", "fnet"):
    fnet = BasicEncoder(256, "instance")
    fnet.load_state_dict(Wm.fnet_weights())
    fnet = fnet.to(dev).eval()
    runs["fnet"] = lambda: fnet([i1, i2])
if which in ("all", "cnet"):
    cnet = Feature("tiny", 256)
    cnet.load_state_dict(Wm.cnet_weights())
    cnet = cnet.to(dev).eval()
    runs["cnet"] = lambda: cnet(i1)
if which in ("all", "sst"):
    sst = SSTBlock()
    sst.load_state_dict(Wm.sst_weights())
    sst = sst.to(dev).eval()
    runs["sst"] = lambda: sst(f16a, f16b, T)
for name, fn in runs.items():
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    print(f"{name}: {(time.perf_counter() - t0) / 5 * 1e3:.3f} ms per call")
