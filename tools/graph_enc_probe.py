"""Does HIP-graph replay help the encoders (hundreds of 5-40 us launches issued from Python)?  Times cnet / fnet / SST eager vs captured."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppmstereo_amd import weights as Wm
from ppmstereo_amd.cnet import Feature
from ppmstereo_amd.encoder import BasicEncoder
from ppmstereo_amd.sst import SSTBlock
dev = "cuda:0"
T, H, W = 5, 320, 512
cnet = Feature("tiny", 256); cnet.load_state_dict(Wm.cnet_weights()); cnet = cnet.to(dev).eval()
fnet = BasicEncoder(256, "instance"); fnet.load_state_dict(Wm.fnet_weights()); fnet = fnet.to(dev).eval()
sst = SSTBlock(); sst.load_state_dict(Wm.sst_weights()); sst = sst.to(dev).eval()
i1, i2 = Wm.hash_uniform((T, 3, H, W), 611).to(dev), Wm.hash_uniform((T, 3, H, W), 612).to(dev)
a, b = Wm.hash_normal((T, 256, 20, 32), 831).to(dev), Wm.hash_normal((T, 256, 20, 32), 832).to(dev)

def timed(fn, reps=10):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3

for name, fn in (("cnet", lambda: cnet(i1)), ("fnet", lambda: fnet([i1, i2])), ("sst", lambda: sst(a, b, T))):
    eager = timed(fn)
    ref = fn()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        out = fn()
    rep = timed(lambda: g.replay())
    g.replay(); torch.cuda.synchronize()
    same = all(torch.equal(x, y) for x, y in zip(out, ref))
    print(f"{name}: eager {eager:.2f} ms, graph replay {rep:.2f} ms, identical results: {same}")
