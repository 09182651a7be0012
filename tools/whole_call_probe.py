"""Where PPMStereo.forward_batch_test(host video) spends its wall time on this box: host enqueue vs GPU time of the cascade, and the
whole call with / without a synchronisation behind every phase (H2D + pad, encoders, SST, cascade, D2H)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppmstereo_amd import weights as Wm
from ppmstereo_amd.ppmstereo import PPMStereo
dev = torch.device("cuda:0")
T, H, W, iters = 5, 320, 512, 10
m = PPMStereo.shipped()
m.load_hot_path_weights(Wm.hot_path_weights())
m.fnet.load_state_dict(Wm.fnet_weights(), strict=True), m.cnet.load_state_dict(Wm.cnet_weights(), strict=True)
sd = m.state_dict()
sd.update(Wm.sst_weights())
m.load_state_dict(sd, strict=True)
m = m.to(dev).eval()
video = Wm.hash_uniform((T, 2, 3, H, W), 613, 0.0, 255.0).round().contiguous()
call = lambda: m.forward_batch_test({"stereo_video": video}, kernel_size=20, iters=iters)
for _ in range(2):
    call()
torch.cuda.synchronize()
for _ in range(5):
    t0 = time.perf_counter()
    out = call()
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"whole call: returned after {1e3 * (t1 - t0):.1f} ms, device idle after {1e3 * (t2 - t0):.1f} ms")
# phases, synchronised
vd = None
for _ in range(3):
    t0 = time.perf_counter()
    vd = video.to(dev)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(f"H2D of the {video.numel() * 4 / 1e6:.1f} MB video (pageable host tensor): {1e3 * (t1 - t0):.2f} ms")
pin = video.pin_memory()
for _ in range(3):
    t0 = time.perf_counter()
    vd = pin.to(dev, non_blocking=True)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    print(f"H2D from pinned memory: {1e3 * (t1 - t0):.2f} ms")
from ppmstereo_amd.synth import synth_cascade_feats
feats = {k: v.to(dev) for k, v in synth_cascade_feats(T, H, W).items()}
for _ in range(3):
    t0 = time.perf_counter()
    m.cascade(feats, iters, T)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"cascade: enqueue {1e3 * (t1 - t0):.1f} ms, total {1e3 * (t2 - t0):.1f} ms")
i1, i2 = vd[:, 0].contiguous(), vd[:, 1].contiguous()
for name, fn in (("fnet", lambda: m.fnet([i1, i2])), ("cnet", lambda: m.cnet(i1))):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter(); fn(); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"{name}: enqueue {1e3 * (t1 - t0):.1f} ms, total {1e3 * (t2 - t0):.1f} ms")
print("host: cpus", os.cpu_count(), "loadavg", os.getloadavg())

# the phases of one call, synchronised behind each (which one carries a slow repetition on a busy host?)
from ppmstereo_amd.ppmstereo import InputPadder
print("phases per call (ms): H2D | pad + forward (encoders + SST + cascade) | unpad + D2H")
for _ in range(8):
    t0 = time.perf_counter()
    win = video.to(dev)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    left, right = win[:, 0], win[:, 1]
    padder = InputPadder(left.shape, divis_by=32)
    left, right = padder.pad(left, right)
    d, u = m.forward(left[None], right[None], iters=iters, test_mode=True)
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    dh, uh = padder.unpad(d[0])[:, None].cpu(), padder.unpad(u[0])[:, None].cpu()
    t3 = time.perf_counter()
    print(f"   {1e3 * (t1 - t0):7.2f} | {1e3 * (t2 - t1):7.2f} | {1e3 * (t3 - t2):7.2f}")
