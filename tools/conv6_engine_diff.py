"""debug: every conv6 op of the 1/4-scale engine at a given geometry against the same op on conv5 / conv2 (same inputs), all engine buffers compared."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppmstereo_amd import weights as Wm, _lib as L, engine as E
from ppmstereo_amd.ppmstereo import PPMStereoHotPath
from ppmstereo_amd.weights import hash_normal
dev = torch.device("cuda:0")
T, h, w = (int(x) for x in os.environ.get("PROBE_SHAPE", "5,184,320").split(","))
def build(conv6):
    E.TUNING["conv6"] = conv6
    m = PPMStereoHotPath().load_hot_path_weights(Wm.hot_path_weights()).to(dev).eval()
    return m, m.update_block04.engine(T, h, w, dev)
mA, A = build(True)
mB, B = build(False)
def tensors(e):
    out = {}
    for k, v in e.__dict__.items():
        if isinstance(v, L.SPTensor): out[k] = v.data
        elif torch.is_tensor(v) and v.is_cuda: out[k] = v
        elif isinstance(v, (list, tuple)):
            for i, x in enumerate(v):
                if isinstance(x, L.SPTensor): out[f"{k}[{i}]"] = x.data
        elif isinstance(v, dict):
            for kk, x in v.items():
                if torch.is_tensor(x) and x.is_cuda: out[f"{k}[{kk}]"] = x
    return out
tA, tB = tensors(A), tensors(B)
g = torch.Generator(device="cpu").manual_seed(1)
for k, t in tA.items():
    if t.dtype in (torch.bfloat16, torch.float32) and k in tB and tB[k].shape == t.shape:
        if t.dtype == torch.float32:
            t.copy_((0.3 * torch.randn(t.shape, generator=g)).to(dev))
        else:
            x = 0.3 * torch.randn(t.shape[1:], generator=g)
            hi = x.to(torch.bfloat16); lo = (x - hi.float()).to(torch.bfloat16)
            t[0].copy_(hi.to(dev)); t[1].copy_(lo.to(dev))
A.Z.copy_(torch.rand(A.Z.shape, generator=g).to(dev))
names = [n for n, op in A.op.items() if getattr(op, "version", 0) == 8]
print("conv6 ops:", names)
for n in names:
    for k in tA:
        if k in tB and tB[k].shape == tA[k].shape: tB[k].copy_(tA[k])
    torch.cuda.synchronize()
    A.op[n](); B.op[n]()
    torch.cuda.synchronize()
    worst = []
    for k in tA:
        if k in tB and tB[k].shape == tA[k].shape and tA[k].dtype in (torch.bfloat16, torch.float32):
            a, b = tA[k].float(), tB[k].float()
            if a.dim() == 3 and tA[k].dtype == torch.bfloat16: a, b = a[0] + a[1], b[0] + b[1]
            nan = int((~torch.isfinite(a)).sum())
            d = float((a - b).abs().nan_to_num(1e9).max())
            if nan or d > 1e-3: worst.append((k, nan, d))
    print(f"{n:12s} v{B.op[n].version}", "OK" if not worst else worst)
