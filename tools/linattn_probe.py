"""Linear-attention launch pair (K^T V sums + apply) at the 1/16 scale of config 2 (GPU box): python tools/linattn_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ppmstereo_amd import _lib as L  # noqa: E402
from ppmstereo_amd.weights import hash_normal  # noqa: E402

dev = "cuda:0"
lib = L.load()
for T, n, heads, d in ((5, 640, 8, 48), (5, 640, 8, 32), (5, 3680, 8, 48)):
    Cc = heads * d
    P = T * n
    Q, K, V = (torch.nn.functional.elu(hash_normal((P, Cc), 1)) + 1).to(dev), (torch.nn.functional.elu(hash_normal((P, Cc), 2)) + 1).to(dev), (hash_normal((P, Cc), 3) / n).to(dev)
    ws = torch.zeros(int(lib.ppms_linear_attention_workspace_floats(T, n, heads, d)), device=dev)
    o = L.SPTensor(P, Cc, dev)
    run = lambda: L.check(lib.ppms_linear_attention(Q.data_ptr(), Cc, K.data_ptr(), Cc, V.data_ptr(), Cc, ws.data_ptr(), o.view(), T, n, heads, d, L.stream_ptr()))
    for _ in range(3):
        run()
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(30)]
    for a, b in ev:
        a.record()
        run()
        b.record()
    torch.cuda.synchronize()
    ms = sorted(a.elapsed_time(b) for a, b in ev)
    print(f"T={T} n={n} dh={d}: kv + apply {ms[len(ms) // 2] * 1e3:.1f} us (median of 30), checksum {o.to_f32().double().sum().item():.9e}")
