"""Runs the memory-attention kernel alone at the BASELINE 1/4-scale shape (T=5, n=10240, 5 picked frames) a few times:
target for rocprofv3 --pmc passes.  Put the interpreter binary itself after `--` (no env / bash / shebang hop: the profiler's
preloaded library has initialised the GPU, and an exec from such a process takes the box down):
    rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_fetch -- /usr/bin/python3 tools/attn_probe.py 4 1
    rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_write -- /usr/bin/python3 tools/attn_probe.py 4 1
arguments: [reps] [split 0/1] [frames per workgroup of the 64-query kernel: 0 = automatic, 1, 2] [p_format: 1 = fp16 P~ (default), 0 = bf16]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppmstereo_amd import _lib as L
from ppmstereo_amd.weights import hash_normal
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
split = int(sys.argv[2]) if len(sys.argv) > 2 else 1
frames = int(sys.argv[3]) if len(sys.argv) > 3 else 0
p_format = int(sys.argv[4]) if len(sys.argv) > 4 else 1
dev = "cuda:0"
T, n = 5, 10240
qb = hash_normal((T, n, 128), 950).to(torch.bfloat16).to(dev)
kb = hash_normal((T, 5, n, 128), 951).to(torch.bfloat16).to(dev)
vt = L.vt_image(hash_normal((T, 128, n), 952), p_format).to(dev)
sel = torch.arange(5, dtype=torch.int32)[None].expand(T, 5).contiguous().to(dev)
X = L.SPTensor(T * n, 256, dev)
beta = torch.tensor([0.5], device=dev)
lib = L.load()
ws = torch.empty(int(lib.ppms_mem_attn_workspace_bytes(T, 5, n)), dtype=torch.uint8, device=dev) if split else None
ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(reps)]
for a, b in ev:
    a.record()
    L.check(lib.ppms_mem_attn(qb.data_ptr(), kb.data_ptr(), vt.data_ptr(), sel.data_ptr(), 5, 0.0522, beta.data_ptr(), X.view(0, 128), X.view(128, 128),
                              None, T, n, L.ptr(ws), frames, p_format, L.stream_ptr()))
    b.record()
torch.cuda.synchronize()
ts = sorted(a.elapsed_time(b) for a, b in ev)
print(f"mem_attn split={split}: median {ts[len(ts)//2]:.3f} ms  -> {1.34217728/ts[len(ts)//2]*1e3:.0f} TFLOP/s")
