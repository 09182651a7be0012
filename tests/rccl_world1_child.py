"""Child process of tests/test_gpu_rccl_world1.py: ONE rank with an RCCL ("nccl") process group on the MI355X.  Rendezvous first (environment as
bench.launch_ranks / torch.distributed.run set it), then the GPU; every exchange of the frame-sharded cascade goes through the communicator
(dist.FrameShard(force_comm=True)): library all-gather, grouped point-to-point gather, asynchronous handles waited on the compute stream.
Prints one JSON line."""
import datetime
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import torch.distributed as dist

from ppmstereo_amd import dist as D

assert os.environ["WORLD_SIZE"] == "1" and os.environ["RANK"] == "0"
# the rendezvous comes BEFORE anything touches the GPU, as under the launcher
dist.init_process_group(backend="nccl", rank=0, world_size=1, timeout=datetime.timedelta(seconds=120))
rank, world, local = D.init_from_env(force=True)             # (already initialised: reads the environment only)
assert (rank, world) == (0, 1) and dist.get_backend() == "nccl"
torch.cuda.set_device(local)
dev = torch.device("cuda", torch.cuda.current_device())
assert D.comm_device() == dev, "device tensors must go through the nccl branch"
D.barrier()
res = dict(backend=dist.get_backend(), max_over_ranks=D.max_over_ranks(3.5), sum_over_ranks=D.sum_over_ranks(2.25))

from ppmstereo_amd import weights as Wm
from ppmstereo_amd.ppmstereo import PPMStereoHotPath
from ppmstereo_amd.synth import synth_cascade_feats

T, H, W, iters = 6, 64, 256, 4
model = PPMStereoHotPath().load_hot_path_weights(Wm.hot_path_weights()).to(dev).eval()
feats = {k: v.to(dev) for k, v in synth_cascade_feats(T, H, W, seed=5).items()}
shard = D.FrameShard(0, 1, T, force_comm=True)
assert shard.force_comm
# the primitives on device tensors
x = torch.arange(24, dtype=torch.float32, device=dev).view(6, 4)
g, h = shard.all_gather(x, async_op=True)
h.wait()
a_out, b_out = torch.zeros(6, 4, device=dev), torch.zeros(6, 3, dtype=torch.bfloat16, device=dev)
b_in = torch.arange(18, device=dev).view(6, 3).to(torch.bfloat16)
hd = shard.gather_many([(x, a_out), (b_in, b_out)], async_op=True)          # grouped send + receive to this rank itself, two tensors in one batch
hd.wait()
buf = torch.ones(T + 4, 5, device=dev)
shard.halo_many([(buf, 2)], async_op=True).wait()                            # no neighbour on either side: nothing travels, the zero padding stays the caller's
torch.cuda.synchronize()
res.update(all_gather_ok=bool(torch.equal(g, x)), gather_many_ok=bool(torch.equal(a_out, x) and torch.equal(b_out, b_in)))
# the whole cascade, sharded over the one rank, against the unsharded one
p1, u1 = [], []
d_sh, c_sh = model.cascade(feats, iters, T, p1, u1, shard=shard)
torch.cuda.synchronize()
eng = model.update_block04.engine(T, H // 4, W // 4, dev, shard)
res["engine_sharded"] = eng.shard is not None
d_full, c_full = model.cascade(feats, iters, T)
torch.cuda.synchronize()
res.update(npred=len(p1), finite=bool(torch.isfinite(d_sh).all()), disp_equal=bool(torch.equal(d_sh, d_full)), unc_equal=bool(torch.equal(c_sh, c_full)),
           disp_maxdiff=float((d_sh - d_full).abs().max().item()), scale=float(d_full.abs().max().item()))
kept = D.gather_kept_frames([(0, d_full[:3]), (3, d_full[3:])], T, H, W)    # the end-of-job exchange of window-sharded runs, on device
res["gather_kept_ok"] = bool(torch.equal(kept.to(dev), d_full.float()))
D.barrier()
dist.destroy_process_group()
print(json.dumps(res), flush=True)
