"""CPU, world_size 2 over gloo: the N > 1 path (window sharding, end-of-job gather, max-over-ranks timing)."""
import os
import socket
import sys

import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from ppmstereo_amd import dist as D
    from ppmstereo_amd.ppmstereo import shard_windows, window_plan
    r, w, _ = D.init_from_env("gloo")
    assert (r, w) == (rank, world)
    N, H, W = 40, 4, 6
    video = torch.arange(N, dtype=torch.float32)[:, None, None, None].expand(N, 1, H, W)    # "disparity" of frame f is f
    plan = window_plan(N, 20)
    local = []
    for (s, e, a, b) in shard_windows(plan, rank, world):
        window_out = video[s:e] * 1.0                       # stands in for one window through the cascade
        local.append((s + a, window_out[a:b]))
    D.barrier()
    full = D.gather_kept_frames(local, N, H, W)
    t = D.max_over_ranks(1.0 + rank)
    tot = D.sum_over_ranks(10.0)
    if rank == 0:
        torch.save(dict(full=full, t=t, tot=tot), out)
    torch.distributed.destroy_process_group()


def test_window_sharding_two_ranks(tmp_path):
    out = str(tmp_path / "r0.pt")
    port = _free_port()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    res = torch.load(out)
    want = torch.arange(40, dtype=torch.float32)[:, None, None, None].expand(40, 1, 4, 6)
    assert torch.equal(res["full"], want), "every frame exactly once, in order, identical to the single-process stitching"
    assert res["t"] == 2.0 and res["tot"] == 20.0
