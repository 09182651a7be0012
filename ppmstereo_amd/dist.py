"""One process per GPU over torch.distributed (backend "nccl" = RCCL on ROCm; "gloo" for CPU rehearsals).

The hot path of one window does not need a collective at BASELINE config 2 (T = 5 frames fit one GPU and do not
divide across 2/4/8 ranks without changing the temporal encoding and top-k pick -- SURVEY.md section 8e), so ranks
run independent units ("replicas only"): either replicas of the clip (bench.py) or the windows of a long video
dealt round-robin (``shard_windows``), with one gather of the kept disparities at the end.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None) -> tuple:
    """(rank, world, local_rank) from RANK / WORLD_SIZE / LOCAL_RANK; initialises the default group when world > 1."""
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        dist.init_process_group(backend=backend, rank=rank, world_size=world)
    return rank, world, local


def comm_device() -> torch.device:
    """Where collective operands must live: the current GPU under RCCL ("nccl"), the host under gloo."""
    if dist.is_initialized() and dist.get_backend() == "nccl":
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def barrier():
    if dist.is_initialized():
        if dist.get_backend() == "nccl":
            dist.barrier(device_ids=[torch.cuda.current_device()])
        else:
            dist.barrier()


def max_over_ranks(value: float, device=None) -> float:
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=comm_device() if device is None else device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device=None) -> float:
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=comm_device() if device is None else device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_kept_frames(local: List[tuple], num_frames: int, H: int, W: int, device=None) -> Optional[torch.Tensor]:
    """local: [(first_frame, disparity (k,1,H,W))] produced by this rank's windows.  One end-of-job exchange (not a
    data-path collective): every rank all-gathers only the frames it owns (padded to the largest per-rank count) plus
    their frame indices, and places the received frames into the (num_frames,1,H,W) video."""
    device = comm_device() if device is None else torch.device(device)
    idx = [first + i for first, disp in local for i in range(disp.shape[0])]
    frames = [disp.to(device=device, dtype=torch.float32) for _, disp in local]
    mine = torch.cat(frames, 0) if frames else torch.zeros(0, 1, H, W, dtype=torch.float32, device=device)
    out = torch.zeros(num_frames, 1, H, W, dtype=torch.float32, device=device)
    if not dist.is_initialized():
        if idx:
            out[torch.tensor(idx, device=device)] = mine
        return out
    world = dist.get_world_size()
    cnt = torch.tensor([len(idx)], dtype=torch.int64, device=device)
    cnts = [torch.zeros_like(cnt) for _ in range(world)]
    dist.all_gather(cnts, cnt)
    cap = max(1, int(max(c.item() for c in cnts)))
    pad_idx = torch.full((cap,), -1, dtype=torch.int64, device=device)
    pad_frames = torch.zeros(cap, 1, H, W, dtype=torch.float32, device=device)
    if idx:
        pad_idx[:len(idx)] = torch.tensor(idx, device=device)
        pad_frames[:len(idx)] = mine
    all_idx = [torch.empty_like(pad_idx) for _ in range(world)]
    all_frames = [torch.empty_like(pad_frames) for _ in range(world)]
    dist.all_gather(all_idx, pad_idx)
    dist.all_gather(all_frames, pad_frames)
    for ii, ff in zip(all_idx, all_frames):
        keep = ii >= 0
        if keep.any():
            out[ii[keep]] = ff[keep]
    return out
