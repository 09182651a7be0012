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


def barrier():
    if dist.is_initialized():
        dist.barrier()


def max_over_ranks(value: float, device="cpu") -> float:
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def sum_over_ranks(value: float, device="cpu") -> float:
    if not dist.is_initialized():
        return value
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.SUM)
    return float(t.item())


def gather_kept_frames(local: List[tuple], num_frames: int, H: int, W: int, device="cpu") -> Optional[torch.Tensor]:
    """local: [(first_frame, disparity (k,1,H,W))] produced by this rank's windows.  Every rank contributes its frames
    into a zero (num_frames,1,H,W) canvas; frames are disjoint across ranks, so one SUM all-reduce assembles the video
    (a single end-of-job exchange, not a data-path collective)."""
    canvas = torch.zeros(num_frames, 1, H, W, dtype=torch.float32, device=device)
    for first, disp in local:
        canvas[first:first + disp.shape[0]] = disp.to(device)
    if dist.is_initialized():
        dist.all_reduce(canvas, op=dist.ReduceOp.SUM)
    return canvas
