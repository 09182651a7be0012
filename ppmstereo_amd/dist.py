"""One process per GPU over torch.distributed (backend "nccl" = RCCL on ROCm; "gloo" for CPU rehearsals).

Two levels of parallelism:

1. Independent units, no data-path collective: replicas of a clip (bench.py at BASELINE config 2 -- T = 5 frames do not
   divide across 2/4/8 ranks without changing the temporal encoding and the top-k pick) or the sliding windows of a
   long video dealt round-robin (``shard_windows``), with one gather of the kept disparities at the end
   (``gather_kept_frames``).
2. ``FrameShard``: the T frames of ONE window in contiguous blocks of f = T / G frames per rank (BASELINE configs 4-5:
   T = 40, 5 frames per GPU).  What couples frames inside forward_update_block, and how it is exchanged:

   ====================================  ======================  ==============================================================
   coupling (reference lines)            when                    exchange
   ====================================  ======================  ==============================================================
   pooled q / k frame descriptors        once per scale          all-gather, (f, 2, h/4*w/4) fp32 per rank  (ppmstereo.py:397-423)
   memory keys K (constant per scale)    once per scale          all-gather, f*n*128 fp32 per rank: kept fp32 so that
                                                                 bf16(K s + PE) rounds exactly as unsharded         (:524-548)
   memory values V (new every iteration) every iteration         all-gather, f*128*n bf16 per rank (the reference casts V to bf16) (:550)
   frame confidences (QAM)               every iteration         all-gather, f*nblk fp32 partial sums; every rank then runs the
                                                                 same T x T pick and keeps the rows of its clips    (:505-513)
   GRU pass T, (5,1,1) convs             every iteration, twice  +-2-frame halo of [h | mf, mfg] before the z/r conv and of
                                                                 r*h before the q conv (inp's halo: once per scale)
                                                                 (ppmtereo_update.py:281-289, 305-310)
   FlowHead3D 3x3x3 (+ mask_3d head)     every iteration         +-1-frame halo of the new hidden state, +-1 of the 54-channel
                                                                 pre-gather flow-head output                         (:670-678)
   convex_upsample_3d (use_convex_3d)    every iteration         +-1-frame halo of the 2-channel flow     (ppmstereo.py:199-228)
   update_block16 TimeAttnBlock          every iteration (1/16)  all-gather of x = [inp, mf, mfg], f*n*384 (640 pixels per frame)
   ====================================  ======================  ==============================================================

   Contiguous blocks make every halo a nearest-neighbour exchange (one xGMI link each way).  The per-iteration all-gather is
   DIRECT (``gather_many``): one grouped batch of point-to-point operations in which every rank sends its block to each of the
   other ranks and receives theirs straight into place -- xGMI is a full mesh of point-to-point links, so the 7 transfers of a rank
   travel on 7 different links at once, whereas a ring would be bound by one ~153 GB/s link; one group also carries several tensors
   (the values V and the frame confidences share one exchange).  The once-per-scale gathers use the library collective
   (``all_gather``: RCCL picks the algorithm).  Several halo'd tensors likewise share ONE batch (``halo_many``), and an exchange can
   be left in flight (``async_op``) while the compute stream goes on: RCCL runs it on its own stream, ``wait()`` makes the compute
   stream wait for it right where the data is read.  Sizes at 320x512, 1/4 scale, 5 frames per GPU: K 13 MB per rank once per
   scale, V 6.6 MB per rank and iteration, halos 2 x 15.7 MB ([h | x], split-bf16 = 4 B per value) + 2 x 5.2 MB (r*h) + small ones
   per iteration.  Per-iteration schedule and its latency budget: docs/LOG_r01_r05.md section 6.
"""
from __future__ import annotations

import os
from typing import List, Optional

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None, force: bool = False) -> tuple:
    """(rank, world, local_rank) from RANK / WORLD_SIZE / LOCAL_RANK; initialises the default group when world > 1 (force: also for a
    world of one -- the communicator of a single rank, which is how a one-GPU box exercises the RCCL code path: tests/test_gpu_rccl_world1.py)."""
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", str(rank)))
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:          # RCCL on GPUs; PPMS_DIST_BACKEND=gloo rehearses the multi-rank paths on a box with fewer GPUs than ranks
            backend = os.environ.get("PPMS_DIST_BACKEND") or ("nccl" if torch.cuda.is_available() else "gloo")
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


class _Done:
    def wait(self):
        return None


class _Staged:
    """Handle of an exchange staged through host memory (gloo with device tensors): completes the copy back on wait()."""

    def __init__(self, finish):
        self._finish = finish

    def wait(self):
        if self._finish is not None:
            self._finish()
            self._finish = None


class FrameShard:
    """Contiguous block of f = T / world frames of one window on each rank, and the two exchange primitives the sharded loop
    is built from: ``all_gather`` (frame blocks of every rank, in rank order) and ``halo`` (boundary frames with the two
    neighbours).  Works on host tensors (gloo) and on device tensors (RCCL; under gloo device tensors are staged through the
    host, which is how one GPU box rehearses two ranks).  Every rank must call the same sequence of exchanges."""

    HALO = 2            # frames of slack on both sides of a rank's block in halo'd buffers (the GRU's (5,1,1) convs need 2)

    def __init__(self, rank: int, world: int, T: int, group=None, force_comm: bool = False):
        """force_comm: a world of ONE still sends its exchanges through the process group's backend -- the library all-gather on one rank, the
        direct gather as a grouped send + receive to the rank itself -- instead of the local copies a single rank needs.  The results are the same
        bytes; it exists so that a one-GPU box runs the RCCL branch of every exchange (communicator, grouped point-to-point launch, asynchronous
        handles, stream waits) before an 8-GPU node ever does."""
        if T % world != 0:
            raise ValueError(f"FrameShard: {T} frames do not divide over {world} ranks")
        self.rank, self.world, self.T, self.group = rank, world, T, group
        # (RCCL / NCCL accept a send to the own rank inside a group; gloo does not: there a single rank keeps its local copies)
        self.force_comm = bool(force_comm and world == 1 and dist.is_initialized() and dist.get_backend(group) == "nccl")
        self.f = T // world
        if world > 1 and self.f < self.HALO:
            raise ValueError("FrameShard: at least 2 frames per rank (the temporal GRU pass reads +-2 frames: one neighbour each side)")
        self.lo, self.hi = rank * self.f, (rank + 1) * self.f

    # ------------------------------------------------------------------ helpers
    def _stage(self, t: torch.Tensor) -> bool:
        return (self.world > 1 or self.force_comm) and t.is_cuda and dist.get_backend(self.group) != "nccl"

    # ------------------------------------------------------------------ all-gather of frame blocks
    def all_gather(self, local: torch.Tensor, out: Optional[torch.Tensor] = None, async_op: bool = False):
        """local: this rank's block, leading dimension = frames (or any per-rank leading extent); returns (out, handle) with
        out = the blocks of all ranks concatenated along dimension 0 in rank order.  async_op: the caller waits on the handle."""
        local = local.contiguous()
        if out is None:
            out = torch.empty((self.world * local.shape[0],) + tuple(local.shape[1:]), dtype=local.dtype, device=local.device)
        if self.world == 1 and not self.force_comm:
            out.copy_(local)
            return out, _Done()
        if self._stage(local):
            h_out = torch.empty(out.shape, dtype=out.dtype)
            dist.all_gather_into_tensor(h_out, local.cpu(), group=self.group)
            out.copy_(h_out)
            return out, _Done()
        work = dist.all_gather_into_tensor(out, local, group=self.group, async_op=async_op)
        return out, (work if async_op else _Done())

    def gather_many(self, pairs, async_op: bool = False):
        """Direct all-gather of SEVERAL tensors in one grouped exchange.  pairs: [(local, out)], out = the blocks of all ranks along
        dimension 0 in rank order (out[rank * len(local):][:len(local)] is written with a local copy).  One batch_isend_irecv: this
        rank's block of every tensor goes to each peer, the peers' blocks are received straight into their place in ``out``
        (contiguous slices), i.e. world - 1 sends + world - 1 receives per tensor in ONE group.  Returns a handle; with
        async_op the caller waits on it right before the gathered data is read."""
        if self.world == 1 and not self.force_comm:
            for local, out in pairs:
                out.copy_(local)
            return _Done()
        ops, finish = [], []
        for local, out in pairs:
            local = local.contiguous()
            m = local.shape[0]
            assert out.shape[0] == self.world * m and out.is_contiguous(), "gather_many: out must hold world blocks along dimension 0"
            stage = self._stage(local)
            src = local.cpu() if stage else local
            for peer in range(self.world):
                dst = out[peer * m:(peer + 1) * m]
                if peer == self.rank and not (self.force_comm and dst.data_ptr() != local.data_ptr()):
                    if dst.data_ptr() != local.data_ptr():
                        dst.copy_(local)
                    continue
                ops.append(dist.P2POp(dist.isend, src, peer, group=self.group))
                if stage:
                    tmp = torch.empty(dst.shape, dtype=dst.dtype)
                    ops.append(dist.P2POp(dist.irecv, tmp, peer, group=self.group))
                    finish.append((dst, tmp))
                else:
                    ops.append(dist.P2POp(dist.irecv, dst, peer, group=self.group))
        works = dist.batch_isend_irecv(ops)

        def done():
            for w in works:
                w.wait()
            for dst, tmp in finish:
                dst.copy_(tmp)

        if async_op and not finish:
            return _Staged(done)
        done()
        return _Done()

    # ------------------------------------------------------------------ halo exchange with the two neighbours
    def halo(self, buf: torch.Tensor, k: int, async_op: bool = False):
        return self.halo_many([(buf, k)], async_op)

    def halo_many(self, bufs, async_op: bool = False):
        """bufs: [(buf, k)], every buf (HALO + f + HALO frames, ...) with the frame axis as dimension 0: buf[HALO - k:HALO] receives
        the left neighbour's last k frames, buf[HALO + f:HALO + f + k] the right neighbour's first k frames; rank 0's left and the
        last rank's right halo are left untouched (zeros = the convolution's zero padding at the window's ends).  All tensors
        travel in ONE batch of point-to-point operations (<= 4 per tensor).  Returns a handle (wait() before a halo is read)."""
        H, f = self.HALO, self.f
        if self.world == 1:
            return _Done()
        left, right = self.rank - 1, self.rank + 1
        ops, finish = [], []
        stage = any(self._stage(b) for b, _ in bufs)

        def send(src, peer):
            src = src.contiguous()
            ops.append(dist.P2POp(dist.isend, src.cpu() if stage else src, peer, group=self.group))

        def recv(dst, peer):
            if stage:
                tmp = torch.empty(dst.shape, dtype=dst.dtype)
                ops.append(dist.P2POp(dist.irecv, tmp, peer, group=self.group))
                finish.append((dst, tmp))
            elif dst.is_contiguous():
                ops.append(dist.P2POp(dist.irecv, dst, peer, group=self.group))
            else:
                tmp = torch.empty(dst.shape, dtype=dst.dtype, device=dst.device)
                ops.append(dist.P2POp(dist.irecv, tmp, peer, group=self.group))
                finish.append((dst, tmp))

        for buf, k in bufs:
            assert buf.shape[0] == f + 2 * H and 1 <= k <= H
            if left >= 0:
                send(buf[H:H + k], left)
                recv(buf[H - k:H], left)
            if right < self.world:
                send(buf[H + f - k:H + f], right)
                recv(buf[H + f:H + f + k], right)
        if not ops:
            return _Done()
        works = dist.batch_isend_irecv(ops)

        def done():
            for w in works:
                w.wait()
            for dst, tmp in finish:
                dst.copy_(tmp)

        if async_op and not stage:
            return _Staged(done)
        done()
        return _Done()
