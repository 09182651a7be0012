"""CPU, world sizes 2, 4 and 8 over gloo: the N > 1 paths (window sharding with its end-of-job gather, max-over-ranks timing, frame sharding of one window)."""
import os
import socket
import sys

import pytest
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


def _shard_worker(rank, world, port, out, c3d, drop=None, T=8, blocks=None):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    torch.set_num_threads(max(1, 8 // world))
    from oracle import ppm_oracle as O
    from ppmstereo_amd import dist as D
    from ppmstereo_amd import weights as Wm
    from ppmstereo_amd.synth import synth_scale_inputs
    import sharded_oracle
    from sharded_oracle import forward_update_block_sharded
    sharded_oracle.DROP = drop                              # fault injection: skip one exchange (the test must then fail)
    r, w, _ = D.init_from_env("gloo")
    h, wd, iters = 8, 32, 2
    shard = D.FrameShard(rank, world, T)
    W = Wm.hot_path_weights(use_convex_3d=c3d)
    res = {}
    for tag, ai, isc, attn, mh in (("update_block04", 2, 1, False, True), ("update_block16", 0, 4, True, False)):
        if blocks is not None and tag not in blocks:
            continue
        d = synth_scale_inputs(T, h, wd, seed=77, with_mhs=mh, frame_contrast=1.0)
        sl = slice(shard.lo, shard.hi)
        pyr = O.corr_pyramid(d["fmap1"][sl], d["fmap2"][sl])                  # per frame: built from the local frames only
        preds, uncs = [], []
        fo, net, mhs = forward_update_block_sharded(shard, W[tag], W[f"att.{ai}"], pyr, d["flow"][sl], d["net"][sl], d["inp"][sl],
                                                    None if d["mhs"] is None else d["mhs"][sl], iters, isc, attn, preds, uncs)
        res[tag] = dict(fo=fo, net=net, mhs=mhs, pred=preds[-1], unc=uncs[-1], lo=shard.lo, hi=shard.hi)
    torch.save(res, out + f".{rank}")
    D.barrier()
    torch.distributed.destroy_process_group()


def test_frame_sharded_loop_equals_unsharded(tmp_path):
    """SURVEY.md section 8e level 2 (BASELINE configs 4-5): a T = 8 window sharded 4 frames per rank over two gloo ranks -- K /
    V / confidence / descriptor all-gathers, +-2 and +-1 frame halos (dist.FrameShard), oracle math per rank -- gives the same
    flow, hidden state and predictions as the unsharded loop, for update_block04 and update_block16 (time attention), with the
    2-D and the 3-D convex upsampling."""
    from oracle import ppm_oracle as O
    from ppmstereo_amd import weights as Wm
    from ppmstereo_amd.synth import synth_scale_inputs
    for c3d in (False, True):
        out = str(tmp_path / f"shard{int(c3d)}.pt")
        port = _free_port()
        mp.spawn(_shard_worker, args=(2, port, out, c3d), nprocs=2, join=True)
        parts = [torch.load(out + f".{r}") for r in range(2)]
        W = Wm.hot_path_weights(use_convex_3d=c3d)
        T, h, wd, iters = 8, 8, 32, 2
        for tag, ai, isc, attn, mh in (("update_block04", 2, 1, False, True), ("update_block16", 0, 4, True, False)):
            d = synth_scale_inputs(T, h, wd, seed=77, with_mhs=mh, frame_contrast=1.0)
            rp, ru = [], []
            rfo, rnet, rmhs = O.forward_update_block(W[tag], W[f"att.{ai}"], O.corr_pyramid(d["fmap1"], d["fmap2"]), d["flow"], d["net"], d["inp"],
                                                     d["mhs"], iters, isc, T, attn, rp, ru)
            for p_ in parts:
                sl = slice(p_[tag]["lo"], p_[tag]["hi"])
                for key, ref in (("fo", rfo), ("net", rnet), ("mhs", rmhs), ("pred", rp[-1]), ("unc", ru[-1])):
                    # not bit for bit on CPU: the fp32 conv library blocks a 4 + 4-frame tensor differently from an 8-frame one
                    # (~1e-7 relative), and a bf16 rounding of an attention operand that flips on such a difference moves the
                    # result by ~1e-4 -- the same tolerances as oracle vs reference (test_oracle_golden.py).  A dropped or
                    # misplaced exchange is two orders larger (checked below by leaving one out).
                    err = (p_[tag][key] - ref[sl]).abs().max().item()
                    tol = {"fo": 3e-4, "pred": 3e-4 * isc, "net": 6e-4, "mhs": 2e-4, "unc": 5e-5}[key]
                    assert err <= tol, (c3d, tag, key, err)


def _check_against_unsharded(parts, T, c3d, blocks):
    from oracle import ppm_oracle as O
    from ppmstereo_amd import weights as Wm
    from ppmstereo_amd.synth import synth_scale_inputs
    W = Wm.hot_path_weights(use_convex_3d=c3d)
    h, wd, iters = 8, 32, 2
    for tag, ai, isc, attn, mh in (("update_block04", 2, 1, False, True), ("update_block16", 0, 4, True, False)):
        if tag not in blocks:
            continue
        d = synth_scale_inputs(T, h, wd, seed=77, with_mhs=mh, frame_contrast=1.0)
        rp, ru = [], []
        rfo, rnet, rmhs = O.forward_update_block(W[tag], W[f"att.{ai}"], O.corr_pyramid(d["fmap1"], d["fmap2"]), d["flow"], d["net"], d["inp"],
                                                 d["mhs"], iters, isc, T, attn, rp, ru)
        covered = []
        for p_ in parts:
            sl = slice(p_[tag]["lo"], p_[tag]["hi"])
            covered += list(range(sl.start, sl.stop))
            for key, ref in (("fo", rfo), ("net", rnet), ("mhs", rmhs), ("pred", rp[-1]), ("unc", ru[-1])):
                err = (p_[tag][key] - ref[sl]).abs().max().item()
                tol = {"fo": 3e-4, "pred": 3e-4 * isc, "net": 6e-4, "mhs": 2e-4, "unc": 5e-5}[key]
                assert err <= tol, (T, len(parts), c3d, tag, key, p_[tag]["lo"], err)
        assert covered == list(range(T))


@pytest.mark.parametrize("T,c3d,blocks", [(8, False, ("update_block04", "update_block16")), (8, True, ("update_block04",)),
                                          (20, False, ("update_block04",))])
def test_frame_sharded_loop_four_ranks(tmp_path, T, c3d, blocks):
    """World size 4: the INTERIOR ranks have a left and a right neighbour (four point-to-point operations per halo'd tensor in one
    batch, dist.FrameShard.halo_many), every rank exchanges with three peers in the direct all-gather (gather_many).  T = 8: f = 2
    frames per rank = HALO, i.e. a rank's whole block is its neighbours' halo; T = 20: f = 5, BASELINE config 4's share (T = 40 over
    8 GPUs).  Oracle math per rank (tests/sharded_oracle.py), every rank's block against the unsharded loop."""
    out = str(tmp_path / "s4.pt")
    mp.spawn(_shard_worker, args=(4, _free_port(), out, c3d, None, T, blocks), nprocs=4, join=True)
    _check_against_unsharded([torch.load(out + f".{r}") for r in range(4)], T, c3d, blocks)


@pytest.mark.parametrize("T,blocks", [(16, ("update_block04", "update_block16")), (40, ("update_block04",))])
def test_frame_sharded_loop_eight_ranks(tmp_path, T, blocks):
    """World size 8 = the node BASELINE configs 4-5 are quoted on: T = 40 -> f = 5 frames per rank (config 4's exact split: six interior
    ranks, seven peers in every direct all-gather, the top-5 pick over 40 frames computed identically on all eight ranks) and T = 16 -> f = 2
    = the halo depth with the 1/16 block's time-attention gather.  gloo on the CPU, oracle math per rank; every rank's block against the
    unsharded loop.  (The RCCL transport of the same calls is unmeasured on hardware: docs/LOG_r01_r05.md section 6.)"""
    out = str(tmp_path / "s8.pt")
    mp.spawn(_shard_worker, args=(8, _free_port(), out, False, None, T, blocks), nprocs=8, join=True)
    _check_against_unsharded([torch.load(out + f".{r}") for r in range(8)], T, False, blocks)


def test_sharded_check_detects_a_dropped_exchange(tmp_path):
    """Power of the comparison above: with the r*h halo of the temporal GRU pass left out the sharded result is wrong at the
    block boundary by far more than the tolerance."""
    from oracle import ppm_oracle as O
    from ppmstereo_amd import weights as Wm
    from ppmstereo_amd.synth import synth_scale_inputs
    out = str(tmp_path / "drop.pt")
    mp.spawn(_shard_worker, args=(2, _free_port(), out, False, "rh"), nprocs=2, join=True)
    parts = [torch.load(out + f".{r}") for r in range(2)]
    W = Wm.hot_path_weights()
    T, h, wd, iters = 8, 8, 32, 2
    d = synth_scale_inputs(T, h, wd, seed=77, with_mhs=True, frame_contrast=1.0)
    rfo, rnet, _ = O.forward_update_block(W["update_block04"], W["att.2"], O.corr_pyramid(d["fmap1"], d["fmap2"]), d["flow"], d["net"], d["inp"],
                                          d["mhs"], iters, 1, T, False, [], [])
    worst = max((p_["update_block04"]["net"] - rnet[p_["update_block04"]["lo"]:p_["update_block04"]["hi"]]).abs().max().item() for p_ in parts)
    assert worst > 1e-2, worst
