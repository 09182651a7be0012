"""GPU: the frame-sharded engine (SURVEY.md section 8e level 2, BASELINE configs 4-5) on the real kernels.  One GPU box has one
GPU, so two ranks share it and talk over gloo (dist.FrameShard stages device tensors through the host in that case; on a
multi-GPU node the same calls go through RCCL).  Each rank runs its block of frames of a T = 8 window through the whole 3-scale
cascade and compares with the unsharded cascade run in the same process."""
import os
import socket
import sys

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, out, c3d):
    sys.path.insert(0, ROOT)
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from ppmstereo_amd import dist as D
    from ppmstereo_amd import weights as Wm
    from ppmstereo_amd.ppmstereo import PPMStereoHotPath
    from ppmstereo_amd.synth import synth_cascade_feats
    D.init_from_env("gloo")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    T, H, W, iters = 8, 64, 256, 4
    model = PPMStereoHotPath(use_convex_3d=c3d).load_hot_path_weights(Wm.hot_path_weights(use_convex_3d=c3d)).to(dev).eval()
    feats = synth_cascade_feats(T, H, W, seed=5)
    shard = D.FrameShard(rank, world, T)
    local = {k: v[shard.lo:shard.hi].to(dev) for k, v in feats.items()}
    p1, u1 = [], []
    d_sh, c_sh = model.cascade(local, iters, T, p1, u1, shard=shard)
    torch.cuda.synchronize()
    D.barrier()
    d_full, c_full = model.cascade({k: v.to(dev) for k, v in feats.items()}, iters, T)
    torch.cuda.synchronize()
    sl = slice(shard.lo, shard.hi)
    res = dict(disp=(d_sh - d_full[sl]).abs().max().item(), unc=(c_sh - c_full[sl]).abs().max().item(), npred=len(p1),
               finite=bool(torch.isfinite(d_sh).all()), scale=d_full.abs().max().item())
    torch.save(res, out + f".{rank}")
    D.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.parametrize("world,c3d", [(2, False), (2, True), (4, False)])
def test_sharded_cascade_equals_unsharded_on_the_gpu(tmp_path, world, c3d):
    """All three scales (update_block16 with its temporal attention gather, 08, 04), 2 / 2 / 4 iterations, K / V / confidence
    all-gathers and every temporal halo, against the same cascade on one rank.  The kernels compute every pixel from the same
    operands whichever rank holds it (zero halos stand for the window's zero padding); what differs is the launch plan -- a rank
    with half the frames has half the tiles, so the K-sliced convolutions of the small scales split their sums differently -- i.e.
    fp32 summation order, which the loop amplifies like any other rounding (measured 2.5e-4 px after 8 predictions).  A missing
    exchange shows as 0.1-1 px (tests/test_dist_gloo.py has the fault-injection twin).
    world = 4: two frames per rank (= the halo depth), interior ranks with two neighbours, three peers in the direct all-gather."""
    out = str(tmp_path / "sh.pt")
    mp.spawn(_worker, args=(world, _free_port(), out, c3d), nprocs=world, join=True)
    for r in range(world):
        res = torch.load(out + f".{r}")
        assert res["finite"] and res["npred"] == 8
        assert res["disp"] <= 5e-5 * max(1.0, res["scale"]), res
        assert res["unc"] <= 2e-4, res


def _window_worker(rank, world, port, out):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from ppmstereo_amd import dist as D
    from ppmstereo_amd import weights as Wm
    from ppmstereo_amd.ppmstereo import PPMStereo
    from stub_encoders import StubCNet, StubFNet, frame_video
    D.init_from_env("gloo")
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    m = PPMStereo.shipped(fnet=StubFNet(), cnet=StubCNet(), sst=None).load_hot_path_weights(Wm.hot_path_weights()).to(dev).eval()
    video = frame_video(45, 60, 250)                        # kernel_size 20: windows [0,20) [10,30) [20,40) [30,45)
    sharded = m.forward_batch_test({"stereo_video": video}, kernel_size=20, iters=2, shard_ranks=True)
    torch.cuda.synchronize()
    D.barrier()
    res = dict(shape=tuple(sharded["disparity"].shape))
    if rank == 0:
        full = m.forward_batch_test({"stereo_video": video}, kernel_size=20, iters=2)
        res.update(equal=bool(torch.equal(full["disparity"], sharded["disparity"]) and torch.equal(full["uncertainties"], sharded["uncertainties"])),
                   nonzero=bool((sharded["disparity"].abs().amax((1, 2, 3)) > 0).all()))
    torch.save(res, out + f".{rank}")
    D.barrier()
    torch.distributed.destroy_process_group()


def test_window_sharded_forward_batch_test_on_the_gpu(tmp_path):
    """SURVEY 8e level 1 with a driver on the GPU: forward_batch_test(shard_ranks=True) deals the sliding windows of a 45-frame video
    round-robin over two ranks (independent units, no data-path collective), gathers the kept frames once at the end
    (dist.gather_kept_frames) and must return, on every rank, exactly the single-process result (same kernels per window)."""
    out = str(tmp_path / "win.pt")
    mp.spawn(_window_worker, args=(2, _free_port(), out), nprocs=2, join=True)
    r0, r1 = torch.load(out + ".0"), torch.load(out + ".1")
    assert r0["shape"] == r1["shape"] == (45, 1, 60, 250)
    assert r0["equal"] and r0["nonzero"], r0


def _bench_line(argv, timeout=560):
    """python bench.py <argv> with no launcher around it (bench.launch_ranks starts the ranks), gloo between the ranks: both share this box's
    one GPU, device tensors are staged through the host by dist.FrameShard -- a rehearsal of the N > 1 bench paths with the REAL kernels, never
    a measurement (on a multi-GPU node the same command line runs over RCCL, one rank per GPU)."""
    import json
    import subprocess
    env = dict(os.environ, PPMS_DIST_BACKEND="gloo", OMP_NUM_THREADS="2")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + argv, capture_output=True, text=True, cwd=ROOT, env=env, timeout=timeout)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.parametrize("T,scaling", [(5, "weak"), (8, "strong")])
def test_bench_gpus_2_runs_end_to_end_on_one_gpu_over_gloo(T, scaling):
    """`python bench.py --gpus 2`: two ranks launched by bench.py itself; T = 5 -> clip replicas (what the driver's scaling run does at
    config 2), T = 8 -> the window's frames sharded 4 + 4 with every exchange of dist.FrameShard inside the timed steps."""
    out = _bench_line(["--gpus", "2", "--T", str(T), "--H", "64", "--W", "256", "--iters", "4", "--steps", "2", "--warmup", "1",
                       "--no-cpu-baseline", "--no-encoders", "--sharded-T", "8", "--sharded-iters", "4", "--sharded-steps", "1"])
    assert out["n_gpus"] == 2 and out["scaling"] == scaling and out["value"] > 0 and out["steps"] == 2
    px = T * 64 * 256
    want = (1 if scaling == "strong" else 2) * 2 * px / (out["ms_per_step"] * 2e-3)
    assert abs(out["value"] - want) <= 1e-3 * want          # value = whole-job pixels over the max-over-ranks time of exactly K steps
    assert ("sharded" in out["config"]["parallelism"]) == (scaling == "strong")
    if scaling == "weak":                # replicas as the measurement: ONE extra frame-sharded window (here T = 8, 4 + 4 frames) on the same ranks, checked
        sh, chk = out["sharded"], out["sharded_check"]          # against the unsharded result on rank 0 before it is timed
        assert sh["T"] == 8 and sh["frames_per_gpu"] == 4 and sh["ms_per_window"] > 0 and sh["scaling"] == "strong" and "rehearsal" in sh["note"]
        assert chk["passed"] and chk["max_abs_disparity_diff_px"] <= chk["tolerance_px"]
    else:
        assert out["sharded"] is None
