"""Benchmark of the PPMStereo hot path on MI355X (contract: see the round prompt / docs/LOG_r01_r05.md section 5).

    python bench.py --gpus 1 --steps 10 --warmup 3
    python bench.py --gpus 8 --steps 10 --warmup 3          # no launcher around it: starts the 8 ranks itself (launch_ranks), exit code = theirs
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one pass of the hot path over one clip of BASELINE config 2: T=5 frames, 320x512, iters=10
(cascade 1/16 -> 1/8 -> 1/4: 3 correlation pyramid builds + 5/5/10 iterations of lookup, motion encoder,
uncertainty, QAM pick, pick-and-play memory attention, ConvGRU3D, heads, convex upsample).  Inputs (the encoder / SST
outputs the loop consumes) are synthetic and resident in HBM before the timed region.

N > 1, T = 5 (config 2): every rank runs its own replica of the clip (T=5 does not divide across ranks without changing
the result -- docs/LOG_r01_r05.md section 6), scaling "weak", value = N * pixels / max-over-ranks time.
N > 1, T = 5: behind the replica measurement the same ranks run ONE frame-sharded window (--sharded-T 40 --sharded-iters 20 = BASELINE config 4), checked
against the unsharded result on rank 0 first; it is reported under `sharded` / `sharded_check`, never in `value`.
N > 1, --T divisible by N and >= 2 frames per rank (configs 4-5, e.g. --T 40): the window's frames are SHARDED over the
ranks (ppmstereo_amd.dist.FrameShard: all-gather of the memory keys once per scale and of the values / confidences every
iteration, +-2 / +-1 frame halos for the temporal convs), scaling "strong", value = pixels / max-over-ranks time.
Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BF16_DENSE_PEAK_TFLOPS = 2500.0        # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA
HBM_PEAK_GBS = 8000.0                  # MI355X_MICROARCH.md: HBM3E 8 TB/s peak (~6.3 TB/s achievable by a streaming copy)
CONV_BOUND_TFLOPS = BF16_DENSE_PEAK_TFLOPS / 3.0     # bf16x3 split (hi*hi + hi*lo + lo*hi): 3 MFMAs per algorithmic product
BASELINE_CONFIGS = {(5, 320, 512, 10): "BASELINE config 2", (5, 736, 1280, 20): "BASELINE config 3 (720x1280 padded to 736x1280)",
                    (40, 320, 512, 20): "BASELINE config 4 (one T=40 window)", (40, 736, 1280, 20): "BASELINE config 5 (one T=40 window)"}


def _latest_profile(suffix):
    """profiles/rNN_<suffix> of the latest round (committed PMC summaries, tools/traffic_pmc.sh), or None."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_" + suffix)))
    return files[-1] if files else None


def cpu_baseline(T, H, W, iters, threads):
    """The oracle (a CPU restatement of the reference path) timed on a bounded sample of the same workload: ONE
    iteration of forward_update_block at each of the three scales plus the three pyramid builds, extrapolated with the
    real iteration counts (iters//2, iters//2, iters)."""
    from oracle import ppm_oracle as O
    from ppmstereo_amd import weights as Wm
    from ppmstereo_amd.synth import synth_cascade_feats
    torch.set_num_threads(threads)
    Wt = Wm.hot_path_weights()
    feats = synth_cascade_feats(T, H, W)
    total, sample = 0.0, 0.0
    for s, tag, att, isc, n_it in ((16, "update_block16", "att.0", 4, iters // 2), (8, "update_block08", "att.1", 2, iters // 2),
                                   (4, "update_block04", "att.2", 1, iters)):
        f1, f2 = feats[f"f1_{s}"], feats[f"f2_{s}"]
        t0 = time.perf_counter()
        pyr = O.corr_pyramid(f1, f2)
        t1 = time.perf_counter()
        flow = torch.zeros(T, 2, f1.shape[2], f1.shape[3])
        mhs = None if s == 16 else torch.zeros(T, 64, f1.shape[2], f1.shape[3])
        O.forward_update_block(Wt[tag], Wt[att], pyr, flow, feats[f"net_{s}"], feats[f"inp_{s}"], mhs, 1, isc, T, s == 16, [], [])
        t2 = time.perf_counter()
        total += (t1 - t0) + n_it * (t2 - t1)
        sample += t2 - t0
    return dict(value=T * H * W / total, unit="disparity-px/s", cores=threads, kind="port",
                sample=f"1 of {iters // 2}/{iters // 2}/{iters} iterations at each scale + 3 pyramid builds of the same T={T} {H}x{W} clip "
                       f"({sample:.1f} s of CPU work), extrapolated by iteration count to {total:.1f} s per clip; fp32 torch-CPU oracle",
                seconds_per_clip=total)


def cpu_baseline_full(T, H, W, iters, threads, runs=3):
    """SURVEY 8(d)'s procedure: the whole clip through the oracle's cascade (3 pyramid builds + iters//2, iters//2, iters iterations), 1 warm-up + `runs`
    timed runs, the median.  ~6 minutes at config 2 on 16 threads: behind --cpu-baseline full only (the default line carries the sampled form;
    profiles/rNN_cpu_baseline_full.json keeps one full run with the sampled figure of the same process beside it)."""
    from oracle import ppm_oracle as O
    from ppmstereo_amd import weights as Wm
    from ppmstereo_amd.synth import synth_cascade_feats
    torch.set_num_threads(threads)
    Wt = Wm.hot_path_weights()
    feats = synth_cascade_feats(T, H, W)
    ts = []
    for r in range(runs + 1):
        t0 = time.perf_counter()
        O.cascade(Wt, feats, iters, T, [], [])
        ts.append(time.perf_counter() - t0)
    timed = sorted(ts[1:])
    med = timed[len(timed) // 2]
    return dict(value=T * H * W / med, unit="disparity-px/s", cores=threads, kind="port", seconds_per_clip=med, runs_s=[round(t, 2) for t in ts],
                sample=f"the whole T={T} {H}x{W} clip, iters={iters}: 1 warm-up + {runs} runs of the fp32 torch-CPU oracle's cascade, median ({med:.1f} s per clip)")


def launch_ranks(n, argv):
    """One process per GPU on this node through torch.distributed.run (the same command line the driver uses), rendezvous on 127.0.0.1 at a free
    port; rank 0's JSON line goes to this process's stdout.  Returns the launcher's exit code (non-zero if any rank failed)."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")        # dmabuf IPC: RCCL's device-buffer sharing needs it on this host driver
    env.setdefault("OMP_NUM_THREADS", str(max(1, (os.cpu_count() or 8) // n)))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + list(argv)
    return subprocess.run(cmd, env=env).returncode


class ExtraPhaseGuard:
    """Keeps the EXTRA frame-sharded window (N > 1, behind the replica measurement) from costing the replica measurement or hiding a failure.

    The replica line is complete before the extra phase starts; `emit(reason)` prints it (rank 0) with the failure recorded under `sharded`.  A rank
    whose extra phase raises calls fail(): emit + a fresh non-zero exit (os._exit: the process group may be wedged behind the failed exchange, no
    teardown is attempted; never a re-exec).  The OTHER ranks are then blocked inside a collective, in C, where no Python signal handler runs: a
    helper thread waits on the signal wake-up pipe (signal.set_wakeup_fd: written by the C-level handler at once, whatever the main thread is doing)
    and on a deadline -- on the launcher's SIGTERM (torch.distributed.run terminates the remaining ranks when one fails) or at the deadline it emits
    and exits non-zero too.  So: rank 0's line is printed exactly once whichever rank fails, and every rank of a failed run exits non-zero."""

    EXIT_CODE = 4

    def __init__(self, emit, timeout_s: float):
        import select
        import signal
        import threading
        self._emit, self._lock, self._finished = emit, threading.Lock(), False
        self._r, self._w = os.pipe()
        os.set_blocking(self._w, False)
        self._old_handler = signal.signal(signal.SIGTERM, lambda *_: None)          # (main thread) C-level handler -> wake-up pipe
        self._old_fd = signal.set_wakeup_fd(self._w, warn_on_full_buffer=False)

        def watch():
            ready, _, _ = select.select([self._r], [], [], timeout_s)
            if self._finished:
                return
            self.fail("the launcher's SIGTERM arrived during the extra frame-sharded window (another rank failed)" if ready
                      else f"the extra frame-sharded window did not complete within {timeout_s:.0f} s")

        self._thread = threading.Thread(target=watch, daemon=True)
        self._thread.start()

    def fail(self, reason: str):
        with self._lock:                         # first caller wins (main thread's exception vs the watcher)
            if self._finished:
                return
            self._finished = True
            try:
                self._emit(reason)
                sys.stdout.flush(), sys.stderr.flush()
            finally:
                os._exit(self.EXIT_CODE)

    def finish(self):
        import signal
        with self._lock:
            self._finished = True
        os.write(self._w, b"\0")                 # wake the watcher
        self._thread.join()
        signal.set_wakeup_fd(self._old_fd)
        signal.signal(signal.SIGTERM, self._old_handler)
        os.close(self._r), os.close(self._w)


def dry_run(args, D, rank, world):
    """--dry-run: everything around the measurement (rendezvous, barrier, max-over-ranks, frame-shard planning, the one JSON line from rank 0)
    with a stub in place of the step.  No GPU, no kernels, value = null: a rehearsal of the launch path, never a measurement."""
    T = args.T
    sharded = world > 1 and not args.replicas and T % world == 0 and T // world >= 2
    shard = D.FrameShard(rank, world, T) if sharded else None
    D.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        time.sleep(1e-3 * (1 + rank))                 # ranks differ: max-over-ranks must pick the slowest
    D.barrier()
    elapsed = D.max_over_ranks(time.perf_counter() - t0)
    frames = D.sum_over_ranks(float(shard.f if sharded else T))
    out = dict(metric="disparity-px/s", value=None, unit="disparity-px/s", n_gpus=world, steps=args.steps, warmup=args.warmup, sharded=None,
               ms_per_step=round(1e3 * elapsed / args.steps, 3), higher_is_better=True, scaling="strong" if sharded else "weak",
               dry_run=True, backend=torch.distributed.get_backend() if world > 1 else None, frames_over_ranks=frames,
               config=dict(workload="dry run: stub step, no GPU work", T=T, H=args.H, W=args.W, iters=args.iters,
                           parallelism=(f"frames sharded {T // world}/GPU x{world}" if sharded else f"replicas x{world}")))
    if world > 1 and not sharded and args.sharded_T and args.sharded_T % world == 0 and args.sharded_T // world >= 2:
        # the extra frame-sharded window of the real run (sharded_phase): planned and rehearsed on the stub (one all-reduce per repetition), under the
        # same guard: the replica line above is complete, a failure of the extra phase (--inject-sharded-failure RANK: that rank raises in front of a
        # collective the others have entered) must still print it once, with the error, and end every rank non-zero
        def emit(reason):
            if rank == 0:
                print(json.dumps(dict(out, sharded=dict(error=reason, note="the frame-sharded window behind the replica measurement failed; `value` is unaffected"))), flush=True)
            else:
                print(f"[bench rank {rank}] extra sharded phase: {reason}", file=sys.stderr, flush=True)

        guard = ExtraPhaseGuard(emit, args.sharded_timeout)
        try:
            sh = D.FrameShard(rank, world, args.sharded_T)
            D.barrier()
            t1 = time.perf_counter()
            for _ in range(args.sharded_steps):
                time.sleep(1e-3)
            if args.inject_sharded_failure == rank:
                raise RuntimeError(f"injected failure on rank {rank}")
            D.barrier()
            sh_el = D.max_over_ranks(time.perf_counter() - t1)
            sh_frames = D.sum_over_ranks(float(sh.f))
        except Exception as e:                   # noqa: BLE001
            guard.fail(f"{type(e).__name__}: {e}"[:600])
        guard.finish()
        out["sharded"] = dict(T=args.sharded_T, iters=args.sharded_iters, frames_per_gpu=sh.f, frames_over_ranks=sh_frames, ms_per_window=None,
                              stub_ms=round(1e3 * sh_el / args.sharded_steps, 3), note="dry run: planned, not measured")
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()


SINGLE_GPU_T40_MS = 495.4     # one T=40 window at 320x512, iters=20 on ONE MI355X (profiles/r05_bench_T40_320x512_iters20_single_gpu.json; round 4: 542)


def sharded_phase(args, D, model, rank, world, dev):
    """N > 1: ONE window of --sharded-T frames (BASELINE config 4: T = 40, iters = 20) frame-sharded over the same ranks -- the memory keys gathered once
    per scale, the values + confidences every iteration, +-2 / +-1-frame halos for the temporal convolutions (ppmstereo_amd.dist.FrameShard; reference:
    ppmstereo.py:277-307,524-550, ppmtereo_update.py:281-310,670-678) -- behind the replica measurement.  First a CHECK on the hardware this runs on: the
    sharded cascade at reduced iteration counts against the same window computed unsharded on rank 0; a failed check ends the run with a non-zero
    exit code (3) on every rank, after rank 0 has printed the replica line with the failed check in it.  Then --sharded-steps timed windows (barrier + synchronize on both sides, max over ranks).  Returns (sharded, check)."""
    from ppmstereo_amd.synth import synth_cascade_feats
    T, H, W, iters = args.sharded_T, args.H, args.W, args.sharded_iters
    shard = D.FrameShard(rank, world, T)
    full = synth_cascade_feats(T, H, W)                                           # host tensors: every rank builds the same window
    local = {k: v[shard.lo:shard.hi].to(dev) for k, v in full.items()}
    chk_iters = 2
    d_sh, _ = model.cascade(local, chk_iters, T, shard=shard, test_mode=True)
    torch.cuda.synchronize()
    d_all, _ = shard.all_gather(d_sh.float().contiguous())                        # (T, 1, H, W) on every rank
    err = 0.0
    if rank == 0:
        on_dev = {k: v.to(dev) for k, v in full.items()}
        d_ref, _ = model.cascade(on_dev, chk_iters, T, test_mode=True)
        torch.cuda.synchronize()
        err = float((d_all.to(dev) - d_ref.float()).abs().max().item()) if torch.isfinite(d_all).all() else float("inf")
        del on_dev, d_ref
    del full
    err = D.max_over_ranks(err)
    tol = 1e-3                                                                    # px; measured on two / four ranks of one GPU: <= 5e-5 of the range
    check = dict(max_abs_disparity_diff_px=err, tolerance_px=tol, iters=chk_iters, reference="the same window unsharded on rank 0", passed=bool(err <= tol))
    if not check["passed"]:                  # (agreed by all ranks: err is the max over ranks) -- main prints the replica line with the failed check, exit code 3
        return None, check
    model.cascade(local, iters, T, shard=shard, test_mode=True)                   # warm-up at the real iteration count
    torch.cuda.synchronize()
    D.barrier()
    t0 = time.perf_counter()
    for _ in range(args.sharded_steps):
        d_sh, _ = model.cascade(local, iters, T, shard=shard, test_mode=True)
    torch.cuda.synchronize()
    D.barrier()
    ms = 1e3 * D.max_over_ranks(time.perf_counter() - t0) / args.sharded_steps
    if not torch.isfinite(d_sh).all():
        raise RuntimeError("non-finite disparities in the timed frame-sharded window")
    cfg4 = (T, H, W, iters) == (40, 320, 512, 20)
    out = dict(T=T, H=H, W=W, iters=iters, frames_per_gpu=shard.f, steps=args.sharded_steps, ms_per_window=round(ms, 3), px_per_s=round(T * H * W / (ms * 1e-3), 1),
               scaling="strong", vs_single_gpu_495ms=(round(SINGLE_GPU_T40_MS / ms, 3) if cfg4 else None), vs_single_gpu_542ms=(round(542.0 / ms, 3) if cfg4 else None),
               single_gpu_ms_reference=(SINGLE_GPU_T40_MS if cfg4 else None),
               exchanges_per_iteration="6 (7 at the 1/16 scale): values + confidences (direct all-gather), +-2 frames of x (async), of h, of r*h, +-1 frame of the "
                                       "hidden state and of the flow-head taps; keys + frame descriptors once per scale",
               backend=torch.distributed.get_backend(),
               note="one frame-sharded window behind the replica measurement; never part of `value`"
                    + ("" if torch.distributed.get_backend() == "nccl" else "; NOT RCCL: a rehearsal over " + torch.distributed.get_backend()))
    return out, check


BUILD_MODE = None
TIMING_EVERY = 20         # per-launch events in steps 0, 20, 40, ... of the timed region (one step's launches: 117 convs, 20 attention calls)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=None, help="ranks = GPUs of this node (one process per GPU); default: WORLD_SIZE when a launcher set it, else 1")
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--T", type=int, default=5)
    ap.add_argument("--H", type=int, default=320)
    ap.add_argument("--W", type=int, default=512)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-baseline", choices=("sampled", "full"), default="sampled", help="sampled (default): one iteration per scale, extrapolated by iteration "
                    "count (~10 s); full: SURVEY 8(d)'s procedure, 1 warm-up + 3 whole-clip runs, median (~6 min) -- the sampled figure is printed beside it")
    ap.add_argument("--no-kernel-timing", action="store_true", help="skip the HIP-event brackets around the attention / large-map convolution kernels")
    ap.add_argument("--no-encoders", action="store_true", help="skip the encoder / whole-call timings (fnet + cnet + SST block and "
                    "PPMStereo.forward_batch_test on a host video: once per clip, outside `value`, reported under `encoders` / `whole_call_ms`)")
    ap.add_argument("--with-encoders", action="store_true", help=argparse.SUPPRESS)        # (the default since round 3)
    ap.add_argument("--pipeline", action="store_true", help="overlap consecutive clips (clip k + 1's small scales on a second stream under clip k's 1/4 scale: "
                    "ppmstereo_amd.ppmstereo.ClipPipeline; measured 40.8 vs 41.8 ms per clip -- the 1/4 scale leaves ~16 CUs to the second stream); off by "
                    "default: the headline number is one clip after the other.  latency_ms_per_clip (one clip alone) is reported either way")
    ap.add_argument("--replicas", action="store_true", help="N > 1: force clip replicas even when T divides over the ranks")
    ap.add_argument("--sharded-T", type=int, default=40, help="N > 1 with clip replicas as the measurement (the default T = 5 does not divide over the ranks): "
                    "frames of the ONE extra window that is frame-sharded over the same ranks behind the timed region (BASELINE config 4: T = 40, 5 frames per GPU "
                    "on 8 GPUs) and reported under `sharded` with its correctness check `sharded_check`; 0 = skip that phase")
    ap.add_argument("--sharded-iters", type=int, default=20, help="iterations of that window (config 4: 20)")
    ap.add_argument("--sharded-steps", type=int, default=3, help="timed repetitions of that window")
    ap.add_argument("--sharded-timeout", type=float, default=240.0, help="deadline (s) of that extra phase: past it every rank prints / records the failure and exits non-zero")
    ap.add_argument("--inject-sharded-failure", type=int, default=-1, help=argparse.SUPPRESS)      # --dry-run only: this rank raises inside the extra phase (tests)
    ap.add_argument("--dry-run", action="store_true", help="launch / rendezvous / argument plumbing only: the ranks form the process group, run a stub step "
                    "(no GPU work), take the barrier + max-over-ranks path and rank 0 prints a line with value = null (CPU rehearsal, tests/test_bench_launch.py)")
    args = ap.parse_args()
    if args.gpus is None:
        args.gpus = int(os.environ.get("WORLD_SIZE", "1")) if "RANK" in os.environ else 1

    if args.gpus > 1 and "RANK" not in os.environ:
        # `python bench.py --gpus N` without a launcher: start the N ranks ourselves (one process per GPU) and hand back their exit code.
        # Nothing has touched the GPU in this process (importing torch does not), and it never does: it only waits for the children.
        sys.exit(launch_ranks(args.gpus, sys.argv[1:]))

    from ppmstereo_amd import dist as D
    rank, world, local = D.init_from_env()
    if world != max(1, args.gpus):
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the launcher started WORLD_SIZE={world} ranks")
    if args.dry_run:
        return dry_run(args, D, rank, world)
    assert torch.cuda.is_available(), "bench.py needs an MI355X (the product has no CPU path)"
    torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
    dev = torch.device("cuda", torch.cuda.current_device())

    from ppmstereo_amd import _lib as L
    from ppmstereo_amd import build as _build0
    global BUILD_MODE
    BUILD_MODE = "reused" if _build0._fresh(_build0._digest()) else "compiled"      # before the first L.load(): was the library on disk the committed sources?
    L.load()
    from ppmstereo_amd import weights as Wm
    from ppmstereo_amd.ppmstereo import PPMStereoHotPath
    from ppmstereo_amd.synth import synth_cascade_feats
    T, H, W, iters = args.T, args.H, args.W, args.iters
    sharded = world > 1 and not args.replicas and T % world == 0 and T // world >= 2
    model = PPMStereoHotPath().load_hot_path_weights(Wm.hot_path_weights()).to(dev).eval()
    feats = synth_cascade_feats(T, H, W)
    shard = None
    if sharded:                                    # this rank's contiguous block of frames
        shard = D.FrameShard(rank, world, T)
        feats = {k: v[shard.lo:shard.hi] for k, v in feats.items()}
    feats = {k: v.to(dev) for k, v in feats.items()}

    from ppmstereo_amd.ppmstereo import ClipPipeline
    # consecutive steps are independent clips (as the sliding windows of a video are): their small, latency-bound scales are enqueued on a second
    # stream and run under the previous clip's 1/4 scale.  Same kernels, same results (tests/test_gpu_block.py); not with a frame-sharded window
    # (its exchanges are ordered on one stream)
    pipe = ClipPipeline(dev) if (args.pipeline and not sharded) else None
    if pipe is not None:
        pipe.record_done = True

    def step():
        return model.cascade(feats, iters, T, shard=shard, test_mode=True, pipeline=pipe)

    for _ in range(max(1, args.warmup)):
        disp, _ = step()
    torch.cuda.synchronize()
    assert torch.isfinite(disp).all()
    if pipe is not None:
        pipe.done_events.clear()

    # HIP events around every launch (in the sampled steps: TIMING_EVERY) of the two dominant kernel families (memory attention; the large-map implicit-GEMM
    # convolution kernels conv6_kernel / conv5_kernel -- whichever the engine picked per conv), on the stream each is launched on
    Tl = T // world if sharded else T
    engs = [(model.update_block16.engine(Tl, H // 16, W // 16, dev, shard), iters // 2), (model.update_block08.engine(Tl, H // 8, W // 8, dev, shard), iters // 2),
            (model.update_block04.engine(Tl, H // 4, W // 4, dev, shard), iters)]
    big, family = {}, {}
    if not args.no_kernel_timing:
        for (e, n_it), sc in zip(engs, (16, 8, 4)):
            e.enable_attn_timing(args.steps * n_it)
            for name, op in e.conv_family_ops().items():          # every convolution-family launch of every scale (roofline_3, timed in
                family[(sc, name)] = op                            # ONE extra step behind the timed region: ~900 event pairs cost ~5 ms)
                if getattr(op, "version", 0) in (5, 8):            # the large-map kernels (roofline / roofline_2): inside the timed region
                    op.events = []
                    big[(sc, name)] = op

    step_ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(args.steps)]
    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    from ppmstereo_amd import engine as _engine
    for i, (a, b) in enumerate(step_ev):
        # the per-launch HIP events of the roofline entries are recorded in every TIMING_EVERY-th step of the timed region only (steps 0, 20,
        # ...): ~330 event pairs per clip cost ~2 ms of the clip's 43, and `value` is the time of ALL steps
        _engine.KERNEL_TIMING["on"] = (not args.no_kernel_timing) and i % TIMING_EVERY == 0
        if pipe is not None:      # the steps whose launches carry events run un-overlapped (their durations are then the kernels' own): neither
            pipe.serial = _engine.KERNEL_TIMING["on"] or ((not args.no_kernel_timing) and i > 0 and (i - 1) % TIMING_EVERY == 0)   # the step nor its successor overlaps it
        a.record()
        step()
        b.record()
    _engine.KERNEL_TIMING["on"] = False               # nothing below is event-timed per launch
    torch.cuda.synchronize()
    D.barrier()
    elapsed = D.max_over_ranks(time.perf_counter() - t0)
    step_ms = [a.elapsed_time(b) for a, b in step_ev]
    if pipe is not None and len(pipe.done_events) == args.steps and args.steps > 1:      # pipelined: a step's time = completion to completion
        step_ms = [pipe.done_events[i - 1].elapsed_time(pipe.done_events[i]) for i in range(1, args.steps)]
        pipe.serial, pipe.record_done = False, False
    attn_ms = [e.attn_times_ms() for e, _ in engs] if not args.no_kernel_timing else []
    # the large-map convolution launches of the timed region: read their events NOW and detach the lists, so that the extra steps below (which
    # switch KERNEL_TIMING on again) cannot append a second step's launches to them (round 4 counted `total_ms_per_step` twice that way)
    big_ms = {k: [a.elapsed_time(b) for a, b in op.events] for k, op in big.items()}
    for op in big.values():
        op.events = None
    fam_events = {}
    if family:                                        # one extra step (outside `value`) with events around EVERY convolution-family launch
        for op in family.values():
            op.events = []
        for e, _ in engs:
            e._ev = None                              # (the attention events of the timed region stay as they are)
        _engine.KERNEL_TIMING["on"] = True
        step()
        _engine.KERNEL_TIMING["on"] = False
        torch.cuda.synchronize()
        for k, op in family.items():
            fam_events[k] = [a.elapsed_time(b) for a, b in op.events]
            op.events = None

    hbm_roof = None
    if not args.no_kernel_timing:                     # another extra step: the HBM-bound kernels of the path (SURVEY 8d "report both")
        from ppmstereo_amd import corr as _corr
        for e, _ in engs:
            for op in e.hbm.values():
                op.events = []
        _corr.BUILD_EVENTS = []
        _engine.KERNEL_TIMING["on"] = True
        step()
        _engine.KERNEL_TIMING["on"] = False
        torch.cuda.synchronize()
        kernels = {}
        for (e, _), sc in zip(engs, (16, 8, 4)):
            by = e.hbm_bytes()
            rows = {k: [a.elapsed_time(b) for a, b in op.events] for k, op in e.hbm.items()}
            rows["corr_build"] = [a.elapsed_time(b) for (shp, (a, b)) in _corr.BUILD_EVENTS if shp == (e.T, e.h, e.w)]
            for k, ms in rows.items():
                if ms:
                    avg = sum(ms) / len(ms)
                    gbs = by[k] / (avg * 1e-3) / 1e9
                    kernels.setdefault(k, {})[f"1/{sc}"] = dict(launches=len(ms), avg_us=round(avg * 1e3, 2), algorithmic_mb=round(by[k] / 1e6, 2),
                                                                gb_per_s=round(gbs, 1), frac=round(gbs / HBM_PEAK_GBS, 4))
            for op in e.hbm.values():
                op.events = None
        _corr.BUILD_EVENTS = None
        lead = kernels.get("corr_build", {}).get("1/4")
        tfile = _latest_profile("corr_traffic.json")
        hbm_roof = dict(bound="hbm", kernel="corr_build_line_kernel at the 1/4 scale (the all-pairs correlation pyramid build: fp32-MFMA contraction over 256 channels + 4 pooled "
                                            "levels; one launch per scale and clip), HIP events around every launch of ONE extra step behind the timed region; `kernels` lists the "
                                            "other HBM-bound launches of the path per scale (multi-level lookup, key modulation K' = bf16(K s + PE), convex upsampling) the same way; "
                                            "algorithmic bytes = SURVEY 8d's formulas, every tensor touched once",
                        achieved=None if lead is None else lead["gb_per_s"], peak=HBM_PEAK_GBS, unit="GB/s", frac=None if lead is None else lead["frac"],
                        traffic=(json.load(open(tfile)).get("hbm_bytes_per_launch") if tfile and (T, H, W) == (5, 320, 512) else None),
                        traffic_note="HBM bytes of ONE 1/4-scale corr_build launch, profiles/rNN_corr_traffic.json (FETCH_SIZE x 2 + WRITE_SIZE)",
                        avg_ms=None if lead is None else round(lead["avg_us"] / 1e3, 5), algorithmic_bytes_per_launch=None if lead is None else lead["algorithmic_mb"] * 1e6,
                        kernels=kernels)
    n_sampled = len(range(0, args.steps, TIMING_EVERY))                # steps whose launches carried events
    px = T * H * W
    value = (1 if sharded else world) * args.steps * px / elapsed
    ksel = min(5, T)
    roofs, family_roof, consistency = [], None, {}
    if not args.no_kernel_timing:
        # ---- memory attention: algorithmic FLOPs of one launch = 4 * n * (ksel * n) * 128 * (clips of this rank)
        tot_flop, tot_ms, n_launch, per_scale = 0.0, 0.0, 0, {}
        for (e, n_it), sc, ms in zip(engs, (16, 8, 4), attn_ms):
            fl = 4.0 * e.n * (ksel * e.n) * 128 * Tl
            tot_flop += fl * len(ms)
            tot_ms += sum(ms)
            n_launch += len(ms)
            per_scale[f"1/{sc}"] = dict(launches=len(ms), avg_ms=round(sum(ms) / len(ms), 4), flop_per_launch=fl,
                                        tflops=round(fl / (sum(ms) / len(ms) * 1e-3) / 1e12, 1))
        ach = tot_flop / (tot_ms * 1e-3) / 1e12
        traffic = None            # HBM bytes per 1/4-scale launch from the committed PMC passes (same kernel, same shape), if present
        tfile = _latest_profile("attn_traffic.json")
        if tfile and (T, H, W) == (5, 320, 512):
            traffic = json.load(open(tfile)).get("hbm_bytes_per_launch")
        roofs.append(dict(bound="mfma", kernel="memory attention = one ppms_mem_attn call (attention kernel + combine), every call in every 20th step of the timed region (step 0, 20, ...) "
                                               "(3 scales: 1/16, 1/8, 1/4); algorithmic FLOPs = sum over launches of 4*n*(k*n)*128*T",
                          achieved=round(ach, 2), peak=BF16_DENSE_PEAK_TFLOPS, unit="TFLOP/s", frac=round(ach / BF16_DENSE_PEAK_TFLOPS, 4),
                          traffic=traffic, traffic_note="HBM bytes of ONE 1/4-scale launch, profiles/rNN_attn_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)",
                          launches=n_launch, avg_ms=round(tot_ms / n_launch, 4), total_ms_per_step=round(tot_ms / n_sampled, 3),
                          flop_per_launch=tot_flop / n_launch, per_scale=per_scale))
        # ---- large-map conv kernels: algorithmic FLOPs of a launch = 2 * pixels * couts * cin * taps from its descriptor
        if big:
            torch.cuda.synchronize()
            c_flop = c_ms = c_bound_ms = 0.0
            c_n, per_op = 0, {}
            for (sc, name), op in sorted(big.items()):
                ms = big_ms[(sc, name)]
                if not ms:
                    continue
                c_flop += op.flops() * len(ms)
                c_ms += sum(ms)
                c_n += len(ms)
                # MFMAs per algorithmic product of THIS launch: 3 (hi*hi, lo*hi, hi*lo), less the hi*lo products the kernel leaves out for input
                # channels whose lo plane is known to be zero (ppms_conv.lo_zero_from): its bound is dense bf16 / mfma_per_product, not / 3
                mpp = op.mfma_per_product()
                c_bound_ms += len(ms) * op.flops() * mpp / (BF16_DENSE_PEAK_TFLOPS * 1e12) * 1e3
                per_op[f"1/{sc}:{name}"] = dict(kernel=f"conv{6 if op.version == 8 else op.version}_kernel", launches=len(ms), avg_ms=round(sum(ms) / len(ms), 4), gflop=round(op.flops() / 1e9, 2),
                                                tflops=round(op.flops() / (sum(ms) / len(ms) * 1e-3) / 1e12, 1), mfma_per_product=round(mpp, 4))
            if c_n:
                cach = c_flop / (c_ms * 1e-3) / 1e12
                cpeak = c_flop / (c_bound_ms * 1e-3) / 1e12           # the launch mix's own bound: dense bf16 / (flop-weighted MFMAs per product), >= 833.3
                # every sampled step launches the same list (round 4's line held two steps' launches for one sampled step: 234 = 2 x 117): a
                # diagnostic field, never an abort behind the timed region
                consistency["launches_per_sampled_step_equal"] = all(len(ms) % n_sampled == 0 for ms in big_ms.values())
                # HBM bytes of one launch: only an IN-SITU figure counts (a launch inside a clip, inputs cold: profiles/rNN_conv_traffic_insitu.json, tools/traffic_pmc.sh).
                # The back-to-back probe launches of rounds 2-5 read their inputs from the Infinity Cache (0.69x the algorithmic bytes in round 5: not physical
                # as HBM traffic of the clip), so those files are no longer quoted here
                ctraffic, ctraffic_note = None, "null: no in-situ PMC figure committed (back-to-back probe launches hit the Infinity Cache: 0.69x the algorithmic bytes, profiles/r05_conv_traffic.json -- not evidence)"
                tfile = _latest_profile("conv_traffic_insitu.json")
                if tfile and (T, H, W) == (5, 320, 512):
                    ctraffic = json.load(open(tfile)).get("hbm_bytes_per_launch")
                    ctraffic_note = "HBM bytes per launch of the conv6 family INSIDE a clip (all launches of rocprofv3 --pmc passes over whole clips, summed / launches), " + os.path.basename(tfile)
                names = sorted({f"conv{6 if op.version == 8 else op.version}_kernel" for (sc, name), op in big.items() if big_ms[(sc, name)]}, reverse=True)
                roofs.append(dict(bound="mfma", kernel=" / ".join(names) + " (large-map implicit-GEMM convolutions, bf16x3 split MFMA; per_op names the kernel of every launch), "
                                                       "every launch in every 20th step of the timed region (step 0, 20, ...); "
                                                       "algorithmic FLOPs = sum over launches of 2*pixels*couts*cin*taps; peak = dense bf16 / (MFMAs per product): 3 "
                                                       "(hi*hi + lo*hi + hi*lo), less the hi*lo products skipped for input channels with an all-zero lo plane "
                                                       "(per_op[].mfma_per_product; flop-weighted over the launches: mfma_per_product)",
                                  achieved=round(cach, 2), peak=round(cpeak, 1), unit="TFLOP/s", frac=round(cach / cpeak, 4),
                                  mfma_per_product=round(BF16_DENSE_PEAK_TFLOPS / cpeak, 4), frac_of_three_mfma_bound=round(cach / CONV_BOUND_TFLOPS, 4),
                                  mfma_issued_tflops=round(cach * BF16_DENSE_PEAK_TFLOPS / cpeak, 1),
                                  frac_of_bf16_dense=round(cach / BF16_DENSE_PEAK_TFLOPS, 4), traffic=ctraffic,
                                  traffic_note=ctraffic_note,
                                  launches=c_n, avg_ms=round(c_ms / c_n, 4), total_ms_per_step=round(c_ms / n_sampled, 3),
                                  flop_per_launch=c_flop / c_n, per_op=per_op))
        roofs.sort(key=lambda r: -r["total_ms_per_step"])          # the kernel with the largest share of a step first
        if pipe is None and not sharded:
            # consistency (a reported field, not an abort): the two families run one after the other inside a step.  Launches of the convs' two streams
            # overlap each other -- both event pairs then count the shared interval, so the SUM of event times may legitimately exceed the wall time of
            # the step; a sum far above it would mean double-counted launches (round 4)
            tot = sum(r["total_ms_per_step"] for r in roofs)
            consistency["event_sum_ms"] = round(tot, 3)
            consistency["event_sum_within_step"] = bool(tot <= max(step_ms[0], 1e3 * elapsed / args.steps) * 1.05)
        # ---- the whole convolution family of a step, all three scales: every implicit-GEMM launch (large-map and small-map kernels, the
        # slice-reduce halves, the once-per-scale hoisted shares and q/k projections), the fused per-pixel chains, the depthwise 7x7 -- against
        # the reference's algorithmic conv FLOPs (14.128 MFLOP per pixel and iteration at 1/4 and 1/8, 17.75 at 1/16)
        if family:
            f_ms, f_n, by_scale = 0.0, 0, {}
            for (sc, name), op in family.items():
                ms = fam_events.get((sc, name), [])
                f_ms += sum(ms)
                f_n += len(ms)
                by_scale[f"1/{sc}"] = by_scale.get(f"1/{sc}", 0.0) + sum(ms)
            conv_flop = sum(n_it * Tl * c * e.n for (e, n_it), c in zip(engs, (17.75e6, 14.128e6, 14.128e6)))
            fach = conv_flop / (f_ms * 1e-3) / 1e12
            family_roof = dict(bound="mfma", kernel="every convolution-family launch of a step at the three scales (conv6 / conv5 / conv2 / conv_stream / gemm1 kernels incl. K-slice reduces, "
                                                    "hoisted shares, q/k projection, fused per-pixel chains, depthwise 7x7), event-bracketed in ONE extra step run behind the timed region; algorithmic "
                                                    "FLOPs = the reference's conv FLOPs of the clip (SURVEY 8d: 8.42 TFLOP at config 2), not the FLOPs executed (the inp "
                                                    "hoist removes ~13 %); launches of the two streams overlap, so the event sum is an upper bound of the busy time",
                               achieved=round(fach, 2), peak=round(CONV_BOUND_TFLOPS, 1), unit="TFLOP/s", frac=round(fach / CONV_BOUND_TFLOPS, 4),
                               traffic=None, launches_per_step=f_n, total_ms_per_step=round(f_ms, 3),
                               ms_per_step_by_scale={k: round(v, 3) for k, v in by_scale.items()}, algorithmic_tflop_per_step=round(conv_flop / 1e12, 3))
    # the same clip with test_mode=False (the reference's training-style return): every iteration runs the mask head, the convex
    # upsampling and the full-resolution resize of its prediction.  Reported beside the headline number, never as `value`.
    n_all = 3
    all_ms = lat_ms = None
    if not sharded:                                # one clip at a time, nothing overlapped: the latency of a clip
        torch.cuda.synchronize()
        D.barrier()
        t_lat = time.perf_counter()
        for _ in range(n_all):
            model.cascade(feats, iters, T, shard=shard, test_mode=True)
            torch.cuda.synchronize()
        D.barrier()
        lat_ms = D.max_over_ranks(time.perf_counter() - t_lat) / n_all * 1e3
    if not sharded:                                # (replicas: every rank times its own clip; the frame-sharded mode reports `value` only)
        torch.cuda.synchronize()
        D.barrier()
        t_all = time.perf_counter()
        for _ in range(n_all):
            model.cascade(feats, iters, T, shard=shard, test_mode=False)
        torch.cuda.synchronize()
        D.barrier()
        all_ms = D.max_over_ranks(time.perf_counter() - t_all) / n_all * 1e3
    encoders = None
    if not args.no_encoders and world == 1 and (T, H, W) == (5, 320, 512):      # (N > 1: the ranks time their clips only)
        # SURVEY 8 rows f3-f5 on the same clip geometry: fnet on the 2T images, cnet on the T left images, SST on the 1/16 features
        from ppmstereo_amd.cnet import Feature
        from ppmstereo_amd.encoder import BasicEncoder
        from ppmstereo_amd.sst import SSTBlock
        fnet, cnet, sst = BasicEncoder(256, "instance"), Feature("tiny", 256), SSTBlock()
        fnet.load_state_dict(Wm.fnet_weights()), cnet.load_state_dict(Wm.cnet_weights()), sst.load_state_dict(Wm.sst_weights())
        fnet, cnet, sst = fnet.to(dev).eval(), cnet.to(dev).eval(), sst.to(dev).eval()
        i1, i2 = Wm.hash_uniform((T, 3, H, W), 611).to(dev), Wm.hash_uniform((T, 3, H, W), 612).to(dev)
        f16a, f16b = feats["f1_16"], feats["f2_16"]

        def timed(fn, reps=5):
            fn()
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return round((time.perf_counter() - t1) / reps * 1e3, 3)

        encoders = dict(fnet_ms=timed(lambda: fnet([i1, i2])), cnet_ms=timed(lambda: cnet(i1)), sst_ms=timed(lambda: sst(f16a, f16b, T)),
                        note="once per clip, in front of the timed path; not part of `value`")
        # the whole reference call with HOST buffers on both sides (SURVEY 8d's wall time): PPMStereo.forward_batch_test on a CPU video
        # tensor -> pad, host->device copy, fnet + cnet + SST, the cascade, unpad, device->host copy.  PCIe inclusive; never `value`.
        from ppmstereo_amd.ppmstereo import PPMStereo
        del fnet, cnet, sst
        whole = PPMStereo.shipped()
        whole.load_hot_path_weights(Wm.hot_path_weights())
        whole.fnet.load_state_dict(Wm.fnet_weights(), strict=True), whole.cnet.load_state_dict(Wm.cnet_weights(), strict=True)
        sd = whole.state_dict()
        sd.update(Wm.sst_weights())
        whole.load_state_dict(sd, strict=True)
        whole = whole.to(dev).eval()
        video = Wm.hash_uniform((T, 2, 3, H, W), 613, 0.0, 255.0).round().contiguous()           # host tensor, as the reference's loader hands it over
        def timed_median(fn, reps=7):             # (the call ends with its device->host copy: every repetition is a complete wall time; the median
            fn()                                  #  keeps one slow repetition on a busy host from moving the number)
            ts = []
            for _ in range(reps):
                torch.cuda.synchronize()
                t1 = time.perf_counter()
                fn()
                torch.cuda.synchronize()
                ts.append(time.perf_counter() - t1)
            return round(sorted(ts)[len(ts) // 2] * 1e3, 3)

        call_ms = timed_median(lambda: whole.forward_batch_test({"stereo_video": video}, kernel_size=20, iters=iters))
        encoders["whole_call_ms"] = call_ms
        encoders["whole_call_px_per_s"] = round(T * H * W / (call_ms * 1e-3), 1)
        encoders["whole_call_note"] = ("PPMStereo.forward_batch_test(host video) -> host disparity: H2D + encoders + SST + cascade + D2H, "
                                       "one window; PCIe inclusive, not `value`")
    out = None
    if rank == 0:
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            # the GPU box gives one GPU's share of the host (16 cores); os.cpu_count() reports the whole machine
            cores = min(16, len(os.sched_getaffinity(0))) if hasattr(os, "sched_getaffinity") else min(16, os.cpu_count() or 1)
            cpu = cpu_baseline(T, H, W, iters, cores)
            if args.cpu_baseline == "full":
                sampled = cpu
                cpu = cpu_baseline_full(T, H, W, iters, cores)
                cpu["sampled_seconds_per_clip"] = sampled["seconds_per_clip"]
                cpu["sampled_vs_full"] = round(sampled["seconds_per_clip"] / cpu["seconds_per_clip"], 4)
        label = BASELINE_CONFIGS.get((T, H, W, iters), "custom configuration")
        par = f"frames sharded {T // world}/GPU x{world} (RCCL all-gather of memory K/V + temporal halos)" if sharded else f"replicas x{world}"
        from ppmstereo_amd import build as _build
        from ppmstereo_amd import engine as _eng
        try:
            stamp = open(_build.STAMP).read().strip()
        except OSError:
            stamp = ""
        out = dict(metric="disparity-px/s", value=round(value, 1), unit="disparity-px/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=round(1e3 * elapsed / args.steps, 3), ms_per_step_median=round(statistics.median(step_ms), 3),
                   ms_per_step_min=round(min(step_ms), 3), higher_is_better=True, scaling="strong" if sharded else "weak", vs_baseline=None, dtype="bf16",
                   precision=f"attention: Q K'^T on bf16 MFMA, fp32 softmax, P~ V on {_eng.TUNING['attn_p']} MFMA (V = the bf16-rounded values), fp32 accumulate; "
                             "convs: bf16x3 split MFMA (fp32-accurate); correlation: fp32 MFMA",
                   data="synthetic", frames_per_s=round((1 if sharded else world) * args.steps * T / elapsed, 2),
                   config=dict(workload=f"{label}: T={T} clip at {H}x{W}, iters={iters}, hot path only (3-scale cascade from encoder outputs: "
                                        "corr pyramid build + lookup, QAM pick, pick-and-play memory attention, ConvGRU3D update, heads, convex upsample, "
                                        "test_mode=True as PPMStereo.forward_batch_test calls it: predictions[-1] is the output, so the mask head, the "
                                        "convex upsampling and the full-resolution resize run only where their result is consumed -- the last iteration "
                                        "of each scale; ms_per_step_all_predictions times the same clip with every iteration's prediction produced)",
                                T=T, H=H, W=W, iters=iters, parallelism=par),
                   ms_per_step_all_predictions=None if all_ms is None else round(all_ms, 3),
                   latency_ms_per_clip=None if lat_ms is None else round(lat_ms, 3),
                   pipeline=("consecutive clips overlap: 1/16 + 1/8 scales of clip k + 1 on a second stream under the 1/4 scale of clip k (ClipPipeline); "
                             "latency_ms_per_clip = one clip alone" if pipe is not None else "off: clips strictly one after the other"),
                   roofline=roofs[0] if roofs else None, roofline_2=roofs[1] if len(roofs) > 1 else None,
                   roofline_3=family_roof if (not args.no_kernel_timing and family) else None, roofline_hbm=hbm_roof, roofline_consistency=consistency or None,
                   cpu_baseline=cpu, sharded=None, sharded_check=None,
                   whole_call_ms=None if not encoders else encoders["whole_call_ms"],
                   library=os.path.relpath(L.lib_path(), ROOT),
                   # the library that ran: "reused" = libppms.so.stamp matched the digest of the committed sources + flags when this process loaded it (it IS
                   # those sources, built earlier); "compiled" = this process (or a rank beside it) ran hipcc on them first
                   build_mode=BUILD_MODE, library_stamp=stamp[:12], library_stamp_matches_sources=bool(stamp and stamp == _build._digest()),
                   **({"encoders": encoders} if encoders else {}))
    exit_code = 0
    if world > 1 and not sharded and args.sharded_T and args.sharded_T % world == 0 and args.sharded_T // world >= 2:
        # The replica measurement is complete (rank 0 holds its line).  The extra frame-sharded window runs under a guard: whichever rank fails in it --
        # an exception, a hang, the launcher's SIGTERM -- rank 0 prints the line exactly once with the error under `sharded`, and every rank exits non-zero.
        def emit(reason):
            if rank == 0:
                print(json.dumps(dict(out, sharded=dict(error=reason, note="the frame-sharded window behind the replica measurement failed; `value` is unaffected"))), flush=True)
            else:
                print(f"[bench rank {rank}] extra sharded phase: {reason}", file=sys.stderr, flush=True)

        guard = ExtraPhaseGuard(emit, args.sharded_timeout)
        sharded_out = sharded_chk = None
        try:
            sharded_out, sharded_chk = sharded_phase(args, D, model, rank, world, dev)
        except Exception as e:                     # noqa: BLE001
            guard.fail(f"{type(e).__name__}: {e}"[:600])
        guard.finish()
        if rank == 0:
            out["sharded"], out["sharded_check"] = sharded_out, sharded_chk
        if sharded_chk is not None and not sharded_chk["passed"]:
            exit_code = 3                          # the check failed on the hardware (agreed by all ranks): the line says so, the exit code too
            if rank == 0:
                out["sharded"] = dict(error="sharded_check failed: the frame-sharded window differs from the unsharded one", note="`value` is unaffected")
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        torch.distributed.destroy_process_group()
    if exit_code:
        sys.exit(exit_code)


if __name__ == "__main__":
    main()
