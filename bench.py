"""Benchmark of the PPMStereo hot path on MI355X (contract: see the round prompt / DESIGN.md section 5).

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One "step" = one pass of the hot path over one clip of BASELINE config 2: T=5 frames, 320x512, iters=10
(cascade 1/16 -> 1/8 -> 1/4: 3 correlation pyramid builds + 5/5/10 iterations of lookup, motion encoder,
uncertainty, QAM pick, pick-and-play memory attention, ConvGRU3D, heads, convex upsample).  Inputs (the encoder / SST
outputs the loop consumes) are synthetic and resident in HBM before the timed region.  For N > 1 every rank runs its
own replica of the clip (T=5 does not divide across ranks without changing the result -- DESIGN.md section 6), so
scaling is "weak" and value = N * pixels / max-over-ranks time.  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

BF16_DENSE_PEAK_TFLOPS = 2500.0        # MI355X_MICROARCH.md: ~2.5 PF dense bf16 MFMA


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--T", type=int, default=5)
    ap.add_argument("--H", type=int, default=320)
    ap.add_argument("--W", type=int, default=512)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-timing", action="store_true", help="skip the HIP-event bracket around the attention kernel")
    args = ap.parse_args()

    from ppmstereo_amd import dist as D
    rank, world, local = D.init_from_env()
    assert torch.cuda.is_available(), "bench.py needs an MI355X (the product has no CPU path)"
    torch.cuda.set_device(local % max(1, torch.cuda.device_count()))
    dev = torch.device("cuda", torch.cuda.current_device())

    from ppmstereo_amd import _lib as L
    from ppmstereo_amd import weights as Wm
    from ppmstereo_amd.ppmstereo import PPMStereoHotPath
    from ppmstereo_amd.synth import synth_cascade_feats
    T, H, W, iters = args.T, args.H, args.W, args.iters
    model = PPMStereoHotPath().load_hot_path_weights(Wm.hot_path_weights()).to(dev).eval()
    feats = {k: v.to(dev) for k, v in synth_cascade_feats(T, H, W).items()}

    def step():
        return model.cascade(feats, iters, T)

    for _ in range(max(1, args.warmup)):
        disp, _ = step()
    torch.cuda.synchronize()
    assert torch.isfinite(disp).all()

    # HIP events around EVERY launch of the dominant kernel (memory attention, all three scales), on the stream it runs on
    engs = [(model.update_block16.engine(T, H // 16, W // 16, dev), iters // 2), (model.update_block08.engine(T, H // 8, W // 8, dev), iters // 2),
            (model.update_block04.engine(T, H // 4, W // 4, dev), iters)]
    if not args.no_kernel_timing:
        for e, n_it in engs:
            e.enable_attn_timing(args.steps * n_it)

    D.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    D.barrier()
    elapsed = D.max_over_ranks(time.perf_counter() - t0, dev)

    px = T * H * W
    value = world * args.steps * px / elapsed
    ksel = min(5, T)
    roof = None
    if not args.no_kernel_timing:
        # algorithmic FLOPs of one launch = 4 * n * (ksel * n) * 128 * T (all T clips per launch)
        tot_flop, tot_ms, n_launch, per_scale = 0.0, 0.0, 0, {}
        for (e, n_it), sc in zip(engs, (16, 8, 4)):
            ms = e.attn_times_ms()
            fl = 4.0 * e.n * (ksel * e.n) * 128 * T
            tot_flop += fl * len(ms)
            tot_ms += sum(ms)
            n_launch += len(ms)
            per_scale[f"1/{sc}"] = dict(launches=len(ms), avg_ms=round(sum(ms) / len(ms), 4), flop_per_launch=fl,
                                        tflops=round(fl / (sum(ms) / len(ms) * 1e-3) / 1e12, 1))
        ach = tot_flop / (tot_ms * 1e-3) / 1e12
        traffic = None            # HBM bytes per 1/4-scale launch from the committed PMC passes (same kernel, same shape), if present
        tfile = os.path.join(ROOT, "profiles", "r01_attn_traffic.json")
        if os.path.exists(tfile) and (T, H, W) == (5, 320, 512):
            traffic = json.load(open(tfile)).get("hbm_bytes_per_launch")
        roof = dict(bound="mfma", kernel="memory attention (mem_attn64_kernel + its fix-up pass + attn_combine_kernel = one ppms_mem_attn call), every call of the timed region "
                                         "(3 scales: 1/16, 1/8, 1/4); algorithmic FLOPs = sum over launches of 4*n*(k*n)*128*T",
                    achieved=round(ach, 2), peak=BF16_DENSE_PEAK_TFLOPS, unit="TFLOP/s", frac=round(ach / BF16_DENSE_PEAK_TFLOPS, 4),
                    traffic=traffic, traffic_note="HBM bytes of ONE 1/4-scale launch, profiles/r01_attn_traffic.json (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes)",
                    launches=n_launch, avg_ms=round(tot_ms / n_launch, 4), flop_per_launch=tot_flop / n_launch, per_scale=per_scale)
    if rank == 0:
        cpu = None
        if world == 1 and not args.no_cpu_baseline:
            # the GPU box gives one GPU's share of the host (16 cores); os.cpu_count() reports the whole machine
            cores = min(16, len(os.sched_getaffinity(0))) if hasattr(os, "sched_getaffinity") else min(16, os.cpu_count() or 1)
            cpu = cpu_baseline(T, H, W, iters, cores)
        out = dict(metric="disparity-px/s", value=round(value, 1), unit="disparity-px/s", n_gpus=world, steps=args.steps, warmup=args.warmup,
                   ms_per_step=round(1e3 * elapsed / args.steps, 3), higher_is_better=True, scaling="weak", vs_baseline=None, dtype="bf16", precision="attention: bf16 MFMA, fp32 softmax/accumulate; convs: bf16x3 split MFMA (fp32-accurate); correlation: fp32 MFMA",
                   data="synthetic", frames_per_s=round(world * args.steps * T / elapsed, 2),
                   config=dict(workload=f"BASELINE config 2: T={T} clip at {H}x{W}, iters={iters}, hot path only (3-scale cascade from encoder outputs: "
                                        "corr pyramid build + lookup, QAM pick, pick-and-play memory attention, ConvGRU3D update, heads, convex upsample); "
                                        "one clip replica per GPU", T=T, H=H, W=W, iters=iters, parallelism=f"replicas x{world}"),
                   roofline=roof, cpu_baseline=cpu, library=os.path.relpath(L.lib_path(), ROOT))
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
