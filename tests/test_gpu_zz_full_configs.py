"""GPU parity at the north-star iteration counts and at the FULL sizes of the BASELINE configurations (runs last: the file name sorts
behind the other GPU tests; the config-2 oracle comparison costs ~90 s of host time).

* iters = 10 (5 / 5 / 10 iterations, ppmstereo.py:482,708,744,777) against the REFERENCE's own outputs (tests/golden/cascade_it10.npz,
  fub04_it10.npz, tools/gen_golden.py:it10_fixtures) and, at config 2's full size (T=5, 320x512), against the oracle;
* config 4 (one T=40 window at 320x512, iters=20) and config 5 (T=40 at 736x1280, iters=20): properties of the whole run, one 1/16-scale
  iteration of a T=40 window at 46x80 against the oracle, and -- config 5's 1/4 scale, where pyramid level 0 is 3.0 GB, K' 3.0 GB and the
  fp32 attention partials 6.0 GB, i.e. where a 32-bit byte offset would wrap -- every stage of one iteration checked against the oracle
  on the LAST frame / the highest addresses of every buffer.
"""
import numpy as np
import pytest
import torch

from golden_util import Golden
from oracle import ppm_oracle as O
from ppmstereo_amd import weights as Wm
from ppmstereo_amd.synth import synth_cascade_feats, synth_scale_inputs
from ppmstereo_amd.weights import hash_normal

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
W = Wm.hot_path_weights()


@pytest.fixture(scope="module")
def model():
    assert torch.cuda.is_available(), "these tests need the MI355X (no CPU fallback exists)"
    from ppmstereo_amd.ppmstereo import PPMStereoHotPath
    return PPMStereoHotPath().load_hot_path_weights(W).to(DEV).eval()


def g(x):
    return None if x is None else x.to(DEV)


def it10_cascade_inputs():
    T, H, Wd = 5, 64, 256
    fm1 = hash_normal((T, 256, H // 4, Wd // 4), 171)
    fm2 = 0.8 * torch.roll(fm1, shifts=-3, dims=3) + 0.6 * hash_normal((T, 256, H // 4, Wd // 4), 172)
    ctx = [hash_normal((T, 256, H // s, Wd // s), 173 + i) for i, s in enumerate((4, 8, 16))]
    return T, O.pre_loop_glue(fm1, fm2, *ctx)


def test_cascade_iters10_vs_reference(model):
    """The HIP cascade at the north-star iteration counts against the reference's PPMStereo.forward (all 20 predictions)."""
    gd = Golden("cascade_it10")
    T, feats = it10_cascade_inputs()
    preds, uncs = [], []
    disp, unc = model.cascade({k: v.to(DEV) for k, v in feats.items()}, 10, T, preds, uncs)
    assert len(preds) == 20
    P = torch.stack(preds).float().cpu().numpy()
    k, step = gd.keys["predictions"]
    got, ref = P.reshape(-1)[::step], gd.raw("predictions")
    which = np.arange(0, P.size, step) // P[0].size
    for i in range(20):
        e = np.abs(got - ref)[which == i]
        print(f"prediction {i:2d} ({'1/16' if i < 5 else '1/8' if i < 10 else '1/4'}): EPE vs reference {e.mean():.3e} px, max {e.max():.3e} px")
        assert e.mean() < 5e-4, f"prediction {i}: EPE {e.mean()} (north-star budget 1e-3; measured <= 2.5e-4)"
    k, step = gd.keys["disparity"]
    e = np.abs(disp[None].float().cpu().numpy().reshape(-1)[::step] - gd.raw("disparity"))
    print(f"final disparity: EPE vs reference {e.mean():.3e} px, max {e.max():.3e} px")
    assert e.mean() < 5e-4 and e.max() < 1e-2          # north-star budget: 1e-3 px EPE; measured 2.4e-4 with fp16 P~ (4.7e-4 with bf16 P~), profiles/r06_parity_ab.txt
    gd.check("uncertainty", unc[None], 2e-3)


def test_forward_update_block_ten_iterations_vs_reference(model):
    from ppmstereo_amd.corr import CorrBlock1D
    gd = Golden("fub04_it10")
    T, h, w, iters = 5, 16, 64, 10
    d = synth_scale_inputs(T, h, w, seed=1052, with_mhs=True)
    preds, uncs = [], []
    fo, net, mhs = model.forward_update_block(None, model.update_block04, CorrBlock1D(g(d["fmap1"]), g(d["fmap2"])), g(d["flow"]), g(d["net"]),
                                              g(d["inp"]), g(d["mhs"]), model.att[2], preds, uncs, iters, 1, T)
    P = torch.stack(preds).float().cpu().numpy()
    k, step = gd.keys["preds"]
    got, ref = P.reshape(-1)[::step], gd.raw("preds")
    which = np.arange(0, P.size, step) // P[0].size
    for i in range(iters):
        e = np.abs(got - ref)[which == i]
        print(f"iteration {i}: EPE vs reference {e.mean():.3e} px, max {e.max():.3e} px")
        assert e.mean() < 5e-4 and e.max() < 5e-3
    gd.check("flow_out", fo, 5e-3), gd.check("net", net, 5e-3), gd.check("mhs", mhs, 2e-3), gd.check("uncs", torch.stack(uncs), 5e-4)


def test_cascade_and_block_iters20_vs_reference(model):
    """iters = 20 (10 / 10 / 20 iterations: the count of BASELINE configs 3-5, ppmstereo.py:482,708,744,777) against the REFERENCE's own
    PPMStereo.forward(test_mode=False), all 40 predictions (tests/golden/cascade_it20.npz, same inputs as cascade_it10), and twenty
    iterations of forward_update_block (fub04_it20).  The per-prediction table goes to profiles/rNN_parity.log (pytest -s).  The
    oracle itself -- fp32 everywhere but the bf16 attention operands -- is 3.5e-4 px from the reference at prediction 39
    (tests/test_oracle_golden.py): the recurrence amplifies bf16 rounding flips, and the distance grows with the iteration count.
    Asserted: < 1e-3 px on every prediction (north_star's tolerance)."""
    from ppmstereo_amd.corr import CorrBlock1D
    gd = Golden("cascade_it20")
    T, feats = it10_cascade_inputs()
    preds, uncs = [], []
    disp, unc = model.cascade({k: v.to(DEV) for k, v in feats.items()}, 20, T, preds, uncs)
    assert len(preds) == 40
    P = torch.stack(preds).float().cpu().numpy()
    k, step = gd.keys["predictions"]
    got, ref = P.reshape(-1)[::step], gd.raw("predictions")
    which = np.arange(0, P.size, step) // P[0].size
    worst = 0.0
    for i in range(40):
        e = np.abs(got - ref)[which == i]
        worst = max(worst, float(e.mean()))
        print(f"iters=20 prediction {i:2d} ({'1/16' if i < 10 else '1/8' if i < 20 else '1/4'}): EPE vs reference {e.mean():.3e} px, max {e.max():.3e} px")
    k, step = gd.keys["disparity"]
    e = np.abs(disp[None].float().cpu().numpy().reshape(-1)[::step] - gd.raw("disparity"))
    print(f"iters=20 final disparity: EPE vs reference {e.mean():.3e} px, max {e.max():.3e} px; worst prediction EPE {worst:.3e} px")
    # north_star's budget -- 1e-3 px EPE -- on EVERY one of the 40 predictions and on the final disparity.  Round 5 (bf16 P~ in the memory
    # read-out's P~ V product, what flash-attention itself does) ended 1.32e-3 px from the fixture at prediction 39; with fp16 P~ at the same MFMA
    # count (TUNING["attn_p"], include/ppms.h: PPMS_ATTN_P_FP16) the worst prediction is 6.8e-4 px (profiles/r06_parity_ab.txt; the fp32 oracle
    # itself sits at 3.5e-4 there: the recurrence amplifies any rounding difference).
    assert e.mean() < 1e-3 and worst < 1e-3, (e.mean(), worst)
    gd.check("uncertainty", unc[None], 4e-3)
    gd = Golden("fub04_it20")
    T, h, w, iters = 5, 16, 64, 20
    d = synth_scale_inputs(T, h, w, seed=1052, with_mhs=True)
    preds, uncs = [], []
    fo, net, mhs = model.forward_update_block(None, model.update_block04, CorrBlock1D(g(d["fmap1"]), g(d["fmap2"])), g(d["flow"]), g(d["net"]),
                                              g(d["inp"]), g(d["mhs"]), model.att[2], preds, uncs, iters, 1, T)
    P = torch.stack(preds).float().cpu().numpy()
    k, step = gd.keys["preds"]
    got, ref = P.reshape(-1)[::step], gd.raw("preds")
    which = np.arange(0, P.size, step) // P[0].size
    for i in range(iters):
        e = np.abs(got - ref)[which == i]
        print(f"fub04 iters=20 iteration {i:2d}: EPE vs reference {e.mean():.3e} px, max {e.max():.3e} px")
        assert e.mean() < 1e-3
    gd.check("uncs", torch.stack(uncs), 1e-3)


def test_config2_full_iteration_counts_vs_oracle(model):
    """BASELINE config 2 exactly as the metric is quoted: T=5, 320x512, iters=10 (5 / 5 / 10 iterations), the whole cascade against the CPU
    oracle (~90 s of host time), EPE printed after every iteration so that the amplification over the recurrence is on record.
    north_star: "disparity output matches the reference PyTorch path within 1e-3 EPE on identical inputs"."""
    T, H, Wd = 5, 320, 512
    feats = synth_cascade_feats(T, H, Wd)
    p1, u1 = [], []
    d1, c1 = model.cascade({k: v.to(DEV) for k, v in feats.items()}, 10, T, p1, u1)
    torch.cuda.synchronize()
    rp, ru = [], []
    rd, rc = O.cascade(W, feats, 10, T, rp, ru)
    assert len(p1) == len(rp) == 20
    for i, (a, b) in enumerate(zip(p1, rp)):
        err = (a.cpu() - b).abs()
        print(f"prediction {i:2d} ({'1/16' if i < 5 else '1/8' if i < 10 else '1/4'}): EPE vs oracle {err.mean().item():.3e} px, max {err.max().item():.3e} px,"
              f" mean |disparity| {b.abs().mean().item():.2f} px")
        assert err.mean().item() < 1e-3, (i, err.mean().item())
    err = (d1.cpu() - rd).abs()
    print(f"config 2, final disparity: EPE vs oracle {err.mean().item():.3e} px, max {err.max().item():.3e} px")
    assert err.mean().item() < 1e-3, "north-star parity budget (1e-3 px EPE) exceeded at config 2"
    assert (c1.cpu() - rc).abs().max().item() < 5e-3


def _tile_frames(x5: torch.Tensor, T: int) -> torch.Tensor:
    """(5, C, h, w) on the GPU -> (T, C, h, w): frame t = frame t % 5 rolled by 2 (t // 5) columns and scaled by 1 + 0.05 (t // 5)
    (a T = 40 window at 736x1280 is 2.4 GB per tensor: generated on the device from the five hashed frames)."""
    out = []
    for t in range(T):
        k = t // 5
        out.append(torch.roll(x5[t % 5], shifts=2 * k, dims=2) * (1.0 + 0.05 * k))
    return torch.stack(out)


def test_config4_full_size(model):
    """BASELINE config 4's unit of work on one GPU: ONE T=40 window at 320x512, iters=20 (10 / 10 / 20 iterations)."""
    T, H, Wd = 40, 320, 512
    f5 = {k: v.to(DEV) for k, v in synth_cascade_feats(5, H, Wd).items()}
    feats = {k: _tile_frames(v, T) for k, v in f5.items()}
    d1, c1 = model.cascade(feats, 20, T, test_mode=True)
    d1, c1 = d1.clone(), c1.clone()
    d2, c2 = model.cascade(feats, 20, T, test_mode=True)
    assert d1.shape == (T, 1, H, Wd) and torch.isfinite(d1).all() and torch.isfinite(c1).all()
    assert torch.equal(d1, d2) and torch.equal(c1, c2), "the loop must be deterministic"
    assert (c1 > 0).all() and (c1 < 1).all() and d1.abs().max() < 4 * Wd
    # frames are not copies of each other after the loop (temporal coupling at work), and the picks were not degenerate
    assert (d1[0] - d1[5]).abs().mean() > 1e-3


def test_T40_one_sixteenth_scale_iteration_vs_oracle(model):
    """One iteration of update_block16 (time / space attention over T = 40 frames, tail-masking attention kernel: n = 3 680 is not a
    multiple of 64) at config 5's 1/16-scale geometry, 46 x 80, against the oracle."""
    from ppmstereo_amd.corr import CorrBlock1D
    T, h, w = 40, 46, 80
    d = synth_scale_inputs(T, h, w, seed=813, with_mhs=False, frame_contrast=1.0)
    preds, uncs, rp, ru = [], [], [], []
    fo, net, mhs = model.forward_update_block(None, model.update_block16, CorrBlock1D(g(d["fmap1"]), g(d["fmap2"])), g(d["flow"]), g(d["net"]),
                                              g(d["inp"]), None, model.att[0], preds, uncs, 1, 4, T)
    rfo, rnet, rmhs = O.forward_update_block(W["update_block16"], W["att.0"], O.corr_pyramid(d["fmap1"], d["fmap2"]), d["flow"], d["net"], d["inp"],
                                             None, 1, 4, T, True, rp, ru)
    epe = (fo[:, 0].cpu() - rfo[:, 0]).abs().mean().item()
    print(f"T=40, 46x80, one 1/16-scale iteration: EPE vs oracle {epe:.3e} px, max {(fo.cpu() - rfo).abs().max().item():.3e}")
    assert epe < 2e-4 and (fo.cpu() - rfo).abs().max().item() < 2e-3
    assert (net.cpu() - rnet).abs().max().item() < 4e-3 and (mhs.cpu() - rmhs).abs().max().item() < 2e-3


def _config5_quarter_scale_inputs(T, h, w):
    d5 = synth_scale_inputs(5, h, w, seed=59, with_mhs=True, shift=3)
    return {k: _tile_frames(v.to(DEV), T) for k, v in d5.items()}


def test_config5_quarter_scale_iteration_high_addresses_vs_oracle(model):
    """Config 5's 1/4 scale (T=40, 184x320: P = 2 355 200 pixels, pyramid level 0 = 3.0 GB, K' = 3.0 GB, attention partials 6.0 GB):
    ONE iteration, every stage compared with the oracle on the LAST frame -- the highest addresses of every buffer, where 32-bit
    byte offsets would have wrapped.  The per-frame stages (lookup, motion encoder, uncertainty, convex upsampling) are compared on
    the whole last frame; the memory attention on 384 sampled queries of the last clip (all 5 x 58 880 keys); the GRU / flow head /
    mask head (temporal taps reach +-6 frames in all) on the bottom-right corner of the last frame from a crop of the last ten."""
    from ppmstereo_amd.corr import CorrBlock1D
    T, h, w = 40, 184, 320
    n = h * w
    Wb, Wa = W["update_block04"], W["att.2"]
    d = _config5_quarter_scale_inputs(T, h, w)
    eng = model.update_block04.engine(T, h, w, torch.device(DEV))
    with torch.cuda.device(DEV):
        eng.set_inp(d["inp"]), eng.set_net(d["net"]), eng.set_flow(d["flow"]), eng.set_mhs(d["mhs"])
        eng.begin(CorrBlock1D(d["fmap1"], d["fmap2"]).levels, model.att[2].packed(torch.device(DEV)))
        eng.lookup()
        corr_last = eng.store_nchw(eng.CORR.view(0, 36), 36)[-1:].cpu()
        eng.motion_and_value()
        mf, val, mhs = eng.get_mf(), eng.get_value(), eng.get_mhs()
        eng.uncertainty()
        unc_last = eng.get_unc()[-1:].cpu()
        eng.pick()
        eng.attend()
        mfg = eng.get_mfg()
        sel, shat = eng.SEL.cpu(), eng.SHAT.cpu()
        eng.update()
        net_new, mask, dflow, flow_new = eng.get_net(), eng.get_mask(), eng.get_dflow(), eng.get_flow()
        up_last = eng.upsample()[-1:].cpu()
        torch.cuda.synchronize()
    last = lambda k: d[k][-1:].cpu()
    # --- per-frame stages on the whole last frame
    pyr = O.corr_pyramid(last("fmap1"), last("fmap2"))
    rcorr = O.corr_lookup(pyr, last("flow"))
    assert (corr_last - rcorr).abs().max().item() < 3e-4, "corr lookup, last frame"
    rmf, rmhs, rval = O.get_motion_and_value(Wb, last("flow"), rcorr, last("mhs"), last("inp"))
    for name, a, b in (("mf", mf, rmf), ("mhs", mhs, rmhs), ("value", val, rval)):
        e = (a[-1:].cpu() - b).abs().max().item()
        print(f"config 5, 1/4 scale, last frame: {name} max |diff| {e:.3e} (|ref| max {b.abs().max().item():.2f})")
        assert e < 2e-4 * max(1.0, b.abs().max().item()), name
    runc = O.get_uncertainty(Wb, torch.cat([last("net"), rval], 1))
    assert (unc_last - runc).abs().max().item() < 5e-5, "uncertainty, last frame"
    # --- memory attention of the last clip, sampled queries, from the oracle's own q / k and the engine's pick
    clip = T - 1
    J = sorted(int(j) for j in sel[clip].tolist())
    assert len(set(J)) == 5 and all(0 <= j < T for j in J)
    qs = torch.arange(0, n, n // 384)[:384]
    qk_clip = torch.nn.functional.conv2d(last("inp"), Wa["to_qk.weight"])[0]                     # (256, h, w)
    pe = O.temporal_pe(T, 128)
    Q = (qk_clip[:128].reshape(128, n).t()[qs] + pe[clip]).to(torch.bfloat16).float()
    Ks, Vs = [], []
    for slot in range(5):
        j = int(sel[clip, slot])
        kj = torch.nn.functional.conv2d(d["inp"][j:j + 1].cpu(), Wa["to_qk.weight"])[0, 128:].reshape(128, n).t()
        Ks.append((kj * shat[clip, slot] + pe[j]).to(torch.bfloat16).float())
        Vs.append(val[j].cpu().reshape(128, n).t().to(torch.bfloat16).float())
    K, V = torch.cat(Ks), torch.cat(Vs)
    hid = O.flash_attn_math(Q, K, V, O.softmax_scale(128))                                       # (384, 128), bf16-rounded
    want = mf[clip].cpu().reshape(128, n).t()[qs] + Wb["aggregator.beta"] * hid
    got = mfg[clip].cpu().reshape(128, n).t()[qs]
    e = (got - want).abs()
    print(f"config 5, memory attention, clip {clip}, {len(qs)} sampled queries x {K.shape[0]} keys: max |mfg diff| {e.max().item():.3e}, mean {e.mean().item():.3e}")
    assert e.mean().item() < 2e-3 * hid.abs().mean().item() + 1e-5 and e.max().item() < 2 ** -7 * hid.abs().max().item() + 1e-4
    # --- GRU + flow head + mask head on the corner of the volume (crop: frames 30..39, rows 150..183, columns 256..319)
    t0, y0, x0 = 30, 150, 256
    crop = lambda a: a[t0:, :, y0:, x0:].cpu()
    rnet, rmask, rdflow = O.update_block_forward(Wb, crop(d["net"]), crop(d["inp"]), crop(mf), crop(mfg), T - t0, False)
    iy, ix = 10, 16                                    # interior of the crop whose receptive field (+-6 rows, +-13 columns, +-6 frames) lies inside it
    for name, a, b, tol in (("net", net_new, rnet, 2e-4), ("mask", mask, rmask, 5e-4), ("dflow", dflow, rdflow, 2e-4)):
        e = (crop(a)[-1, :, iy:, ix:] - b[-1, :, iy:, ix:]).abs().max().item()
        print(f"config 5, update block, corner of the last frame: {name} max |diff| {e:.3e}")
        assert e < tol, name
    rup = O.convex_upsample(flow_new[-1:].cpu(), mask[-1:].cpu())
    assert (up_last - rup).abs().max().item() < 2e-5, "convex upsampling, last frame"


def test_config5_full_size(model):
    """BASELINE config 5's unit of work on one GPU: ONE T=40 window at 736x1280 (720x1280 padded), iters=20: finite, bit-reproducible,
    uncertainty in (0, 1), disparity bounded."""
    T, H, Wd = 40, 736, 1280
    f5 = {k: v.to(DEV) for k, v in synth_cascade_feats(5, H, Wd).items()}
    feats = {k: _tile_frames(v, T) for k, v in f5.items()}
    del f5
    d1, c1 = model.cascade(feats, 20, T, test_mode=True)
    d1, c1 = d1.clone(), c1.clone()
    d2, c2 = model.cascade(feats, 20, T, test_mode=True)
    assert d1.shape == (T, 1, H, Wd) and torch.isfinite(d1).all() and torch.isfinite(c1).all()
    assert torch.equal(d1, d2) and torch.equal(c1, c2), "the loop must be deterministic"
    assert (c1 > 0).all() and (c1 < 1).all() and d1.abs().max() < 4 * Wd
    for blk in (model.update_block16, model.update_block08, model.update_block04):      # tens of GB of engine buffers: release them
        blk._engines.clear()
    torch.cuda.empty_cache()
