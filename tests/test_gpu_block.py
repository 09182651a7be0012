"""GPU parity tests, block level: the update-block stages, the whole forward_update_block loop and the 3-scale
cascade against the CPU oracle and the reference-generated golden vectors; plus size-independent properties at the
full BASELINE configuration (T=5, 320x512, iters=10)."""
import pytest
import torch

from golden_util import Golden
from oracle import ppm_oracle as O
from ppmstereo_amd import weights as Wm
from ppmstereo_amd.synth import T40_CASES, synth_cascade_feats, synth_scale_inputs
from ppmstereo_amd.weights import hash_normal

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
W = Wm.hot_path_weights()


@pytest.fixture(scope="module")
def model():
    assert torch.cuda.is_available(), "these tests need the MI355X (no CPU fallback exists)"
    from ppmstereo_amd.ppmstereo import PPMStereoHotPath
    return PPMStereoHotPath().load_hot_path_weights(W).to(DEV).eval()


def maxdiff(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    assert torch.isfinite(a).all(), "non-finite GPU output"
    return (a - b).abs().max().item()


def g(x):
    return None if x is None else x.to(DEV)


@pytest.mark.parametrize("tag", ["update_block16", "update_block04"])
def test_update_block_methods(model, tag):
    """SequenceUpdateBlock3D.get_motion_and_value / get_uncertainty / forward (reference signatures, NCHW)."""
    blk, Wb = getattr(model, tag), W[tag]
    T, h, w = 5, 8, 32
    d = synth_scale_inputs(T, h, w, seed=41, with_mhs=False)
    corr = hash_normal((T, 36, h, w), 42)
    mf, mhs, val = blk.get_motion_and_value(g(d["flow"]), g(corr), None, g(d["inp"]))
    rmf, rmhs, rval = O.get_motion_and_value(Wb, d["flow"], corr, None, d["inp"])
    assert maxdiff(mf, rmf) < 1e-4 and maxdiff(mhs, rmhs) < 1e-4 and maxdiff(val, rval) < 1e-4
    mf2, mhs2, _ = blk.get_motion_and_value(g(d["flow"]), g(corr), g(rmhs), g(d["inp"]))
    r2 = O.get_motion_and_value(Wb, d["flow"], corr, rmhs, d["inp"])
    assert maxdiff(mf2, r2[0]) < 1e-4 and maxdiff(mhs2, r2[1]) < 1e-4
    unc = blk.get_uncertainty(torch.cat([g(d["net"]), g(rval)], 1))
    assert maxdiff(unc, O.get_uncertainty(Wb, torch.cat([d["net"], rval], 1))) < 2e-5
    mfg = rmf + 0.3 * hash_normal((T, 128, h, w), 43)
    net, mask, dflow = blk(g(d["net"]), g(d["inp"]), g(rmf), g(mfg), t=T)
    rnet, rmask, rdflow = O.update_block_forward(Wb, d["net"], d["inp"], rmf, mfg, T, tag == "update_block16")
    assert maxdiff(net, rnet) < 1e-4 and maxdiff(mask, rmask) < 2e-4 and maxdiff(dflow, rdflow) < 1e-4
    if tag == "update_block16":
        gd = Golden("update_block16_pieces")
        gd.check("mf", mf, 1e-4), gd.check("mhs", mhs, 1e-4), gd.check("value", val, 1e-4), gd.check("unc", unc, 2e-5)
        gd.check("net", net, 1e-4), gd.check("mask", mask, 2e-4), gd.check("dflow", dflow, 1e-4)
    else:
        # the golden of block04 was generated from block16's motion features (tools/gen_golden.py G4)
        m16, _, _ = O.get_motion_and_value(W["update_block16"], d["flow"], corr, None, d["inp"])
        mfg16 = m16 + 0.3 * hash_normal((T, 128, h, w), 43)
        net, mask, dflow = blk(g(d["net"]), g(d["inp"]), g(m16), g(mfg16), t=T)
        gd = Golden("update_block04_pieces")
        gd.check("net", net, 1e-4), gd.check("mask", mask, 2e-4), gd.check("dflow", dflow, 1e-4)


@pytest.mark.parametrize("tag,T,h,w", [("update_block04", 5, 80, 128), ("update_block08", 5, 16, 64), ("update_block04", 2, 8, 32)])
def test_gru_convs_on_the_read_out_equal_the_mfg_form(model, tag, T, h, w):
    """Inside the loop the hoisted blocks feed the GRU convs [h | mf, hid] with weights (W_mf + W_mfg | beta W_mfg) instead of
    [h | mf, mfg = mf + beta hid] (ppmstereo.py:552, ppmtereo_update.py:985-988): hid is the attention's bf16 read-out, its lo plane is all
    zero and the products with it are skipped (ppms_conv.lo_zero_from).  The same sum in another order: SequenceUpdateBlock3D.forward on a
    caller's mfg (the reference's operands) and the engine's update() on hid agree to fp32 rounding, at the 1/4 scale's full size (conv_gemm5)
    and on small maps."""
    blk = getattr(model, tag)
    d = synth_scale_inputs(T, h, w, seed=51, with_mhs=False)
    mf = hash_normal((T, 128, h, w), 52)
    hid = hash_normal((T, 128, h, w), 53).to(torch.bfloat16).float()
    beta = float(W[tag]["aggregator.beta"].reshape(-1)[0])
    net_a, mask_a, dflow_a = blk(g(d["net"]), g(d["inp"]), g(mf), g(mf + beta * hid), t=T)
    e = blk.engine(T, h, w, torch.device(DEV))
    assert e.hid_mode
    e.set_net(g(d["net"])), e.set_inp(g(d["inp"])), e.set_mf(g(mf))
    e.load_nchw(g(hid), e.X.view(256, 128))
    assert (e.X.own()[1, :, 256:] == 0).all()
    e._x_hid = True
    e.update()
    assert maxdiff(e.get_mfg(), mf + beta * hid) < 1e-4          # (mf is stored split, hi + lo: 2^-16 relative)
    for a, b, name in ((net_a, e.get_net(), "net"), (mask_a, e.get_mask(), "mask"), (dflow_a, e.get_dflow(), "dflow")):
        assert maxdiff(a, b) < 2e-5 * max(1.0, a.abs().max().item()), name


FUB = [("fub16", "update_block16", 0, 5, 8, 32, 2, 4, False), ("fub08", "update_block08", 1, 8, 8, 32, 3, 2, True),
       ("fub04", "update_block04", 2, 5, 16, 64, 2, 1, True), ("fub04_T2", "update_block04", 2, 2, 8, 32, 2, 1, True),
       # BASELINE configs 4-5 have T = 40 >> top-k: QAM pick / usage counter over several iterations, temporal_pe(40) inside the
       # kernels, T * ksel * n workspaces (inputs with a well-conditioned pick, ppmstereo_amd.synth.T40_CASES)
       ("fub04_T40", "update_block04", 2, 40, 8, 32, 3, 1, True), ("fub16_T40", "update_block16", 0, 40, 8, 32, 2, 4, False)]


@pytest.mark.parametrize("name,tag,ai,T,h,w,iters,isc,mh", FUB)
def test_forward_update_block(model, name, tag, ai, T, h, w, iters, isc, mh):
    """The loop itself (reference signature) vs oracle and vs the reference's own outputs (golden).
    Tolerances: 1e-3 px is the north-star EPE budget; the loop is compared well inside it."""
    from ppmstereo_amd.corr import CorrBlock1D
    d = synth_scale_inputs(T, h, w, with_mhs=mh, **T40_CASES.get(name, dict(seed=50 + ai + 10 * T)))
    cb = CorrBlock1D(g(d["fmap1"]), g(d["fmap2"]))
    preds, uncs = [], []
    fo, net, mhs = model.forward_update_block(None, getattr(model, tag), cb, g(d["flow"]), g(d["net"]), g(d["inp"]), g(d["mhs"]), model.att[ai],
                                              preds, uncs, iters, isc, T)
    rp, ru = [], []
    rfo, rnet, rmhs = O.forward_update_block(W[tag], W[f"att.{ai}"], O.corr_pyramid(d["fmap1"], d["fmap2"]), d["flow"], d["net"], d["inp"], d["mhs"],
                                             iters, isc, T, tag == "update_block16", rp, ru)
    assert len(preds) == iters and len(uncs) == iters
    assert preds[-1].shape == rp[-1].shape and uncs[-1].shape == ru[-1].shape
    epe = (fo[:, 0].cpu() - rfo[:, 0]).abs().mean().item()
    assert epe < 2e-4, f"mean |disparity diff| {epe}"
    mhs_tol = 1e-3 if name in T40_CASES else 5e-4          # the T = 40 inputs carry a per-frame gain: activations up to 4x larger
    assert maxdiff(fo, rfo) < 1e-3 and maxdiff(net, rnet) < 2e-3 and maxdiff(mhs, rmhs) < mhs_tol
    assert maxdiff(torch.stack(preds), torch.stack(rp)) < 1e-3 * isc and maxdiff(torch.stack(uncs), torch.stack(ru)) < 2e-4
    gd = Golden(name)
    gd.check("flow_out", fo, 1e-3), gd.check("net", net, 2e-3), gd.check("mhs", mhs, mhs_tol)
    gd.check("preds", torch.stack(preds), 1e-3 * isc), gd.check("uncs", torch.stack(uncs), 2e-4)


@pytest.mark.parametrize("tag,ai,isc,mh", [("update_block04", 2, 1, True), ("update_block16", 0, 4, False)])
def test_forward_update_block_batch_of_two(model, tag, ai, isc, mh):
    """b = 2 clips in one forward_update_block call (ppmstereo.py:443-449; training-style batches): per element everything is independent
    except the normaliser of the picked frames' scores, which the reference averages over the batch as well (:533) -- against the oracle,
    which generalises to b > 1 the way the reference does; and the two elements must differ from their b = 1 results exactly through that."""
    from ppmstereo_amd.corr import CorrBlock1D
    b, T, h, w, iters = 2, 3, 8, 32, 2
    ds = [synth_scale_inputs(T, h, w, seed=870 + i, with_mhs=mh, frame_contrast=0.5) for i in range(b)]
    cat = lambda k: None if ds[0][k] is None else torch.cat([d[k] for d in ds])
    d = {k: cat(k) for k in ds[0]}
    preds, uncs, rp, ru = [], [], [], []
    fo, net, mhs = model.forward_update_block(None, getattr(model, tag), CorrBlock1D(g(d["fmap1"]), g(d["fmap2"])), g(d["flow"]), g(d["net"]), g(d["inp"]),
                                              g(d["mhs"]), model.att[ai], preds, uncs, iters, isc, T)
    rfo, rnet, rmhs = O.forward_update_block(W[tag], W[f"att.{ai}"], O.corr_pyramid(d["fmap1"], d["fmap2"]), d["flow"], d["net"], d["inp"], d["mhs"],
                                             iters, isc, T, tag == "update_block16", rp, ru)
    assert fo.shape == rfo.shape and len(preds) == iters and preds[-1].shape == rp[-1].shape
    assert (fo[:, 0].cpu() - rfo[:, 0]).abs().mean().item() < 2e-4 and maxdiff(fo, rfo) < 1e-3
    assert maxdiff(net, rnet) < 2e-3 and maxdiff(mhs, rmhs) < 1e-3 and maxdiff(torch.stack(uncs), torch.stack(ru)) < 2e-4
    # b = 1 on the first clip alone: a different normaliser, hence a (slightly) different result
    p1, u1 = [], []
    d0 = ds[0]
    fo1, _, _ = model.forward_update_block(None, getattr(model, tag), CorrBlock1D(g(d0["fmap1"]), g(d0["fmap2"])), g(d0["flow"]), g(d0["net"]), g(d0["inp"]),
                                           g(d0["mhs"]), model.att[ai], p1, u1, iters, isc, T)
    assert 0 < (fo1 - fo[:T]).abs().max().item() < 0.5


def test_cascade_batch_of_two_vs_oracle(model):
    """b = 2 clips through the whole 3-scale cascade (ppmstereo.py:696-791 with frame index bi * t + ti): the glue between the three
    batched forward_update_block calls, against the oracle's cascade on the same 2 x T frames; every prediction of every scale."""
    b, T, H, Wd, iters = 2, 3, 64, 256, 2
    per = [synth_cascade_feats(T, H, Wd, seed=31 + i) for i in range(b)]
    feats = {k: torch.cat([p_[k] for p_ in per]) for k in per[0]}
    preds, uncs, rp, ru = [], [], [], []
    disp, unc = model.cascade({k: v.to(DEV) for k, v in feats.items()}, iters, T, preds, uncs)
    rdisp, runc = O.cascade(W, feats, iters, T, rp, ru)
    assert disp.shape == rdisp.shape == (b * T, 1, H, Wd) and len(preds) == len(rp) == 4
    epe = (disp.cpu() - rdisp).abs().mean().item()
    print(f"batched cascade vs oracle: EPE {epe:.3e} px, max {maxdiff(disp, rdisp):.3e}")
    assert epe < 3e-4 and maxdiff(disp, rdisp) < 3e-3 and maxdiff(unc, runc) < 5e-4
    for a, r in zip(preds, rp):
        assert (a.cpu() - r).abs().mean().item() < 3e-4
    # the batch elements are coupled (the mean of the picked scores, :533): clip 0 alone gives a slightly different answer
    d1, _ = model.cascade({k: v.to(DEV) for k, v in per[0].items()}, iters, T)
    assert 0 < (d1 - disp[:T]).abs().max().item() < 1.0


def test_cascade_golden(model):
    """Three-scale cascade vs the reference's PPMStereo.forward output (stub encoders) -- the 1e-3 EPE gate."""
    T, H, Wd = 3, 64, 256
    fm1, fm2 = hash_normal((T, 256, H // 4, Wd // 4), 71), hash_normal((T, 256, H // 4, Wd // 4), 72)
    ctx = [hash_normal((T, 256, H // s, Wd // s), 73 + i) for i, s in enumerate((4, 8, 16))]
    feats = O.pre_loop_glue(fm1, fm2, *ctx)
    preds, uncs = [], []
    disp, unc = model.cascade({k: v.to(DEV) for k, v in feats.items()}, 4, T, preds, uncs)
    assert len(preds) == 8
    gd = Golden("cascade")
    k, step = gd.keys["disparity"]
    got = disp[None].float().cpu().numpy().reshape(-1)[::step]
    err = abs(got - gd.raw("disparity"))
    epe, worst = float(err.mean()), float(err.max())
    print(f"cascade vs reference: EPE (mean |d disparity|) = {epe:.3e} px, max = {worst:.3e} px")
    assert epe < 6e-4, f"EPE vs reference {epe} (north-star budget 1e-3)"
    assert worst < 5e-3, f"max disparity diff vs reference {worst}"
    gd.check("uncertainty", unc[None], 1e-3)


def test_convex_3d_variant_vs_reference():
    """use_convex_3d=True (the default of the reference's train.py / test.py): mask_3d head (Conv3d 3x3x3 + 1x1x1 -> 432,
    ppmtereo_update.py:903-908, 993-996) and convex_upsample_3d (ppmstereo.py:199-228) against the reference's outputs
    (tools/gen_golden.py G9; unfoldNd restated there) and the oracle."""
    from ppmstereo_amd.corr import CorrBlock1D
    from ppmstereo_amd.ppmstereo import PPMStereoHotPath, convex_upsample_3d
    W3 = Wm.hot_path_weights(use_convex_3d=True)
    m3 = PPMStereoHotPath(use_convex_3d=True).load_hot_path_weights(W3).to(DEV).eval()
    fl, mk = hash_normal((4, 2, 6, 10), 33), hash_normal((4, 432, 6, 10), 34)
    gd = Golden("convex_upsample_3d")
    gd.check("out", convex_upsample_3d(g(fl), g(mk), 4, 4), 5e-6)
    gd.check("out_T1", m3.convex_upsample_3d(g(fl[:1]), g(mk[:1]), 4, 1), 5e-6)
    T, h, w = 5, 8, 32
    d = synth_scale_inputs(T, h, w, seed=41, with_mhs=False)
    corr = hash_normal((T, 36, h, w), 42)
    mf, _, _ = O.get_motion_and_value(W3["update_block04"], d["flow"], corr, None, d["inp"])
    mfg = mf + 0.3 * hash_normal((T, 128, h, w), 43)
    net, mask, dflow = m3.update_block04(g(d["net"]), g(d["inp"]), g(mf), g(mfg), t=T)
    assert mask.shape[1] == 432
    gd = Golden("update_block04_c3d_pieces")
    gd.check("net", net, 1e-4), gd.check("mask", mask, 2e-4), gd.check("dflow", dflow, 1e-4)
    for name, tag, ai, T, h, w, iters, isc, mh in (("fub04_c3d", "update_block04", 2, 5, 8, 32, 2, 1, True),
                                                  ("fub16_c3d", "update_block16", 0, 3, 8, 32, 2, 4, False)):
        d = synth_scale_inputs(T, h, w, seed=50 + ai + 10 * T, with_mhs=mh)
        preds, uncs, rp, ru = [], [], [], []
        fo, net, mhs = m3.forward_update_block(None, getattr(m3, tag), CorrBlock1D(g(d["fmap1"]), g(d["fmap2"])), g(d["flow"]), g(d["net"]),
                                               g(d["inp"]), g(d["mhs"]), m3.att[ai], preds, uncs, iters, isc, T)
        rfo, rnet, rmhs = O.forward_update_block(W3[tag], W3[f"att.{ai}"], O.corr_pyramid(d["fmap1"], d["fmap2"]), d["flow"], d["net"], d["inp"],
                                                 d["mhs"], iters, isc, T, tag == "update_block16", rp, ru)
        assert (fo[:, 0].cpu() - rfo[:, 0]).abs().mean().item() < 2e-4
        assert maxdiff(fo, rfo) < 1e-3 and maxdiff(net, rnet) < 2e-3 and maxdiff(mhs, rmhs) < 5e-4
        gd = Golden(name)
        gd.check("flow_out", fo, 1e-3), gd.check("net", net, 2e-3), gd.check("mhs", mhs, 5e-4)
        gd.check("preds", torch.stack(preds), 1e-3 * isc), gd.check("uncs", torch.stack(uncs), 2e-4)


def test_forward_batch_test_vs_reference():
    """PPMStereo.forward_batch_test (padder, sliding windows, centre-frame stitching, H->D / D->H per window) against the
    REFERENCE's own forward_batch_test output (tests/golden/fbt_*.npz, tools/gen_golden.py G8) with stub encoders: 25 frames
    of 60x250 -> padded 64x256, kernel_size 20 -> windows [0,20) [10,25), kept frames 0-14 / 15-24; and the single-window
    branch (7 frames < kernel_size)."""
    from ppmstereo_amd.ppmstereo import PPMStereo
    from stub_encoders import StubCNet, StubFNet, frame_video
    m = PPMStereo.shipped(fnet=StubFNet(), cnet=StubCNet(), sst=None).load_hot_path_weights(W).to(DEV).eval()      # (the G8 fixtures: attention_type None at model level)
    for name, N in (("fbt_N25_k20", 25), ("fbt_N7_k20", 7)):
        out = m.forward_batch_test({"stereo_video": frame_video(N, 60, 250)}, kernel_size=20, iters=4)
        assert tuple(out["disparity"].shape) == (N, 1, 60, 250) and not out["disparity"].is_cuda
        gd = Golden(name)
        k, step = gd.keys["disparity"]
        err = abs(out["disparity"].float().numpy().reshape(-1)[::step] - gd.raw("disparity"))
        print(f"{name}: EPE vs reference {err.mean():.3e} px, max {err.max():.3e} px")
        assert err.mean() < 6e-4 and err.max() < 5e-3, (name, err.mean(), err.max())
        gd.check("uncertainties", out["uncertainties"], 1e-3)
    # forward() without test_mode returns every prediction of the cascade (ppmstereo.py:795-810)
    v = frame_video(3, 64, 256).to(DEV)
    preds, uncs = m.forward(v[None, :, 0], v[None, :, 1], iters=4, test_mode=False)
    last, _ = m.forward(v[None, :, 0], v[None, :, 1], iters=4, test_mode=True)
    assert tuple(preds.shape) == (8, 1, 3, 1, 64, 256) and torch.equal(preds[-1], last)


def test_forward_with_hip_encoders_and_sst_vs_oracle():
    """Rows f3 + f4 + f5 in place: PPMStereo.forward from the IMAGES with this package's fnet (encoder.py), cnet (cnet.py) and SST block
    (sst.py) -- the whole model, nothing stubbed -- against the oracle's forward with its own BasicEncoder / Feature / forward_sst_block
    restatements (all pinned to the reference by the fnet_* / cnet_* / sst_* fixtures).  Also: the model's state_dict IS the
    reference's: same keys in the same order for every sub-module.
    Tolerance: north_star's 1e-3 px, asserted FROM THE IMAGES since round 4.  Rounds 2-3 could only keep a sanity bound here (1.6e-3 px
    measured): the procedural SST weights had unit LayerNorm gains on the eight residual LoFTR layers, which took O(1) features to
    rms 3.3 and the 1/16 correlation volume (quadratic in them) to values of several hundred -- a bf16 ulp of the attention operands
    was 0.15 there.  With those gains at a quarter (ppmstereo_amd.weights._gen; the sst_* fixtures regenerated from the reference with
    the same weights) the block returns features at the magnitude it received and the same code measures 5.8e-4 px."""
    from ppmstereo_amd.ppmstereo import PPMStereo
    from stub_encoders import StubCNet
    m = PPMStereo.shipped()
    keys = list(m.state_dict().keys())
    assert keys[0] == "time_embed" and not any(k.startswith(("sst.", "_sst")) for k in keys)
    expect = (["time_embed"] + ["fnet." + k for k in Wm.fnet_param_shapes()] + ["cnet." + k for k in Wm.cnet_param_shapes()] +
              ["att.%d.%s" % (i, k) for i in range(3) for k in Wm.att_param_shapes()])
    assert keys[:len(expect)] == expect                     # ctor order of the reference: fnet, cnet, att, update blocks, SST modules
    assert [k for k in keys if k.split(".")[0].endswith("attn_blocks")] == list(Wm.sst_param_shapes().keys())[1:]
    m.load_hot_path_weights(W)
    m.fnet.load_state_dict(Wm.fnet_weights(), strict=True)
    m.cnet.load_state_dict(Wm.cnet_weights(), strict=True)
    sd = m.state_dict()
    sd.update(Wm.sst_weights())
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).eval()
    T, H, Wd = 3, 64, 256
    img1 = (torch.sigmoid(hash_normal((1, T, 3, H, Wd), 901)) * 255.0).contiguous()
    img2 = (torch.sigmoid(hash_normal((1, T, 3, H, Wd), 902)) * 255.0).contiguous()
    d, u = m.forward(img1.to(DEV), img2.to(DEV), iters=4, test_mode=True)
    Wf, Ws, Wc = Wm.fnet_weights(), Wm.sst_weights(), Wm.cnet_weights()
    rd, ru = O.forward(W, lambda x: O.basic_encoder(Wf, x), lambda im: O.feature_cnet(Wc, im), img1, img2, 4, sst_fn=lambda a, b: O.sst_block(Ws, a, b, T))
    err = (d.cpu() - rd).abs()
    print(f"forward with HIP fnet + cnet + SST: EPE vs oracle {err.mean().item():.3e} px, max {err.max().item():.3e} px")
    # measured 5.8e-4 / 2.5e-3 (see the decomposition test below: the encoders add nothing measurable to the loop's own 6.8e-4)
    assert tuple(d.shape) == (1, T, 1, H, Wd) and err.mean().item() < 1e-3 and err.max().item() < 1e-2
    assert maxdiff(u, ru) < 1e-2


def test_whole_model_parity_decomposition_on_structured_video():
    """Which stage moves the whole-model result?  SURVEY.md section 8(d)'s structured ``stereo_video`` (left = hash-integer frames, right =
    left shifted by a smooth disparity + 2 % noise: real correlation peaks) through PPMStereo.forward with the encoders swapped ONE at a
    time: all three from the oracle (-> the hot path's own error from identical inputs), then the HIP fnet, the HIP cnet or the HIP SST
    block alone, then all three HIP -- each against the oracle's whole forward (ppmstereo.py:601-682 glue + extractor.py:348-423 etc.)."""
    from ppmstereo_amd.cnet import Feature
    from ppmstereo_amd.encoder import BasicEncoder
    from ppmstereo_amd.ppmstereo import PPMStereo
    from ppmstereo_amd.sst import SSTBlock
    from ppmstereo_amd.synth import stereo_video
    T, H, Wd, iters = 3, 64, 256, 4
    video = stereo_video(T, H, Wd)
    img1, img2 = video[None, :, 0].contiguous(), video[None, :, 1].contiguous()
    Wf, Ws, Wc = Wm.fnet_weights(), Wm.sst_weights(), Wm.cnet_weights()
    torch.set_num_threads(16)
    rd, ru = O.forward(W, lambda x: O.basic_encoder(Wf, x), lambda im: O.feature_cnet(Wc, im), img1, img2, iters, sst_fn=lambda a, b: O.sst_block(Ws, a, b, T))
    dev = torch.device(DEV)
    o_fnet = lambda x: tuple(t.to(dev) for t in O.basic_encoder(Wf, [x[0].cpu(), x[1].cpu()]))
    o_cnet = lambda im: tuple(t.to(dev) for t in O.feature_cnet(Wc, im.cpu()))
    o_sst = lambda a, b, t: tuple(x.to(dev) for x in O.sst_block(Ws, a.cpu(), b.cpu(), t))
    h_fnet = BasicEncoder(256, "instance")
    h_fnet.load_state_dict(Wf, strict=True)
    h_cnet = Feature("tiny", 256)
    h_cnet.load_state_dict(Wc, strict=True)
    h_sst = SSTBlock()
    h_sst.load_state_dict(Ws, strict=True)
    h_fnet, h_cnet, h_sst = h_fnet.to(dev).eval(), h_cnet.to(dev).eval(), h_sst.to(dev).eval()
    epe = {}
    for tag, fn, cn, ss in (("oracle encoders", o_fnet, o_cnet, o_sst), ("HIP fnet", h_fnet, o_cnet, o_sst), ("HIP cnet", o_fnet, h_cnet, o_sst),
                            ("HIP SST", o_fnet, o_cnet, h_sst), ("all HIP", h_fnet, h_cnet, h_sst)):
        m = PPMStereo.shipped(fnet=fn, cnet=cn, sst=ss).load_hot_path_weights(W).to(dev).eval()
        d, u = m.forward(img1.to(dev), img2.to(dev), iters=iters, test_mode=True)
        err = (d.cpu() - rd).abs()
        epe[tag] = err.mean().item()
        print(f"structured video, {tag:16s}: EPE vs oracle {err.mean().item():.3e} px, max {err.max().item():.3e} px (mean |disparity| {rd.abs().mean().item():.2f} px)")
    assert all(v < 1e-3 for v in epe.values()), epe             # north_star's budget, from the images (measured 5.2e-4 .. 7.0e-4)


def test_forward_batch_test_whole_model():
    """The reference's evaluation entry point on the whole HIP model (ppmstereo.py:238-320 with nothing stubbed): 7 frames of 60 x 250
    (padded to 64 x 256 by InputPadder, one window since kernel_size > num_ims), fnet + cnet + SST block + the 3-scale cascade, against the
    oracle's forward_batch_test with its encoder restatements.  Sanity bound (see the conditioning note above); also deterministic."""
    from ppmstereo_amd.ppmstereo import PPMStereo
    m = PPMStereo.shipped()
    m.load_hot_path_weights(W)
    m.fnet.load_state_dict(Wm.fnet_weights(), strict=True)
    m.cnet.load_state_dict(Wm.cnet_weights(), strict=True)
    sd = m.state_dict()
    sd.update(Wm.sst_weights())
    m.load_state_dict(sd, strict=True)
    m = m.to(DEV).eval()
    N, H0, W0 = 7, 60, 250
    video = (torch.sigmoid(hash_normal((N, 2, 3, H0, W0), 951)) * 255.0).contiguous()
    out = m.forward_batch_test({"stereo_video": video}, kernel_size=20, iters=4)
    out2 = m.forward_batch_test({"stereo_video": video}, kernel_size=20, iters=4)
    assert tuple(out["disparity"].shape) == (N, 1, H0, W0) and torch.equal(out["disparity"], out2["disparity"])
    Wf, Ws, Wc = Wm.fnet_weights(), Wm.sst_weights(), Wm.cnet_weights()
    ref = O.forward_batch_test(W, lambda x: O.basic_encoder(Wf, x), lambda im: O.feature_cnet(Wc, im), video, 20, 4,
                               sst_fn=lambda a, b: O.sst_block(Ws, a, b, N))
    err = (out["disparity"] - ref["disparity"]).abs()
    print(f"forward_batch_test, whole model: EPE vs oracle {err.mean().item():.3e} px, max {err.max().item():.3e} px")
    assert torch.isfinite(out["disparity"]).all() and err.mean().item() < 1e-3 and err.max().item() < 1e-2          # measured 7.4e-4 / 3.0e-3 (1.4e-3 / 9.1e-3 with the unconditioned SST weights of rounds 2-3)
    assert (out["uncertainties"] - ref["uncertainties"]).abs().max().item() < 2e-2


def test_T1_gives_nan_like_reference(model):
    from ppmstereo_amd.corr import CorrBlock1D
    d = synth_scale_inputs(1, 8, 32, seed=81)
    with pytest.warns(UserWarning):
        fo, _, _ = model.forward_update_block(None, model.update_block04, CorrBlock1D(g(d["fmap1"]), g(d["fmap2"])), g(d["flow"]), g(d["net"]),
                                              g(d["inp"]), g(d["mhs"]), model.att[2], [], [], 1, 1, 1)
    assert torch.isnan(fo).any() and bool(Golden("fub_T1_nan").raw("any_nan"))


def test_config2_full_size_vs_oracle(model):
    """BASELINE config 2 geometry at FULL size (T=5, 320x512) against the CPU oracle: the cascade with 1 / 1 / 2 iterations at
    the 1/16, 1/8, 1/4 scales (the oracle needs ~15 s for it).  This is the comparison of mem_attn64_kernel at n = 10 240,
    conv6_kernel at 5x80x128 and every launch planner at the sizes the headline number is measured on."""
    T, H, Wd = 5, 320, 512
    feats = synth_cascade_feats(T, H, Wd)
    rp, ru = [], []
    rd, rc = O.cascade(W, feats, 2, T, rp, ru)
    p1, u1 = [], []
    d1, c1 = model.cascade({k: v.to(DEV) for k, v in feats.items()}, 2, T, p1, u1)
    assert len(p1) == len(rp) == 4
    for i, (a, b) in enumerate(zip(p1, rp)):            # every prediction of the cascade, coarse to fine
        err = (a.cpu() - b).abs()
        print(f"prediction {i}: mean |d disparity| {err.mean().item():.3e} px, max {err.max().item():.3e} px")
        assert err.mean().item() < 2e-4 and err.max().item() < 1e-3, (i, err.mean().item(), err.max().item())
    assert maxdiff(c1, rc) < 2e-4


def test_config3_geometry_vs_oracle(model):
    """BASELINE config 3 geometry (720x1280 -> 736x1280): (a) one iteration of the 1/16 scale (46x80, n = 3 680: not a multiple
    of 64, so the tail-masking attention kernel and the conv planners of that map run) against the oracle; (b) the whole
    T=5, iters=20 cascade at full size: finite, bit-reproducible, bounded."""
    from ppmstereo_amd.corr import CorrBlock1D
    T, h, w = 5, 46, 80
    d = synth_scale_inputs(T, h, w, seed=333, with_mhs=False)
    preds, uncs, rp, ru = [], [], [], []
    fo, net, mhs = model.forward_update_block(None, model.update_block16, CorrBlock1D(g(d["fmap1"]), g(d["fmap2"])), g(d["flow"]), g(d["net"]),
                                              g(d["inp"]), None, model.att[0], preds, uncs, 1, 4, T)
    rfo, rnet, rmhs = O.forward_update_block(W["update_block16"], W["att.0"], O.corr_pyramid(d["fmap1"], d["fmap2"]), d["flow"], d["net"], d["inp"],
                                             None, 1, 4, T, True, rp, ru)
    assert (fo[:, 0].cpu() - rfo[:, 0]).abs().mean().item() < 2e-4
    assert maxdiff(fo, rfo) < 1e-3 and maxdiff(net, rnet) < 2e-3 and maxdiff(mhs, rmhs) < 5e-4
    T, H, Wd = 5, 736, 1280
    feats = {k: v.to(DEV) for k, v in synth_cascade_feats(T, H, Wd).items()}
    d1, c1 = model.cascade(feats, 20, T)
    torch.cuda.synchronize()
    d2, c2 = model.cascade(feats, 20, T)
    assert d1.shape == (T, 1, H, Wd) and torch.isfinite(d1).all() and torch.isfinite(c1).all()
    assert torch.equal(d1, d2) and torch.equal(c1, c2)
    assert (c1 > 0).all() and (c1 < 1).all() and d1.abs().max() < 4 * Wd


def test_full_config_properties(model):
    """BASELINE config 2 geometry (T=5, 320x512, iters=10): too big for the CPU oracle in test time, so checked through
    properties: finite outputs, bit-reproducible across runs, disparity update bounded, uncertainty in (0,1)."""
    T, H, Wd = 5, 320, 512
    feats = {k: v.to(DEV) for k, v in synth_cascade_feats(T, H, Wd).items()}
    p1, u1 = [], []
    d1, c1 = model.cascade(feats, 10, T, p1, u1)
    torch.cuda.synchronize()
    p2, u2 = [], []
    d2, c2 = model.cascade(feats, 10, T, p2, u2)
    assert len(p1) == 20 and d1.shape == (T, 1, H, Wd) and c1.shape == (T, 1, H, Wd)
    assert torch.isfinite(d1).all() and torch.isfinite(c1).all()
    assert torch.equal(d1, d2) and torch.equal(c1, c2), "the loop must be deterministic (no atomics, fixed reduction orders)"
    assert (c1 > 0).all() and (c1 < 1).all()
    assert d1.abs().max() < 4 * Wd


def test_cascade_is_bit_stable_over_many_runs(model):
    """25 back-to-back config-2 cascades (two-stream branches of every iteration, the once-per-scale hoist stream overlapping the
    first iteration, 200+ launches per iteration queued far ahead of the GPU) give ONE result: guards against any
    ordering / hazard dependence between concurrent kernels (the packed-fp32 finding of round 1 showed up only this way)."""
    T, H, Wd = 5, 320, 512
    feats = {k: v.to(DEV) for k, v in synth_cascade_feats(T, H, Wd).items()}
    d0, c0 = model.cascade(feats, 10, T, test_mode=True)
    d0, c0 = d0.clone(), c0.clone()
    bad = 0
    for _ in range(25):
        d, c = model.cascade(feats, 10, T, test_mode=True)
        bad += int(not (torch.equal(d, d0) and torch.equal(c, c0)))
    assert bad == 0, f"{bad} of 25 runs differ"


def test_clip_pipeline_gives_the_same_bits(model):
    """ClipPipeline (the small scales of clip k + 1 on a second stream under the 1/4 scale of clip k): six consecutive clips with DIFFERENT
    inputs at config 2's size, pipelined, must equal the same clips run one after the other bit for bit -- the engines' buffers are shared
    between the stages of consecutive clips and guarded by events only where a later clip overwrites what an earlier one still reads."""
    from ppmstereo_amd.ppmstereo import ClipPipeline
    T, H, Wd = 5, 320, 512
    base = {k: v.to(DEV) for k, v in synth_cascade_feats(T, H, Wd).items()}
    clips = [{k: (v * (1.0 + 0.03 * c) if k.startswith("f") else torch.roll(v, shifts=c, dims=0)) for k, v in base.items()} for c in range(6)]
    want = []
    for feats in clips:
        d, c = model.cascade(feats, 10, T, test_mode=True)
        want.append((d.clone(), c.clone()))
    torch.cuda.synchronize()
    for rep in range(2):
        pipe = ClipPipeline(torch.device(DEV))
        got = [model.cascade(feats, 10, T, test_mode=True, pipeline=pipe) for feats in clips]
        pipe.wait()
        torch.cuda.synchronize()
        for i, ((d, c), (wd, wc)) in enumerate(zip(got, want)):
            assert torch.equal(d, wd) and torch.equal(c, wc), f"clip {i} differs under the pipeline (repetition {rep})"
    assert not torch.equal(want[0][0], want[1][0])


def test_test_mode_drops_only_dead_work(model):
    """test_mode=True launches the mask head, the convex upsampling and the full-resolution resize only where their result is consumed
    (the last iteration of each scale); what it returns must be bit-identical to predictions[-1] / uncertainties[-1] of the same clip
    with every iteration's prediction produced (ppmstereo.py:801-804), at BASELINE config 2's size."""
    T, H, Wd = 5, 320, 512
    feats = {k: v.to(DEV) for k, v in synth_cascade_feats(T, H, Wd).items()}
    preds, uncs = [], []
    model.cascade(feats, 10, T, preds, uncs, test_mode=False)
    d, c = model.cascade(feats, 10, T, test_mode=True)
    assert len(preds) == 20 and torch.equal(d, preds[-1]) and torch.equal(c, uncs[-1])


def test_grouped_gru_tails_do_not_change_the_cascade(model):
    """engine.TUNING["conv6_grouped"]: the two (1,1,5) tails of convz1 / convr1 as ONE grouped conv_gemm6 launch (ppms_conv.groups = 2) or as two launches on
    two streams -- the same sums up to fp32 rounding, at config 2's size (where conv_gemm6 serves the 1/4 scale)."""
    from ppmstereo_amd import engine as E
    T, H, Wd = 5, 320, 512
    feats = {k: v.to(DEV) for k, v in synth_cascade_feats(T, H, Wd).items()}
    blocks = (model.update_block16, model.update_block08, model.update_block04)

    def run(**sw):
        old = {k: E.TUNING[k] for k in sw}
        E.TUNING.update(sw)
        for b in blocks:
            b.invalidate()                     # weights are packed (and engines built) under the switch
        try:
            preds, uncs = [], []
            model.cascade(feats, 2, T, preds, uncs)
            eng = model.update_block04.engine(T, H // 4, Wd // 4, DEV)
            return torch.stack(preds).float().cpu(), torch.stack(uncs).float().cpu(), "zr1_2" in eng.op
        finally:
            E.TUNING.update(old)
            for b in blocks:
                b.invalidate()

    grouped = run(conv6_grouped=True)
    two = run(conv6_grouped=False)
    assert grouped[2] and not two[2], "the switch must select the launch plan of the 1/4 scale"
    # (one pass over K in the grouped launch, two K halves added in the K-split single launches: fp32 rounding apart, amplified by two iterations)
    assert (grouped[0] - two[0]).abs().max().item() < 2e-4 and (grouped[1] - two[1]).abs().max().item() < 2e-4


def test_attention_is_a_convex_combination(model):
    """Softmax-weighted aggregation property at full 1/4-scale size (n = 10240, 5 frames): with V == const vector c per
    channel the output must equal bf16(c) whatever Q, K are."""
    from ppmstereo_amd import _lib as L
    T, n = 5, 10240
    qb = hash_normal((T, n, 128), 950).to(torch.bfloat16).to(DEV)
    kb = hash_normal((T, 5, n, 128), 951).to(torch.bfloat16).to(DEV)
    cvec = hash_normal((128,), 952)
    vt = L.vt_image(cvec[None, :, None].expand(T, 128, n), L.ATTN_P_FP16).to(DEV)
    sel = torch.arange(5, dtype=torch.int32)[None].expand(T, 5).contiguous().to(DEV)
    X = L.SPTensor(T * n, 256, DEV)
    beta = torch.tensor([1.0], device=DEV)
    raw = torch.zeros(T, n, 128, dtype=torch.bfloat16, device=DEV)
    ws = torch.empty(int(L.load().ppms_mem_attn_workspace_bytes(T, 5, n)), dtype=torch.uint8, device=DEV)
    L.check(L.load().ppms_mem_attn(qb.data_ptr(), kb.data_ptr(), vt.data_ptr(), sel.data_ptr(), 5, 0.05, beta.data_ptr(), X.view(0, 128), X.view(128, 128),
                                   raw.data_ptr(), T, n, ws.data_ptr(), 0, L.ATTN_P_FP16, L.stream_ptr()))
    want = cvec.to(torch.bfloat16).float()
    got = raw.float().cpu()
    assert (got - want).abs().max() <= 0.01 * want.abs().max(), "softmax weights do not sum to one"
