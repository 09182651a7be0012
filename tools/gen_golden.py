"""Generate tests/golden/*.npz by running the REFERENCE itself (build container only).

The reference lives read-only at /root/reference and never travels: this script imports it in place
(four import shims), feeds it the procedural weights of
``ppmstereo_amd.weights`` and the synthetic inputs of ``ppmstereo_amd.synth`` and stores only
OUTPUT vectors (inputs and weights are regenerated from their seeds by the tests).

    python tools/gen_golden.py                     # rewrites tests/golden/
    python tools/gen_golden.py it10 --out /tmp/g   # only the iters=10 / iters=20 fixtures, into another directory

Nothing in tests/, bench.py or smoke() reads /root/reference at run time.
"""
import os
import sys
import types

import numpy as np
import torch
import torch.nn as nn

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.dont_write_bytecode = True
REF = "/root/reference"
sys.path.insert(0, REF)

# --- shims for modules the image lacks ------------------------------------------------------
# unfoldNd (un-vendored, un-pinned; call site ppmstereo.py:219, use_convex_3d=True only) is absent from the image, so its
# published algorithm is restated here: UnfoldNd(kernel_size, padding)(x) is torch.nn.Unfold generalised to N spatial
# dims -- for x (N, C, T, H, W) it returns (N, C * kt*kh*kw, T*H*W) with output channel c * (kt*kh*kw) + k, k = (kt, kh, kw)
# row-major, positions row-major, zero padding, stride 1, dilation 1.  The convex_upsample_3d fixtures are pinned to THIS
# restatement (docs/LOG_r01_r05.md section 4).
_unf = types.ModuleType("unfoldNd")


class _UnfoldNd:
    def __init__(self, kernel_size, dilation=1, padding=0, stride=1):
        self.k = tuple(kernel_size)
        self.p = tuple(padding) if isinstance(padding, (tuple, list)) else (padding,) * len(self.k)
        assert dilation == 1 and stride == 1 and len(self.k) == 3

    def __call__(self, x):
        import torch.nn.functional as F_
        N, C, T, H, W = x.shape
        kt, kh, kw = self.k
        pt, ph, pw = self.p
        xp = F_.pad(x, (pw, pw, ph, ph, pt, pt))
        To, Ho, Wo = T + 2 * pt - kt + 1, H + 2 * ph - kh + 1, W + 2 * pw - kw + 1
        cols = [xp[:, :, a:a + To, b:b + Ho, c:c + Wo] for a in range(kt) for b in range(kh) for c in range(kw)]
        return torch.stack(cols, 2).reshape(N, C * kt * kh * kw, To * Ho * Wo)


_unf.UnfoldNd = _UnfoldNd
sys.modules["unfoldNd"] = _unf
_timm, _tm, _tl = types.ModuleType("timm"), types.ModuleType("timm.models"), types.ModuleType("timm.models.layers")
_tl.trunc_normal_ = nn.init.trunc_normal_


class _DropPath(nn.Identity):
    def __init__(self, *a, **k):
        super().__init__()


_tl.DropPath = _DropPath
sys.modules.update({"timm": _timm, "timm.models": _tm, "timm.models.layers": _tl})

from models.core import corr as rcorr                      # noqa: E402
from models.core import ppmtereo_update as rupd            # noqa: E402
from models.core import ppmstereo as rppm                  # noqa: E402
from models.core import extractor as rext                  # noqa: E402
from models.core import attention as ratt                  # noqa: E402
from models.core import convnext as rcnx                    # noqa: E402

# The fixtures must come from the REFERENCE: this repository ships same-named modules (models/core/{corr,ppmstereo,
# ppmtereo_update}.py, the drop-in import path) as portions of the same PEP 420 namespace, and only the sys.path order above
# (REF first) decides which portion a name binds to.  Refuse to generate anything if one of them is the build's.
for _m in (rcorr, rupd, rppm, rext, ratt, rcnx):
    _f = os.path.realpath(_m.__file__)
    assert _f.startswith(REF + os.sep), f"{_m.__name__} was imported from {_f}, not from the reference under {REF}"
assert rppm.PPMStereo.__module__ == "models.core.ppmstereo" and not hasattr(rppm.PPMStereo, "hot"), "PPMStereo is not the reference's"

from ppmstereo_amd import weights as Wm                    # noqa: E402
from ppmstereo_amd.synth import T40_CASES, synth_scale_inputs   # noqa: E402
from ppmstereo_amd.weights import hash_normal, hash_uniform  # noqa: E402

ATTN_LOG = []


def flash_attn_func(q, k, v, dropout_p=0.0, softmax_scale=None, causal=False):
    """math stand-in for flash_attn.flash_attn_func: (B,S,H,D) bf16 -> bf16, fp32 softmax/accumulate."""
    qf, kf, vf = (t.float().permute(0, 2, 1, 3) for t in (q, k, v))
    p = torch.softmax((qf @ kf.transpose(-1, -2)) * softmax_scale, dim=-1)
    out = (p @ vf).permute(0, 2, 1, 3).to(q.dtype)
    ATTN_LOG.append((q, k, v, softmax_scale, out))
    return out


rppm.flash_attn_func = flash_attn_func

torch.manual_seed(0)
torch.set_num_threads(8)
OUT = os.path.join(ROOT, "tests", "golden")
if "--out" in sys.argv:                          # e.g. the CPU test that regenerates fixtures into a temp dir and compares
    OUT = os.path.abspath(sys.argv[sys.argv.index("--out") + 1])
os.makedirs(OUT, exist_ok=True)


MAX_ELEMS = 16384


def save(name, max_elems=MAX_ELEMS, **arrs):
    """Tensors above max_elems are stored as a strided subsample x.flatten()[::step] under "key__s<step>"
    (tests/golden_util.py applies the same rule), keeping every fixture small."""
    path = os.path.join(OUT, name + ".npz")
    out = {}
    for k, v in arrs.items():
        a = v.detach().cpu().numpy() if torch.is_tensor(v) else np.asarray(v)
        if a.size > max_elems:
            step = -(-a.size // max_elems) | 1          # odd stride: does not alias with power-of-two dims
            out[f"{k}__s{step}"] = a.reshape(-1)[::step].copy()
            out[f"{k}__shape"] = np.asarray(a.shape)
        else:
            out[k] = a
    np.savez_compressed(path, **out)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB", {k: tuple(np.shape(v)) for k, v in out.items()})


def bare_model(attention_type=None, use_convex_3d=False):
    """A reference PPMStereo without its encoders (they need timm/ckpt files; out of scope)."""
    m = rppm.PPMStereo.__new__(rppm.PPMStereo)
    nn.Module.__init__(m)
    m.hidden_dim, m.context_dim, m.dim = 128, 128, 256
    m.use_cnet, m.init_flow, m.use_convex_3d = True, False, use_convex_3d
    m.mixed_precision, m.different_update_blocks, m.use_3d_update_block = False, True, True
    m.attention_type, m.num_frames, m.depth = attention_type, 5, 4
    W = Wm.hot_path_weights(use_convex_3d=use_convex_3d)
    for tag, attn in (("update_block16", "self_stereo_temporal_update_time_update_space"),
                      ("update_block08", None), ("update_block04", None)):
        blk = rupd.SequenceUpdateBlock3D(hidden_dim=128, cor_planes=36, mask_size=4, use_convex_3d=use_convex_3d, attention_type=attn)
        blk.load_state_dict(W[tag], strict=True)          # also proves key names/shapes == reference
        setattr(m, tag, blk)
    m.att = nn.ModuleList([rupd.Attention_qk(num_heads=1, dim_head=128) for _ in range(3)])
    for i in range(3):
        m.att[i].load_state_dict(W[f"att.{i}"], strict=True)
    return m.eval()


def it10_fixtures(m):
    """North-star iteration counts (ppmstereo.py:482: 5 / 5 / 10 iterations at iters=10; 10 / 10 / 20 at iters=20, the count of BASELINE
    configs 3-5): the reference's own PPMStereo.forward with stub
    encoders on a T=5, 64x256 clip, EVERY prediction of the cascade kept (test_mode=False returns the stacked list, :795-810), and ten
    iterations of forward_update_block at one scale.  The right features are the left ones shifted along the epipolar line + noise, so
    the correlation volume has real peaks (as ppmstereo_amd.synth)."""
    T, H, W = 5, 64, 256
    fm1 = hash_normal((T, 256, H // 4, W // 4), 171)
    fm2 = 0.8 * torch.roll(fm1, shifts=-3, dims=3) + 0.6 * hash_normal((T, 256, H // 4, W // 4), 172)
    ctx = [hash_normal((T, 256, H // s, W // s), 173 + i) for i, s in enumerate((4, 8, 16))]

    class FNet(nn.Module):
        def forward(self, x):
            return fm1, fm2

    class CNet(nn.Module):
        def forward(self, x):
            return ctx[0], ctx[1], ctx[2]

    m.fnet, m.cnet = FNet(), CNet()
    img = torch.zeros(1, T, 3, H, W)
    # iters = 10 (config 2) and iters = 20 (configs 3-5: 10 / 10 / 20 iterations, 40 predictions) on the SAME inputs; the iters=20 fixtures
    # keep four times the samples (40 predictions share them)
    for iters, cap in ((10, MAX_ELEMS), (20, 4 * MAX_ELEMS)):
        npred = 2 * (iters // 2) + iters
        ATTN_LOG.clear()
        preds, uncs = m.forward(img, img, iters=iters, test_mode=False)          # (npred, 1, T, 1, H, W)
        assert preds.shape[0] == npred and len(ATTN_LOG) == T * npred
        save(f"cascade_it{iters}", max_elems=cap, predictions=preds[:, 0], uncertainties=uncs[:, 0], disparity=preds[-1], uncertainty=uncs[-1],
             n_attn_calls=len(ATTN_LOG))
        T_, h, w = 5, 16, 64
        d = synth_scale_inputs(T_, h, w, seed=1052, with_mhs=True)
        cb = rcorr.CorrBlock1D(d["fmap1"], d["fmap2"])
        preds, uncs = [], []
        fo, net, mhs = m.forward_update_block(None, m.update_block04, cb, d["flow"], d["net"], d["inp"], d["mhs"], m.att[2], preds, uncs, iters, 1, T_)
        save(f"fub04_it{iters}", max_elems=min(cap, 2 * MAX_ELEMS), flow_out=fo, net=net, mhs=mhs, preds=torch.stack(preds), uncs=torch.stack(uncs))


@torch.no_grad()
def main():
    if "it10" in sys.argv[1:]:                   # only the fixtures of round 3 (the full run regenerates them too)
        it10_fixtures(bare_model())
        return
    # ---- G1: CorrBlock1D build + lookup (corr.py:55-104) -------------------------------------
    d = synth_scale_inputs(2, 4, 32, seed=11)
    cb = rcorr.CorrBlock1D(d["fmap1"], d["fmap2"])
    look = cb(d["flow"])
    save("corr_small", **{f"pyr{i}": p for i, p in enumerate(cb.corr_pyramid)}, lookup=look, coords=cb.coords)
    d = synth_scale_inputs(1, 3, 24, seed=12)               # odd pyramid widths: 24,12,6,3,1
    cb = rcorr.CorrBlock1D(d["fmap1"], d["fmap2"])
    save("corr_odd", **{f"pyr{i}": p for i, p in enumerate(cb.corr_pyramid)}, lookup=cb(d["flow"]))
    # big flows: every tap out of range on some pixels
    d = synth_scale_inputs(1, 2, 32, seed=13)
    cb = rcorr.CorrBlock1D(d["fmap1"], d["fmap2"])
    fl = d["flow"] * 20
    save("corr_oob", lookup=cb(fl))

    # ---- G2: temporal PE (ppmtereo_update.py:25-49) -------------------------------------------
    pes = {f"T{T}": rupd.get_temporal_positional_encoding(T, 128, "cpu", is_normalize=True, scale=1.0).reshape(T, 128) for T in (1, 2, 5, 8)}
    save("temporal_pe", **pes)

    # ---- G3: convex upsample (ppmstereo.py:185-197) ---------------------------------------------
    m = bare_model()
    fl, mk = hash_normal((3, 2, 6, 10), 31), hash_normal((3, 144, 6, 10), 32)
    save("convex_upsample", out=m.convex_upsample(fl, mk, 4))

    # ---- G4: update-block pieces (ppmtereo_update.py) -------------------------------------------
    blk = m.update_block16
    T, h, w = 5, 8, 32
    d = synth_scale_inputs(T, h, w, seed=41, with_mhs=False)
    corr = hash_normal((T, 36, h, w), 42)
    mf, mhs, val = blk.get_motion_and_value(d["flow"], corr, None, d["inp"])
    mf2, mhs2, val2 = blk.get_motion_and_value(d["flow"], corr, mhs, d["inp"])
    unc = blk.get_uncertainty(torch.cat([d["net"], val], 1))
    mfg = mf + 0.3 * hash_normal((T, 128, h, w), 43)
    net, mask, dflow = blk(d["net"], d["inp"], mf, mfg, t=T)
    x = torch.cat([d["inp"], mf, mfg], 1)
    xt = blk.time_attn(x, T=T)
    xs = blk.space_attn(xt, T=T)
    save("update_block16_pieces", mf=mf, mhs=mhs, value=val, mf2=mf2, mhs2=mhs2, unc=unc, net=net, mask=mask,
         dflow=dflow, time_attn=xt, space_attn=xs)
    blk = m.update_block04
    net4, mask4, dflow4 = blk(d["net"], d["inp"], mf, mfg, t=T)
    n5 = d["net"].reshape(1, T, 128, h, w).permute(0, 2, 1, 3, 4)
    x5 = x.reshape(1, T, 384, h, w).permute(0, 2, 1, 3, 4)
    save("update_block04_pieces", net=net4, mask=mask4, dflow=dflow4, gru=blk.gru(n5, x5), flow_head=blk.flow_head(n5))

    # ---- G5: forward_update_block at the three scales (ppmstereo.py:426-594) -------------------
    for name, tag, ai, T, h, w, iters, isc, mh in (("fub16", "update_block16", 0, 5, 8, 32, 2, 4, False),
                                                  ("fub08", "update_block08", 1, 8, 8, 32, 3, 2, True),
                                                  ("fub04", "update_block04", 2, 5, 16, 64, 2, 1, True),
                                                  ("fub04_T2", "update_block04", 2, 2, 8, 32, 2, 1, True),
                                                  # T = 40 >> top-k (BASELINE configs 4-5): QAM pick / usage counter over 4 iterations
                                                  # (seeds / frame contrast: ppmstereo_amd.synth.T40_CASES -- inputs whose top-5 pick is
                                                  #  well conditioned, smallest 5th-to-6th score gap > 2e-4)
                                                  ("fub04_T40", "update_block04", 2, 40, 8, 32, 3, 1, True),
                                                  ("fub16_T40", "update_block16", 0, 40, 8, 32, 2, 4, False)):
        d = synth_scale_inputs(T, h, w, with_mhs=mh, **T40_CASES.get(name, dict(seed=50 + ai + 10 * T)))
        cb = rcorr.CorrBlock1D(d["fmap1"], d["fmap2"])
        preds, uncs = [], []
        ATTN_LOG.clear()
        fo, net, mhs = m.forward_update_block(None, getattr(m, tag), cb, d["flow"], d["net"], d["inp"], d["mhs"], m.att[ai],
                                              preds, uncs, iters, isc, T)
        q0, k0, v0, sc, o0 = ATTN_LOG[T * (iters - 1) + 1]     # last iteration, clip 1
        save(name, flow_out=fo, net=net, mhs=mhs, preds=torch.stack(preds), uncs=torch.stack(uncs),
             attn_q=q0.float()[0, :, 0], attn_k=k0.float()[0, :, 0], attn_v=v0.float()[0, :, 0], attn_o=o0.float()[0, :, 0],
             attn_scale=sc, n_attn_calls=len(ATTN_LOG))

    # ---- G6: the whole cascade through PPMStereo.forward with stubbed encoders ------------------
    T, H, W = 3, 64, 256
    fm1, fm2 = hash_normal((T, 256, H // 4, W // 4), 71), hash_normal((T, 256, H // 4, W // 4), 72)
    ctx = [hash_normal((T, 256, H // s, W // s), 73 + i) for i, s in enumerate((4, 8, 16))]

    class FNet(nn.Module):
        def forward(self, x):
            return fm1, fm2

    class CNet(nn.Module):
        def forward(self, x):
            return ctx[0], ctx[1], ctx[2]

    m.fnet, m.cnet = FNet(), CNet()
    img = torch.zeros(1, T, 3, H, W)
    ATTN_LOG.clear()
    disp, unc = m.forward(img, img, iters=4, test_mode=True)
    save("cascade", disparity=disp, uncertainty=unc, n_attn_calls=len(ATTN_LOG))


    # ---- G8: forward_batch_test itself (ppmstereo.py:238-320): padder, sliding windows, centre-frame stitching --------
    # The reference hard-codes .cuda() on the window tensors (:261-262,287-288); on this CPU-only box Tensor.cuda is made
    # the identity for the duration of the call.  Encoders are stubs keyed on the frame content: frame f of the video is a
    # constant image of value f, the stub returns hash features of seed f, so every window sees its own frames.
    N, k, iters = 25, 20, 4
    H0, W0 = 60, 250                                     # pads to 64 x 256 (2 px top/bottom, 3 px left/right)
    H, W = 64, 256
    video = torch.arange(N, dtype=torch.float32)[:, None, None, None, None].expand(N, 2, 3, H0, W0).contiguous()

    def frame_ids(img):                                  # img: (BT,3,H,W) normalised 2*(x/255)-1
        return [int(round(float((v + 1.0) * 255.0 / 2.0))) for v in img[:, 0, H // 2, W // 2]]

    class FNetV(nn.Module):
        def forward(self, x):
            ids = frame_ids(x[0])
            return (torch.stack([hash_normal((256, H // 4, W // 4), 2000 + f) for f in ids]),
                    torch.stack([hash_normal((256, H // 4, W // 4), 3000 + f) for f in ids]))

    class CNetV(nn.Module):
        def forward(self, x):
            ids = frame_ids(x)
            return tuple(torch.stack([hash_normal((256, H // s_, W // s_), 4000 + 100 * i + f) for f in ids]) for i, s_ in enumerate((4, 8, 16)))

    m.fnet, m.cnet = FNetV(), CNetV()
    real_cuda = torch.Tensor.cuda
    torch.Tensor.cuda = lambda self, *a, **kw: self
    try:
        out = m.forward_batch_test({"stereo_video": video}, kernel_size=k, iters=iters)
        out_one = m.forward_batch_test({"stereo_video": video[:7]}, kernel_size=k, iters=iters)      # kernel_size > num_ims branch
    finally:
        torch.Tensor.cuda = real_cuda
    save("fbt_N25_k20", disparity=out["disparity"], uncertainties=out["uncertainties"])
    save("fbt_N7_k20", disparity=out_one["disparity"], uncertainties=out_one["uncertainties"])

    # ---- G9: use_convex_3d=True (mask_3d head ppmtereo_update.py:903-908,993-996; convex_upsample_3d ppmstereo.py:199-228) --
    m3 = bare_model(use_convex_3d=True)
    fl, mk = hash_normal((4, 2, 6, 10), 33), hash_normal((4, 432, 6, 10), 34)
    save("convex_upsample_3d", out=m3.convex_upsample_3d(fl, mk, 4, 4), out_T1=m3.convex_upsample_3d(fl[:1], mk[:1], 4, 1))
    T, h, w = 5, 8, 32
    d = synth_scale_inputs(T, h, w, seed=41, with_mhs=False)
    corr = hash_normal((T, 36, h, w), 42)
    blk = m3.update_block04
    mf, mhs, val = blk.get_motion_and_value(d["flow"], corr, None, d["inp"])
    mfg = mf + 0.3 * hash_normal((T, 128, h, w), 43)
    net, mask, dflow = blk(d["net"], d["inp"], mf, mfg, t=T)
    save("update_block04_c3d_pieces", net=net, mask=mask, dflow=dflow)
    for name, tag, ai, T, h, w, iters, isc, mh in (("fub04_c3d", "update_block04", 2, 5, 8, 32, 2, 1, True),
                                                  ("fub16_c3d", "update_block16", 0, 3, 8, 32, 2, 4, False)):
        d = synth_scale_inputs(T, h, w, seed=50 + ai + 10 * T, with_mhs=mh)
        cb = rcorr.CorrBlock1D(d["fmap1"], d["fmap2"])
        preds, uncs = [], []
        fo, net, mhs = m3.forward_update_block(None, getattr(m3, tag), cb, d["flow"], d["net"], d["inp"], d["mhs"], m3.att[ai],
                                               preds, uncs, iters, isc, T)
        save(name, flow_out=fo, net=net, mhs=mhs, preds=torch.stack(preds), uncs=torch.stack(uncs))

    # ---- G10: fnet = BasicEncoder(output_dim=256, norm_fn="instance") (extractor.py:348-423), as PPMStereo builds it (ppmstereo.py:64) ---
    fnet = rext.BasicEncoder(output_dim=256, norm_fn="instance", dropout=0.0)
    fnet.load_state_dict(Wm.fnet_weights(), strict=True)          # also proves key names / shapes / order == reference
    assert list(fnet.state_dict().keys()) == list(Wm.fnet_param_shapes().keys())
    fnet.eval()
    for name, n, hh, ww in (("fnet_small", 2, 64, 96), ("fnet_odd", 1, 40, 72)):
        i1, i2 = hash_uniform((n, 3, hh, ww), 600 + hh), hash_uniform((n, 3, hh, ww), 700 + hh)   # normalised images: [-1, 1]
        f1, f2 = fnet([i1, i2])
        save(name, fmap1=f1, fmap2=f2)

    # ---- G11: the SST block (forward_sst_block, ppmstereo.py:322-395) with the reference's own modules (ctor :139-171) ---------------
    at = "self_stereo_temporal_update_time_update_space"
    ms = bare_model(attention_type=at)
    Ws = Wm.sst_weights()
    ms.time_embed = nn.Parameter(torch.zeros(1, 5, 256))
    ms.time_attn_blocks = nn.ModuleList([rupd.TimeAttnBlock(dim=256, num_heads=8) for _ in range(4)])
    ms.self_attn_blocks = nn.ModuleList([ratt.LocalFeatureTransformer(d_model=256, nhead=8, layer_names=["self"], attention="linear") for _ in range(4)])
    ms.cross_attn_blocks = nn.ModuleList([ratt.LocalFeatureTransformer(d_model=256, nhead=8, layer_names=["cross"], attention="linear") for _ in range(4)])
    sst_keys = [k for k in ms.state_dict().keys() if k.split(".")[0] in ("time_embed", "time_attn_blocks", "self_attn_blocks", "cross_attn_blocks")]
    assert sst_keys == list(Wm.sst_param_shapes().keys()), "SST state_dict order"
    missing, unexpected = ms.load_state_dict(Ws, strict=False)
    assert not unexpected and all(k.split(".")[0] not in ("time_embed", "time_attn_blocks", "self_attn_blocks", "cross_attn_blocks") for k in missing)
    ms.eval()
    for name, T, h, w in (("sst_T5", 5, 8, 12), ("sst_T3", 3, 6, 10)):          # T = 3: time_embed interpolation branch (:347-352)
        a, b = hash_normal((T, 256, h, w), 810 + T), hash_normal((T, 256, h, w), 820 + T)
        o1, o2 = ms.forward_sst_block(a, b, T)
        save(name, f1=o1, f2=o2)

    # ---- G12: cnet = Feature("tiny", 256) (convnext.py:202-264).  Its ctor loads a checkpoint from a path of the authors' machine
    # (:221-222); torch.load is answered with a freshly initialised backbone's own state_dict for the duration of the call, the
    # procedural weights are loaded right after (strict) -----------------------------------------------------------------------------
    real_load = torch.load
    torch.load = lambda *a, **k: {"model": rcnx.convnextv2_tiny().state_dict()}
    try:
        cnet = rcnx.Feature("tiny", 256)
    finally:
        torch.load = real_load
    assert list(cnet.state_dict().keys()) == list(Wm.cnet_param_shapes().keys()), "cnet state_dict order"
    cnet.load_state_dict(Wm.cnet_weights(), strict=True)
    cnet.eval()
    for name, n, hh, ww in (("cnet_small", 2, 64, 96), ("cnet_32", 1, 32, 64)):
        img = hash_uniform((n, 3, hh, ww), 900 + hh)
        c4, c8, c16 = cnet(img)
        save(name, c4=c4, c8=c8, c16=c16)

    # ---- G13: iters = 10 (5 / 5 / 10 iterations), the north-star iteration counts ----------------
    it10_fixtures(bare_model())

    # ---- G7: T == 1 -> NaN known answer (SURVEY.md hazard 1) ------------------------------------
    d = synth_scale_inputs(1, 8, 32, seed=81)
    cb = rcorr.CorrBlock1D(d["fmap1"], d["fmap2"])
    preds, uncs = [], []
    fo, net, mhs = m.forward_update_block(None, m.update_block04, cb, d["flow"], d["net"], d["inp"], d["mhs"], m.att[2], preds, uncs, 1, 1, 1)
    save("fub_T1_nan", all_nan=bool(torch.isnan(fo).all()), any_nan=bool(torch.isnan(fo).any()))


if __name__ == "__main__":
    main()
