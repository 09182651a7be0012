"""GPU parity tests, op level: every HIP kernel (called through the C ABI) against the CPU oracle / a plain
PyTorch fp32 reference of the same op, on seeded inputs.  Run with -m gpu on an MI355X."""
import ctypes as C
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from golden_util import Golden
from oracle import ppm_oracle as O
from ppmstereo_amd import weights as Wm
from ppmstereo_amd.synth import synth_scale_inputs
from ppmstereo_amd.weights import hash_normal

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def lib():
    assert torch.cuda.is_available(), "these tests need the MI355X (no CPU fallback exists)"
    from ppmstereo_amd import _lib as L
    return L


def maxdiff(a, b):
    a, b = a.detach().float().cpu(), b.detach().float().cpu()
    assert a.shape == b.shape, (a.shape, b.shape)
    assert torch.isfinite(a).all(), "non-finite GPU output"
    return (a - b).abs().max().item()


# ------------------------------------------------------------------------------------------------ correlation
# widths: 32 / 24 / 20 (one 32-pixel block per line; 24, 20: ragged pyramid widths 3, 1 / 5, 2, 1), 64 / 48 (64-pixel blocks), 80 / 128 / 160 /
# 320 / 352 (128-pixel blocks: partial, exact, 2 blocks, 3 blocks with a 64- and a 96-pixel remainder), 18 (W % 4 != 0: the general kernel)
@pytest.mark.parametrize("B,H,W,seed,gold", [(2, 4, 32, 11, "corr_small"), (1, 3, 24, 12, "corr_odd"), (5, 20, 32, 14, None),
                                              (2, 5, 80, 15, None), (1, 2, 320, 16, None), (1, 3, 128, 17, None), (2, 2, 64, 18, None),
                                              (1, 2, 48, 19, None), (1, 3, 20, 20, None), (1, 2, 18, 21, None), (1, 9, 160, 22, None),
                                              (1, 1, 352, 23, None)])
def test_corr_build_and_lookup(lib, B, H, W, seed, gold):
    from ppmstereo_amd.corr import CorrBlock1D
    d = synth_scale_inputs(B, H, W, seed=seed)
    cb = CorrBlock1D(d["fmap1"].to(DEV), d["fmap2"].to(DEV))
    pyr = O.corr_pyramid(d["fmap1"], d["fmap2"])
    assert len(cb.corr_pyramid) == 5
    for i in range(5):
        assert cb.corr_pyramid[i].shape == pyr[i].shape
        assert maxdiff(cb.corr_pyramid[i], pyr[i]) < 3e-5, f"level {i}"
    assert maxdiff(cb.coords, O.coords_grid(B, H, W)) == 0
    out = cb(d["flow"].to(DEV))
    assert out.shape == (B, 36, H, W) and out.is_contiguous()
    assert maxdiff(out, O.corr_lookup(pyr, d["flow"])) < 5e-5
    big = cb((d["flow"] * 20).to(DEV))                      # many taps out of range -> zeros
    assert maxdiff(big, O.corr_lookup(pyr, d["flow"] * 20)) < 5e-5
    if gold:
        g = Golden(gold)
        g.check("lookup", out, 5e-5)
        for i in range(5):
            g.check(f"pyr{i}", cb.corr_pyramid[i], 3e-5)
    assert maxdiff(CorrBlock1D.corr(d["fmap1"].to(DEV), d["fmap2"].to(DEV)), O.corr_volume(d["fmap1"], d["fmap2"])[:, :, :, None]) < 3e-5


@pytest.mark.parametrize("C", [16, 256])
def test_corr_build_line_kernel_equals_the_general_kernel_bit_for_bit(lib, C):
    """The line-resident pyramid build (W % 4 == 0, C % 16 == 0, 16-byte aligned operands) against the general kernel, which the library
    falls back to for a 4-byte-misaligned copy of the same features: every level of the pyramid, every width class -- one / two / four
    32-pixel tiles per block side, ragged last tiles and blocks, several x1 and x2 blocks per line, pyramid widths that are not
    multiples of 8 / 16 (levels 3 and 4 drop their incomplete windows)."""
    import ctypes as C_
    L = lib
    for W in (16, 20, 24, 28, 36, 44, 52, 60, 64, 68, 96, 100, 128, 132, 200, 256, 260, 384):
        B, H = 2, 3
        f1, f2 = hash_normal((B, C, H, W), 3100 + W).to(DEV), hash_normal((B, C, H, W), 3200 + W).to(DEV)
        rows = B * H * W
        widths = [W >> l for l in range(5)]

        def build(a, b):
            store = torch.full((rows * sum(widths),), float("nan"), device=DEV)
            lv, off = [], 0
            for wl in widths:
                lv.append(store[off:off + rows * wl])
                off += rows * wl
            ptrs = (C_.c_void_p * 5)(*[t.data_ptr() for t in lv])
            L.check(L.load().ppms_corr_build(a.data_ptr(), b.data_ptr(), ptrs, B, C, H, W, L.stream_ptr()))
            torch.cuda.synchronize()
            return store

        fast = build(f1, f2)
        p1, p2 = torch.empty(f1.numel() + 1, device=DEV), torch.empty(f2.numel() + 1, device=DEV)
        m1, m2 = p1[1:].view_as(f1), p2[1:].view_as(f2)
        m1.copy_(f1), m2.copy_(f2)
        assert m1.data_ptr() % 16 == 4
        general = build(m1, m2)
        assert torch.isfinite(fast).all() and torch.isfinite(general).all(), W      # every pyramid element written by both
        assert torch.equal(fast, general), f"W={W} C={C}: {int((fast != general).sum())} elements differ"


def test_corr_rejects_narrow_maps(lib):
    from ppmstereo_amd.corr import CorrBlock1D
    with pytest.raises(RuntimeError):
        CorrBlock1D(torch.zeros(1, 256, 4, 8, device=DEV), torch.zeros(1, 256, 4, 8, device=DEV))


def test_ops_refuse_cpu_tensors(lib):
    from ppmstereo_amd.corr import CorrBlock1D
    with pytest.raises(RuntimeError):
        CorrBlock1D(torch.zeros(1, 256, 4, 32), torch.zeros(1, 256, 4, 32))


# ------------------------------------------------------------------------------------------------ conv GEMM
def _run_conv(L, x_list, weight, bias, k3, T, H, W, act=0, kind=0, aux=None, z=None, scale=1.0, seg_pad=None, version=2, wm=0, pre=None, nslice=None, ysweep=False,
              m_pad=None, lo_zero_from=0):
    """x_list: list of (P, C_i) fp32 CPU tensors (channel-last).  Returns (P, Cout) fp32 from the SP output."""
    from ppmstereo_amd.engine import ConvOp, epilogue
    from ppmstereo_amd.packing import pack_conv2, pack_conv4
    tile_px = 0
    if version in (5, 5007, 5008):                                         # conv_gemm5; 5007 / 5008 force the blocks per tile
        tile_px, version = (0 if version == 5 else version - 5000), 5
    from ppmstereo_amd.packing import pack_conv6, pack_gemm1, pack_stream
    pack_conv = pack_conv4 if version == 5 else pack_gemm1 if version == 6 else pack_stream if version == 7 else pack_conv6 if version == 8 else pack_conv2
    P = T * H * W
    segs, keep = [], []
    seg_pad = seg_pad or [((x.shape[1] + 31) // 32) * 32 for x in x_list]
    for x, cp in zip(x_list, seg_pad):
        t = L.SPTensor(P, cp, DEV)
        t.set_f32(x.to(DEV))
        segs.append(t.view())
        keep.append(t)
    wpack = weight
    if (version in (5, 8) or ysweep) and k3[2] == 1 and k3[1] > 1:       # y-swept kernels: pack with kh / kw swapped
        wpack = (weight if weight.dim() == 5 else weight[:, :, None]).transpose(3, 4).contiguous()
    if (version in (5, 8) or ysweep) and k3[2] > 1 and k3[1] > 1:        # 2-D swept: (ky, kx) flattened into the x axis
        w5 = weight if weight.dim() == 5 else weight[:, :, None]
        wpack = w5.reshape(w5.shape[0], w5.shape[1], w5.shape[2], 1, k3[1] * k3[2]).contiguous()
    packed, b, meta = pack_conv(wpack.to(DEV), None if bias is None else bias.to(DEV), [x.shape[1] for x in x_list], seg_pad,
                                **({} if m_pad is None else dict(m_pad=m_pad)))
    cout = weight.shape[0]
    out = L.SPTensor(P, meta["M"], DEV)
    outf = torch.zeros(P, meta["M"], device=DEV)
    e = epilogue(kind=kind, act=act, scale=scale, n_valid=cout, out_sp=out.view(), out_f32=outf, out_f32_ld=meta["M"])
    if aux is not None:
        a = L.SPTensor(P, meta["M"], DEV)
        a.set_f32(aux.to(DEV))
        e.aux_sp = a.view()
        keep.append(a)
    if z is not None:
        zt = torch.zeros(P, meta["M"], device=DEV)
        zt[:, :cout] = z.to(DEV)
        e.aux_f32, e.aux_f32_ld = zt.data_ptr(), meta["M"]
        keep.append(zt)
    if pre is not None:                                      # iteration-invariant share of the pre-activation
        pt = torch.zeros(P, meta["M"], device=DEV)
        pt[:, :cout] = pre.to(DEV)
        e.pre_f32, e.pre_f32_ld = pt.data_ptr(), meta["M"]
        keep.append(pt)
    d = L.Conv()
    for i, s in enumerate(segs):
        d.seg[i] = s
    d.nseg, d.w, d.bias = len(segs), packed.data_ptr(), b.data_ptr()
    d.T, d.H, d.W = T, H, W
    d.kt, d.kh, d.kw = k3
    d.M = d.m_split = meta["M"]
    d.lo_zero_from = lo_zero_from
    d.epi[0] = e
    if version == 5 and nslice is not None and nslice < 0:                 # -1: let the library plan the slices (must find some)
        nslice = int(L.load().ppms_conv_gemm5_slices(C.byref(d)))
        assert nslice >= 2, nslice
    ConvOp(d, keep, version, tile_px if version == 5 else wm, nslice=nslice if nslice is not None else 1, ysweep=ysweep)()
    torch.cuda.synchronize()
    sp = out.to_f32()[:, :cout].cpu()
    assert (sp - outf[:, :cout].cpu()).abs().max() < 2e-5 * (1 + sp.abs().max()), "SP and fp32 outputs of one launch disagree"
    assert (out.to_f32()[:, cout:] == 0).all(), "padded couts must not be written"
    return sp


def _ref_conv(x_list, weight, bias, k3, T, H, W):
    x = torch.cat(x_list, 1)                                                # (P, Cin)
    x5 = x.reshape(1, T, H, W, -1).permute(0, 4, 1, 2, 3)
    w5 = weight if weight.dim() == 5 else weight[:, :, None]
    y = F.conv3d(x5, w5, bias, padding=tuple(k // 2 for k in k3))
    return y.permute(0, 2, 3, 4, 1).reshape(T * H * W, -1)


CONV_CASES = [
    # name, T,H,W, segs, cout, k3
    ("1x1_small", 2, 5, 7, [36], 54, (1, 1, 1)),
    ("3x3_two_segs", 3, 9, 13, [128, 128], 128, (1, 3, 3)),
    ("1x15_gru", 2, 4, 40, [128, 384], 256, (1, 1, 15)),
    ("5x1x1_time", 5, 6, 10, [128, 64], 128, (5, 1, 1)),
    ("1x5x1", 2, 11, 9, [64], 64, (1, 5, 1)),
    ("3x3x3", 4, 7, 9, [128], 190, (3, 3, 3)),
    ("one_tile_exact", 1, 8, 32, [32], 64, (1, 3, 3)),
    ("T1_tiny", 1, 1, 3, [32], 2, (3, 3, 3)),
]


CONV_CASES += [("wide_1x15", 1, 3, 300, [64], 128, (1, 1, 15)), ("w80_3x3", 2, 46, 80, [32], 64, (1, 3, 3)), ("w18_1x5", 2, 10, 18, [64, 32], 192, (1, 1, 5))]


@pytest.mark.parametrize("version,wm", [(2, 0), (2, 1)])
@pytest.mark.parametrize("name,T,H,W,segs,cout,k3", CONV_CASES)
def test_conv_gemm_vs_torch(lib, name, T, H, W, segs, cout, k3, version, wm):
    P = T * H * W
    xs = [hash_normal((P, c), 100 + i) for i, c in enumerate(segs)]
    cin = sum(segs)
    fan = cin * k3[0] * k3[1] * k3[2]
    wt = hash_normal((cout, cin, *k3), 200) / math.sqrt(fan)
    bs = hash_normal((cout,), 201) * 0.1
    ref = _ref_conv(xs, wt, bs, k3, T, H, W)
    got = _run_conv(lib, xs, wt, bs, k3, T, H, W, version=version, wm=wm)
    tol = 3e-5 * max(1.0, ref.abs().max().item())           # bf16x3 split: ~2^-16 relative per product, fp32 accumulate
    assert maxdiff(got, ref) < tol, name


@pytest.mark.parametrize("version", [2])
def test_conv_gemm_epilogues(lib, version):
    import functools
    global _run_conv
    run = functools.partial(_run_conv, version=version)
    T, H, W, cin, cout, k3 = 2, 6, 10, 64, 64, (1, 3, 3)
    P = T * H * W
    x = hash_normal((P, cin), 300)
    wt = hash_normal((cout, cin, *k3), 301) / math.sqrt(cin * 9)
    bs = hash_normal((cout,), 302) * 0.1
    aux = hash_normal((P, cout), 303)
    z = torch.sigmoid(hash_normal((P, cout), 304))
    lin = _ref_conv([x], wt, bs, k3, T, H, W)
    L = lib
    assert maxdiff(run(L, [x], wt, bs, k3, T, H, W, act=L.ACT_RELU), F.relu(lin)) < 5e-5
    assert maxdiff(run(L, [x], wt, bs, k3, T, H, W, act=L.ACT_GELU), F.gelu(lin)) < 5e-5
    assert maxdiff(run(L, [x], wt, bs, k3, T, H, W, act=L.ACT_SIGMOID), torch.sigmoid(lin)) < 5e-5
    assert maxdiff(run(L, [x], wt, bs, k3, T, H, W, act=L.ACT_TANH), torch.tanh(lin)) < 5e-5
    assert maxdiff(run(L, [x], wt, bs, k3, T, H, W, scale=0.25), 0.25 * lin) < 5e-5
    assert maxdiff(run(L, [x], wt, bs, k3, T, H, W, kind=L.EPI_RESID, act=L.ACT_GELU, aux=aux), F.gelu(aux + lin)) < 5e-5
    assert maxdiff(run(L, [x], wt, bs, k3, T, H, W, kind=L.EPI_RH, aux=aux), torch.sigmoid(lin) * aux) < 5e-5
    assert maxdiff(run(L, [x], wt, bs, k3, T, H, W, kind=L.EPI_GRU, aux=aux, z=z), (1 - z) * aux + z * torch.tanh(lin)) < 5e-5


LARGE_MAP_CASES = [          # shapes of the large-map kernels' sweeps (x, y, 2-D, with a temporal extent), ragged tiles included
    ("gru_1x15", 2, 6, 128, [128, 384], 256, (1, 1, 15)),
    ("q_1x5", 1, 5, 128, [128, 64], 128, (1, 1, 5)),
    ("3x3_m128", 3, 20, 32, [128, 128], 128, (1, 3, 3)),
    ("3x3x3_m256", 4, 9, 64, [128], 256, (3, 3, 3)),
    ("y_1x5x1", 2, 40, 32, [128, 32], 256, (1, 5, 1)),
    ("y_w80", 1, 23, 80, [64], 128, (1, 5, 1)),
    ("x_w80_1x5", 2, 7, 80, [64], 128, (1, 1, 5)),
    ("3x3_w80_ragged", 2, 23, 80, [32, 32], 128, (1, 3, 3)),
    ("3x3x3_odd", 3, 11, 50, [64], 128, (3, 3, 3)),
]


GEMM1_CASES = [
    # name, T,H,W, segs, cout, M-pad
    ("to_v_like", 2, 8, 32, [128], 128),
    ("fh2_like_54_of_64", 3, 5, 40, [256], 54),
    ("block16_384", 5, 20, 32, [384], 384),
    ("two_segments_384", 2, 6, 10, [256, 128], 768),
    ("ragged_pixels", 1, 3, 11, [128], 96),
    ("mask_tail_144", 2, 8, 16, [256], 144),
]


@pytest.mark.parametrize("name,T,H,W,segs,cout", GEMM1_CASES)
def test_gemm1_vs_torch(lib, name, T, H, W, segs, cout):
    """The thin-GEMM kernel of the 1x1 convolutions (gemm1.hip: K split over the four waves of a workgroup, operands straight to registers,
    LDS reduction in wave order, shared row epilogue) against torch: one and two segments, every K it serves, padded / ragged couts, pixel
    counts that are not multiples of the 32-pixel tile, the epilogue kinds, and bit-identical results from run to run."""
    P = T * H * W
    xs = [hash_normal((P, c), 100 + i) for i, c in enumerate(segs)]
    cin = sum(segs)
    wt = hash_normal((cout, cin, 1, 1, 1), 200) / math.sqrt(cin)
    bs = hash_normal((cout,), 201) * 0.1
    ref = _ref_conv(xs, wt, bs, (1, 1, 1), T, H, W)
    got = _run_conv(lib, xs, wt, bs, (1, 1, 1), T, H, W, version=6)
    assert maxdiff(got, ref) < 3e-5 * max(1.0, ref.abs().max().item()), name
    assert torch.equal(got, _run_conv(lib, xs, wt, bs, (1, 1, 1), T, H, W, version=6))
    assert maxdiff(got, _run_conv(lib, xs, wt, bs, (1, 1, 1), T, H, W, version=2)) < 2e-5 * max(1.0, ref.abs().max().item())
    aux = hash_normal((P, cout), 303)
    L = lib
    assert maxdiff(_run_conv(L, xs, wt, bs, (1, 1, 1), T, H, W, version=6, kind=L.EPI_RESID, act=L.ACT_GELU, aux=aux), F.gelu(aux + ref)) < 5e-5 * max(1.0, ref.abs().max().item())
    assert maxdiff(_run_conv(L, xs, wt, bs, (1, 1, 1), T, H, W, version=6, act=L.ACT_ELU1), F.elu(ref) + 1) < 5e-5 * max(1.0, ref.abs().max().item())
    assert maxdiff(_run_conv(L, xs, wt, None, (1, 1, 1), T, H, W, version=6, scale=0.25), 0.25 * (ref - bs)) < 5e-5 * max(1.0, ref.abs().max().item())


STREAM_CASES = [
    # name, T,H,W, segs, cout, k3
    ("gru_1x15_block16", 5, 20, 32, [128, 384], 256, (1, 1, 15)),
    ("gru_1x15_hoisted", 2, 10, 18, [128, 256], 256, (1, 1, 15)),
    ("q_1x5", 2, 10, 18, [128, 256], 128, (1, 1, 5)),
    ("z_tail_1x5", 3, 7, 9, [128], 128, (1, 1, 5)),
    ("y_1x5x1", 2, 11, 9, [128, 64], 128, (1, 5, 1)),
    ("t_5x1x1", 5, 6, 10, [128, 64], 128, (5, 1, 1)),
    ("t_5x1x1_T2", 2, 5, 7, [64], 64, (5, 1, 1)),
    ("flow_head_3x3x3", 4, 7, 9, [128], 256, (3, 3, 3)),
    ("final_3x3_320_190", 2, 9, 13, [320], 190, (1, 3, 3)),
    ("unc_3x3_two_segs", 3, 9, 13, [128, 128], 128, (1, 3, 3)),
    ("linear_768", 2, 6, 10, [384, 384], 768, (1, 1, 1)),
    ("one_pixel", 1, 1, 1, [64], 64, (3, 3, 3)),
    ("frames_inside_a_tile", 5, 3, 5, [64], 64, (3, 3, 3)),
    ("w80_3x3", 2, 46, 80, [64], 64, (1, 3, 3)),
]


@pytest.mark.parametrize("hint", [1, 2])
@pytest.mark.parametrize("name,T,H,W,segs,cout,k3", STREAM_CASES)
def test_conv_stream_vs_torch(lib, name, T, H, W, segs, cout, k3, hint):
    """The register-streamed small-map kernel (conv_stream.hip: four waves share the chunks of every tap of one tile, operands straight to
    registers through a ring, LDS reduction in wave order) against torch conv3d: every tap shape of the update block, one and two segments,
    ragged pixel counts and couts, tiles that span several frames (temporal-tap skipping), both tile sizes, bit-identical from run to run,
    and against the LDS-staged kernel it replaces."""
    P = T * H * W
    xs = [hash_normal((P, c), 100 + i) for i, c in enumerate(segs)]
    cin = sum(segs)
    wt = hash_normal((cout, cin, *k3), 200) / math.sqrt(cin * k3[0] * k3[1] * k3[2])
    bs = hash_normal((cout,), 201) * 0.1
    ref = _ref_conv(xs, wt, bs, k3, T, H, W)
    got = _run_conv(lib, xs, wt, bs, k3, T, H, W, version=7, wm=hint)
    tol = 3e-5 * max(1.0, ref.abs().max().item())
    assert maxdiff(got, ref) < tol, name
    assert torch.equal(got, _run_conv(lib, xs, wt, bs, k3, T, H, W, version=7, wm=hint))
    assert maxdiff(got, _run_conv(lib, xs, wt, bs, k3, T, H, W, version=2)) < 2e-5 * max(1.0, ref.abs().max().item())
    aux = hash_normal((P, cout), 303)
    z = torch.sigmoid(hash_normal((P, cout), 304))
    got = _run_conv(lib, xs, wt, bs, k3, T, H, W, version=7, wm=hint, kind=lib.EPI_GRU, aux=aux, z=z)
    assert maxdiff(got, (1 - z) * aux + z * torch.tanh(ref)) < 5e-5, name


def test_conv_stream_epilogues_and_hoisted_share(lib):
    """Every epilogue class of the shared row epilogue through conv_stream, and the hoisted input share (pre_f32) of the GRU gates."""
    import functools
    L = lib
    T, H, W, k3 = 2, 6, 10, (1, 3, 3)
    P = T * H * W
    x = hash_normal((P, 64), 300)
    wt = hash_normal((64, 64, *k3), 301) / math.sqrt(64 * 9)
    bs = hash_normal((64,), 302) * 0.1
    aux = hash_normal((P, 64), 303)
    z = torch.sigmoid(hash_normal((P, 64), 304))
    lin = _ref_conv([x], wt, bs, k3, T, H, W)
    run = functools.partial(_run_conv, L, [x], wt, bs, k3, T, H, W, version=7)
    assert maxdiff(run(act=L.ACT_RELU), F.relu(lin)) < 5e-5
    assert maxdiff(run(act=L.ACT_GELU), F.gelu(lin)) < 5e-5
    assert maxdiff(run(act=L.ACT_SIGMOID), torch.sigmoid(lin)) < 5e-5
    assert maxdiff(run(act=L.ACT_TANH), torch.tanh(lin)) < 5e-5
    assert maxdiff(run(scale=0.25), 0.25 * lin) < 5e-5
    assert maxdiff(run(kind=L.EPI_RESID, act=L.ACT_GELU, aux=aux), F.gelu(aux + lin)) < 5e-5
    assert maxdiff(run(kind=L.EPI_RH, aux=aux), torch.sigmoid(lin) * aux) < 5e-5
    assert maxdiff(run(kind=L.EPI_GRU, aux=aux, z=z), (1 - z) * aux + z * torch.tanh(lin)) < 5e-5
    # hoisted share: conv([h | inp | rest]) == conv_h_rest([h | rest]) + pre
    T, H, W, k3 = 2, 6, 24, (1, 1, 5)
    P = T * H * W
    h, inp, rest = hash_normal((P, 128), 400), hash_normal((P, 128), 401), hash_normal((P, 64), 402)
    wt = hash_normal((128, 320, *k3), 403) / math.sqrt(320 * 5)
    bs = hash_normal((128,), 404) * 0.1
    full = _ref_conv([h, inp, rest], wt, bs, k3, T, H, W)
    pre = _run_conv(L, [inp], wt[:, 128:256].contiguous(), bs, k3, T, H, W, version=7)
    w_h = torch.cat([wt[:, :128], wt[:, 256:]], 1).contiguous()
    aux = hash_normal((P, 128), 405)
    z = torch.sigmoid(hash_normal((P, 128), 406))
    run = lambda **k: _run_conv(L, [h, rest], w_h, None, k3, T, H, W, version=7, pre=pre, **k)
    assert maxdiff(run(), full) < 5e-5
    assert maxdiff(run(act=L.ACT_GELU), F.gelu(full)) < 5e-5
    assert maxdiff(run(kind=L.EPI_RH, aux=aux), torch.sigmoid(full) * aux) < 5e-5
    assert maxdiff(run(kind=L.EPI_GRU, aux=aux, z=z), (1 - z) * aux + z * torch.tanh(full)) < 5e-5


@pytest.mark.parametrize("version", [7, 2])
@pytest.mark.parametrize("T,H,W,halo,k3", [(3, 4, 8, 2, (5, 1, 1)), (2, 5, 7, 1, (3, 3, 3)), (1, 3, 5, 2, (5, 1, 1))])
def test_conv_temporal_halo_slabs(lib, version, T, H, W, halo, k3):
    """ppms_conv.t_halo (frame-sharded windows, ppmstereo_amd/dist.py): the T frames of a rank sit between `halo` readable frames of its
    neighbours, and temporal taps read those instead of zero padding.  A conv over the middle T frames of a (T + 2 halo)-frame volume with
    t_halo = halo must equal the middle of the conv over the whole volume wherever the taps stay inside it (every output frame: the halo is as
    deep as the taps reach) -- conv_stream (tiles that span frames, per-tile temporal-tap skipping) and conv_gemm2."""
    from ppmstereo_amd.engine import ConvOp, epilogue
    from ppmstereo_amd.packing import pack_conv2, pack_stream
    L = lib
    assert halo >= k3[0] // 2
    Tf, HW = T + 2 * halo, H * W
    cin, cout = 64, 64
    xf = hash_normal((Tf * HW, cin), 600)
    wt = hash_normal((cout, cin, *k3), 601) / math.sqrt(cin * k3[0] * k3[1] * k3[2])
    bs = hash_normal((cout,), 602) * 0.1
    ref = _ref_conv([xf], wt, bs, k3, Tf, H, W)[halo * HW:(halo + T) * HW]
    xt = L.SPTensor(T * HW, cin, DEV, before=halo * HW, after=halo * HW)
    hi = xf.to(torch.bfloat16)
    xt.data[0].copy_(hi.to(DEV))
    xt.data[1].copy_((xf - hi.float()).to(torch.bfloat16).to(DEV))
    packed, b, meta = (pack_stream if version == 7 else pack_conv2)(wt.to(DEV), bs.to(DEV), [cin], [cin])
    out = L.SPTensor(T * HW, meta["M"], DEV)
    d = L.Conv()
    d.seg[0] = xt.view()
    d.nseg, d.w, d.bias = 1, packed.data_ptr(), b.data_ptr()
    d.T, d.H, d.W = T, H, W
    d.kt, d.kh, d.kw = k3
    d.t_halo = halo
    d.M = d.m_split = meta["M"]
    d.epi[0] = epilogue(n_valid=cout, out_sp=out.view())
    ConvOp(d, [packed, b], version, nslice=1)()
    torch.cuda.synchronize()
    got = out.to_f32()[:, :cout].cpu()
    assert maxdiff(got, ref) < 3e-5 * max(1.0, ref.abs().max().item())


def test_conv_stream_refuses_what_it_does_not_serve(lib):
    L = lib
    d = L.Conv()
    assert L.load().ppms_conv_stream_applicable(C.byref(d)) == 0
    with pytest.raises(RuntimeError):
        L.check(L.load().ppms_conv_stream(C.byref(d), None, 0, None))


@pytest.mark.parametrize("T,H,W", [(2, 8, 32), (3, 23, 40), (1, 5, 7)])
def test_gemm1_two_halves_and_transposed_v(lib, T, H, W):
    """Two epilogue halves in one launch (the linear attention's q | k and v projections: rows 0..767 elu + 1 -> fp32, rows 768..1151
    scaled -> fp32) and the attention's transposed bf16 V operand (to_v: SP output + V^T [frame][128][H*W], with H*W a multiple of 32 --
    16-byte row pieces -- and not)."""
    from ppmstereo_amd.engine import ConvOp, epilogue
    from ppmstereo_amd.packing import pack_gemm1
    L = lib
    P = T * H * W
    x = hash_normal((P, 384), 410)
    xt = L.SPTensor(P, 384, DEV)
    xt.set_f32(x.to(DEV))
    wq = hash_normal((1152, 384, 1, 1), 411) / math.sqrt(384)
    packed, b, meta = pack_gemm1(wq.to(DEV), None, [384])
    qkf, vf = torch.zeros(P, 768, device=DEV), torch.zeros(P, 384, device=DEV)
    d = L.Conv()
    d.seg[0] = xt.view()
    d.nseg, d.w, d.bias = 1, packed.data_ptr(), b.data_ptr()
    d.T, d.H, d.W, d.kt, d.kh, d.kw = T, H, W, 1, 1, 1
    d.M, d.m_split = 1152, 768
    d.epi[0] = epilogue(act=L.ACT_ELU1, n_valid=768, out_f32=qkf, out_f32_ld=768)
    d.epi[1] = epilogue(scale=0.5, n_valid=384, out_f32=vf, out_f32_ld=384)
    assert L.load().ppms_gemm1_applicable(C.byref(d)) in (1, 2)
    ConvOp(d, [xt, packed, b, qkf, vf], 6)()
    torch.cuda.synchronize()
    ref = x @ wq[:, :, 0, 0].t()
    assert maxdiff(qkf, F.elu(ref[:, :768]) + 1) < 5e-5 * ref.abs().max().item() and maxdiff(vf, 0.5 * ref[:, 768:]) < 5e-5 * ref.abs().max().item()
    # to_v: SP + V^T
    x2 = hash_normal((P, 128), 420)
    x2t = L.SPTensor(P, 128, DEV)
    x2t.set_f32(x2.to(DEV))
    wv = hash_normal((128, 128, 1, 1), 421) / math.sqrt(128)
    packed, b, meta = pack_gemm1(wv.to(DEV), None, [128])
    out = L.SPTensor(P, 128, DEV)
    vt = torch.zeros(T, 128, H * W, dtype=torch.bfloat16, device=DEV)
    d = L.Conv()
    d.seg[0] = x2t.view()
    d.nseg, d.w, d.bias = 1, packed.data_ptr(), b.data_ptr()
    d.T, d.H, d.W, d.kt, d.kh, d.kw = T, H, W, 1, 1, 1
    d.M = d.m_split = 128
    d.epi[0] = epilogue(n_valid=128, out_sp=out.view(), out_vt=vt)
    ConvOp(d, [x2t, packed, b, out, vt], 6)()
    torch.cuda.synchronize()
    y = out.to_f32()
    assert maxdiff(y, x2 @ wv[:, :, 0, 0].t()) < 3e-5 * 4
    want = out.own()[0].reshape(T, H * W, 128).permute(0, 2, 1)          # the hi plane IS bf16(y)
    from ppmstereo_amd.engine import attn_p_format
    assert torch.equal(vt, L.vt_image(want, attn_p_format())), "V^T must be the bf16 rounding of the SP output (its hi plane), transposed, in TUNING['attn_p']'s format"
    # both formats explicitly (ppms_epilogue.vt_f16): bf16 itself / the fp16 image of the same numbers
    for fmt in (L.ATTN_P_BF16, L.ATTN_P_FP16):
        vt2 = torch.zeros_like(vt)
        d.epi[0] = epilogue(n_valid=128, out_sp=out.view(), out_vt=vt2, vt_f16=fmt)
        ConvOp(d, [x2t, packed, b, out, vt2], 6)()
        torch.cuda.synchronize()
        assert torch.equal(vt2, L.vt_image(want, fmt))
        # the same numbers: exactly, for every bf16 value in fp16's normal range; to 2^-25 absolute below it (fp16 subnormals)
        held, w32 = L.vt_values(vt2, fmt), want.float()
        normal = w32.abs() >= 2.0 ** -14
        assert torch.equal(held[normal], w32[normal]) and (held - w32).abs().max().item() <= 2.0 ** -25


@pytest.mark.parametrize("name,T,H,W,segs,cout,k3,nslice", [
    ("x15_scale8", 5, 40, 64, [128, 256], 256, (1, 1, 15), -1), ("x15_scale16", 5, 20, 32, [128, 256], 256, (1, 1, 15), -1),
    ("y5_m128", 5, 20, 32, [128, 256], 128, (1, 5, 1), 4), ("3x3_m128_uneven", 2, 24, 40, [64, 32], 100, (1, 3, 3), 2),
    ("3x3x3_m128_T5", 5, 10, 40, [128], 128, (3, 3, 3), 5), ("3x3_m256_scale8", 5, 40, 64, [128], 256, (1, 3, 3), -1)])
def test_conv_gemm5_sliced_vs_torch(lib, name, T, H, W, segs, cout, k3, nslice):
    """conv_gemm5's K-sliced form (small maps: nslice workgroups per tile, each with its share of the windows, fp32 partials + the
    slice-reduce kernel with the fused epilogue): x / y / 2-D sweeps, M = 256 and M = 128 (two K-groups per workgroup), uneven
    shares, temporal taps whose window count differs between frames, the library's own plan (-1) -- vs torch conv3d."""
    P = T * H * W
    xs = [hash_normal((P, c), 100 + i) for i, c in enumerate(segs)]
    cin = sum(segs)
    wt = hash_normal((cout, cin, *k3), 200) / math.sqrt(cin * k3[0] * k3[1] * k3[2])
    bs = hash_normal((cout,), 201) * 0.1
    ref = _ref_conv(xs, wt, bs, k3, T, H, W)
    seg_pad = [((c + 15) // 16) * 16 for c in segs]
    got = _run_conv(lib, xs, wt, bs, k3, T, H, W, version=5, seg_pad=seg_pad, nslice=nslice, act=1)
    assert maxdiff(got, torch.relu(ref)) < 3e-5 * max(1.0, ref.abs().max().item()), name
    again = _run_conv(lib, xs, wt, bs, k3, T, H, W, version=5, seg_pad=seg_pad, nslice=nslice, act=1)
    assert torch.equal(got, again), "sliced launches must be bit-reproducible"


@pytest.mark.parametrize("nbt", [5007, 5008])
@pytest.mark.parametrize("name,T,H,W,segs,cout,k3", LARGE_MAP_CASES + [("3x3_ragged_m256", 2, 13, 45, [48, 16], 190, (1, 3, 3)), ("x15_w24", 1, 9, 24, [32], 128, (1, 1, 15)),
                                                   ("3x3x3_m128_T5", 5, 10, 40, [128], 128, (3, 3, 3)),
                                                   # GEMM mode (kh = kw = 1): 64- / 32-channel windows, the temporal taps as separate windows
                                                   ("t5_gemm_m256", 5, 20, 64, [128, 256], 256, (5, 1, 1)), ("1x1_m256_pad", 2, 13, 45, [256], 144, (1, 1, 1)),
                                                   ("t5_gemm_m128", 5, 10, 40, [128, 256], 128, (5, 1, 1)), ("1x1_m128", 1, 9, 24, [64], 128, (1, 1, 1)),
                                                   ("t3_gemm_m128_T2", 2, 16, 32, [64, 64], 100, (3, 1, 1))])
def test_conv_gemm5_vs_torch(lib, name, T, H, W, segs, cout, k3, nbt):
    """One 8-wave workgroup per tile of 7 / 8 blocks of 32 pixels (4 + 3 split balanced per SIMD), all couts per workgroup:
    M = 256 (4 cout blocks x 2 pixel halves) and M = 128 (K loop split over two wave groups, partial tiles summed through LDS) --
    x / y / 2-D sweeps, temporal taps, two segments, padded couts, ragged maps -- vs torch conv3d, and bit-reproducible."""
    P = T * H * W
    xs = [hash_normal((P, c), 100 + i) for i, c in enumerate(segs)]
    cin = sum(segs)
    wt = hash_normal((cout, cin, *k3), 200) / math.sqrt(cin * k3[0] * k3[1] * k3[2])
    bs = hash_normal((cout,), 201) * 0.1
    ref = _ref_conv(xs, wt, bs, k3, T, H, W)
    seg_pad = [((c + 15) // 16) * 16 for c in segs]
    got = _run_conv(lib, xs, wt, bs, k3, T, H, W, version=nbt, seg_pad=seg_pad)
    assert maxdiff(got, ref) < 3e-5 * max(1.0, ref.abs().max().item()), name
    assert torch.equal(got, _run_conv(lib, xs, wt, bs, k3, T, H, W, version=nbt, seg_pad=seg_pad))
    aux = hash_normal((P, cout), 303)
    z = torch.sigmoid(hash_normal((P, cout), 304))
    got = _run_conv(lib, xs, wt, bs, k3, T, H, W, version=nbt, seg_pad=seg_pad, kind=lib.EPI_GRU, aux=aux, z=z)
    assert maxdiff(got, (1 - z) * aux + z * torch.tanh(ref)) < 5e-5, name


@pytest.mark.parametrize("nbt", [5007, 5008])
@pytest.mark.parametrize("name,T,H,W,segs,cout,k3", [("3x3_m192_final", 2, 13, 45, [320], 190, (1, 3, 3)), ("3x3_m192_full_blocks", 1, 16, 56, [256], 192, (1, 3, 3)),
                                                   ("x15_m192", 2, 9, 40, [64, 32], 160, (1, 1, 15)), ("y5_m192", 1, 24, 32, [48], 129, (1, 5, 1)),
                                                   ("3x3x3_m192_T3", 3, 10, 40, [64], 192, (3, 3, 3))])
def test_conv_gemm5_three_cout_blocks_vs_torch(lib, name, T, H, W, segs, cout, k3, nbt):
    """M = 192 (round 4): three 64-cout blocks dealt over the eight waves -- cout block 2 on waves 0 / 1 with 4 + (NBT - 4) pixel blocks, cout
    blocks 0 and 1 on three waves each with 3 + (NBT - 5) + 2 -- so that convc2's 192 and final_conv's 190 couts no longer run padded to
    256 rows.  x / y / 2-D sweeps, temporal taps, ragged maps and couts vs torch, bit-reproducible, with the GRU epilogue class."""
    P = T * H * W
    xs = [hash_normal((P, c), 100 + i) for i, c in enumerate(segs)]
    cin = sum(segs)
    wt = hash_normal((cout, cin, *k3), 200) / math.sqrt(cin * k3[0] * k3[1] * k3[2])
    bs = hash_normal((cout,), 201) * 0.1
    ref = _ref_conv(xs, wt, bs, k3, T, H, W)
    seg_pad = [((c + 15) // 16) * 16 for c in segs]
    got = _run_conv(lib, xs, wt, bs, k3, T, H, W, version=nbt, seg_pad=seg_pad, m_pad=192)
    assert maxdiff(got, ref) < 3e-5 * max(1.0, ref.abs().max().item()), name
    assert torch.equal(got, _run_conv(lib, xs, wt, bs, k3, T, H, W, version=nbt, seg_pad=seg_pad, m_pad=192))
    assert torch.equal(got, _run_conv(lib, xs, wt, bs, k3, T, H, W, version=nbt, seg_pad=seg_pad, m_pad=256)), "same bits as the 256-row layout"
    aux = hash_normal((P, cout), 303)
    z = torch.sigmoid(hash_normal((P, cout), 304))
    got = _run_conv(lib, xs, wt, bs, k3, T, H, W, version=nbt, seg_pad=seg_pad, kind=lib.EPI_GRU, aux=aux, z=z, m_pad=192)
    assert maxdiff(got, (1 - z) * aux + z * torch.tanh(ref)) < 5e-5, name


@pytest.mark.parametrize("name,T,H,W,segs,cout,k3,lz", [
    ("gru_1x15", 2, 12, 128, [128, 256], 256, (1, 1, 15), 256), ("q_1x5_m128", 2, 12, 128, [128, 256], 128, (1, 1, 5), 256),
    ("y_1x5x1", 2, 40, 32, [128, 256], 256, (1, 5, 1), 256), ("t_5x1x1_gemm_mode", 5, 16, 64, [128, 256], 256, (5, 1, 1), 256),
    ("3x3_m192_one_segment", 2, 13, 45, [320], 190, (1, 3, 3), 192)])
def test_conv_gemm5_skips_products_with_a_zero_lo_plane(lib, name, T, H, W, segs, cout, k3, lz):
    """ppms_conv.lo_zero_from: input channels that hold bf16-exact values (the attention's read-out hid) have an all-zero lo plane, and
    conv_gemm5 leaves their a_hi x b_lo MFMAs out (a scalar branch inside each such asm statement).  Same bits as the full product --
    every sweep mode, both K-group layouts, the GEMM mode's 64-channel windows; and against torch."""
    P = T * H * W
    xs = [hash_normal((P, c), 100 + i) for i, c in enumerate(segs)]
    cin = sum(segs)
    xcat = torch.cat(xs, 1)
    xcat[:, lz:] = xcat[:, lz:].to(torch.bfloat16).float()                  # bf16-exact from channel lz on
    xs = list(torch.split(xcat, segs, 1))
    wt = hash_normal((cout, cin, *k3), 200) / math.sqrt(cin * k3[0] * k3[1] * k3[2])
    bs = hash_normal((cout,), 201) * 0.1
    ref = _ref_conv(xs, wt, bs, k3, T, H, W)
    full = _run_conv(lib, xs, wt, bs, k3, T, H, W, version=5)
    skip = _run_conv(lib, xs, wt, bs, k3, T, H, W, version=5, lo_zero_from=lz)
    assert torch.equal(full, skip), name
    assert maxdiff(skip, ref) < 3e-5 * max(1.0, ref.abs().max().item()), name


@pytest.mark.parametrize("cout", [128, 256, 190])
def test_conv_gemm5_epilogue_classes(lib, cout):
    """conv_gemm5 instantiates its row loop per descriptor class (conv_epilogue.h): plain store (no load in the loop), hoisted share
    (operands of four rows fetched before they are stored), aux state (RESID / RH), GRU, and the run-time-checked rest -- and with 128
    couts both K-groups run it on swapped halves of the tile.  Every class against torch, at 128 / 256 / ragged 190 couts."""
    L = lib
    T, H, W, k3 = 2, 12, 64, (1, 3, 3)
    P = T * H * W
    xs = [hash_normal((P, 64), 700), hash_normal((P, 32), 701)]
    wt = hash_normal((cout, 96, *k3), 702) / math.sqrt(96 * 9)
    bs = hash_normal((cout,), 703) * 0.1
    lin = _ref_conv(xs, wt, bs, k3, T, H, W)
    aux, z, pre = hash_normal((P, cout), 704), torch.sigmoid(hash_normal((P, cout), 705)), hash_normal((P, cout), 706)
    run = lambda **k: _run_conv(L, xs, wt, bs, k3, T, H, W, version=5, **k)
    tol = lambda ref: 5e-5 * max(1.0, ref.abs().max().item())
    for act, f in ((L.ACT_NONE, lambda t: t), (L.ACT_RELU, F.relu), (L.ACT_GELU, F.gelu), (L.ACT_SIGMOID, torch.sigmoid), (L.ACT_TANH, torch.tanh)):
        assert maxdiff(run(act=act), f(lin)) < tol(f(lin)), ("plain", act)                                     # EPI_CLS_PLAIN
    if cout % 4 == 0:                                        # (pre_f32 is for couts in multiples of 4: the library refuses it otherwise)
        assert maxdiff(run(act=L.ACT_GELU, pre=pre), F.gelu(lin + pre)) < tol(lin), "pre"                       # EPI_CLS_PRE
        assert maxdiff(run(kind=L.EPI_GRU, aux=aux, z=z, pre=pre), (1 - z) * aux + z * torch.tanh(lin + pre)) < tol(lin), "gru + pre"
        assert maxdiff(run(kind=L.EPI_RH, aux=aux, pre=pre), torch.sigmoid(lin + pre) * aux) < tol(lin), "rh + pre"   # EPI_CLS_ANY
    assert maxdiff(run(kind=L.EPI_RESID, act=L.ACT_GELU, aux=aux), F.gelu(aux + lin)) < tol(lin), "resid"       # EPI_CLS_AUX
    assert maxdiff(run(kind=L.EPI_RH, aux=aux), torch.sigmoid(lin) * aux) < tol(lin), "rh"
    assert maxdiff(run(kind=L.EPI_GRU, aux=aux, z=z), (1 - z) * aux + z * torch.tanh(lin)) < tol(lin), "gru"    # EPI_CLS_GRU
    assert torch.equal(run(kind=L.EPI_GRU, aux=aux, z=z), run(kind=L.EPI_GRU, aux=aux, z=z))


def test_conv_gemm5_full_map(lib):
    """The shapes the 1/4 scale of BASELINE config 2 really runs (5 x 80 x 128 pixels) through the library's own tile choice
    (7-block tiles: 240 workgroups): the GRU (1,1,15) conv to 256 couts, a 3x3 conv to 256, and the q-gate conv to 128 couts."""
    T, H, W = 5, 80, 128
    P = T * H * W
    for segs, cout, k3, m_pad in (([128, 256], 256, (1, 1, 15), None), ([128], 256, (1, 3, 3), None), ([128, 256], 128, (1, 1, 5), None),
                                  ([320], 190, (1, 3, 3), 192)):                       # (final_conv on the three-cout-block layout)
        xs = [hash_normal((P, c), 100 + i) for i, c in enumerate(segs)]
        cin = sum(segs)
        wt = hash_normal((cout, cin, *k3), 200) / math.sqrt(cin * k3[0] * k3[1] * k3[2])
        bs = hash_normal((cout,), 201) * 0.1
        ref = _ref_conv(xs, wt, bs, k3, T, H, W)
        got = _run_conv(lib, xs, wt, bs, k3, T, H, W, version=5, m_pad=m_pad)
        assert maxdiff(got, ref) < 3e-5 * max(1.0, ref.abs().max().item()), (segs, k3)


CONV6_CASES = [
    # name, T,H,W, segs, cout, k3 -- x / y / 2-D sweeps, temporal taps, two segments, ragged maps and couts, M = 256 and 128
    ("x15_gru_m256", 2, 20, 40, [128, 256], 256, (1, 1, 15)), ("x5_m128", 2, 12, 128, [128, 256], 128, (1, 1, 5)), ("x5_one_seg_m128", 1, 33, 27, [128], 128, (1, 1, 5)),
    ("y5_m256", 2, 40, 32, [128, 256], 256, (1, 5, 1)), ("y5_m128_ragged", 1, 37, 29, [64, 32], 100, (1, 5, 1)),
    ("3x3_m256", 2, 13, 45, [128], 256, (1, 3, 3)), ("3x3_two_segs_m128", 3, 9, 13, [128, 128], 128, (1, 3, 3)), ("3x3x3_m256_T4", 4, 17, 19, [128], 200, (3, 3, 3)),
    ("3x3x3_m128_T5", 5, 10, 40, [128], 128, (3, 3, 3)), ("x15_w24", 1, 9, 24, [32], 128, (1, 1, 15)), ("y3_m128", 2, 20, 30, [64], 128, (1, 3, 1)),
    ("T1_tiny", 1, 1, 3, [32], 130, (3, 3, 3)),
    # no spatial taps: the STREAM form (one k32-step per window, three window buffers) -- the GRU's pass T, a plain 1x1, one frame, ragged maps
    ("t5_gru_m256", 5, 20, 40, [128, 256], 256, (5, 1, 1)), ("t5_m128_ragged", 5, 17, 29, [128], 100, (5, 1, 1)), ("t3_m256_200", 4, 20, 30, [64, 32], 200, (3, 1, 1)),
    ("1x1_m256", 2, 20, 30, [128], 256, (1, 1, 1)), ("t5_T1", 1, 9, 24, [32], 128, (5, 1, 1)), ("1x1_one_window", 1, 16, 13, [32], 128, (1, 1, 1)),
]


@pytest.mark.parametrize("name,T,H,W,segs,cout,k3", CONV6_CASES)
def test_conv_gemm6_vs_torch(lib, name, T, H, W, segs, cout, k3):
    """conv_gemm6 (round 5): one wave per SIMD on v_mfma_f32_16x16x32_bf16, tiles of 16 rows x 13 columns, 32-channel windows stored column-major
    (the STREAM form without spatial taps: one 32-channel window per k32-step, three buffers) -- M = 256 (4 waves x 64 couts x 13 pixel blocks) and M = 128 (2 x 64 couts x 7 / 6 blocks) -- vs torch
    conv3d, bit-reproducible, with the GRU epilogue class."""
    P = T * H * W
    xs = [hash_normal((P, c), 100 + i) for i, c in enumerate(segs)]
    cin = sum(segs)
    wt = hash_normal((cout, cin, *k3), 200) / math.sqrt(cin * k3[0] * k3[1] * k3[2])
    bs = hash_normal((cout,), 201) * 0.1
    ref = _ref_conv(xs, wt, bs, k3, T, H, W)
    seg_pad = [((c + 31) // 32) * 32 for c in segs]
    m_pad = 128 if cout <= 128 else 256
    got = _run_conv(lib, xs, wt, bs, k3, T, H, W, version=8, seg_pad=seg_pad, m_pad=m_pad)
    assert maxdiff(got, ref) < 3e-5 * max(1.0, ref.abs().max().item()), name
    assert torch.equal(got, _run_conv(lib, xs, wt, bs, k3, T, H, W, version=8, seg_pad=seg_pad, m_pad=m_pad))
    aux = hash_normal((P, cout), 303)
    z = torch.sigmoid(hash_normal((P, cout), 304))
    got = _run_conv(lib, xs, wt, bs, k3, T, H, W, version=8, seg_pad=seg_pad, m_pad=m_pad, kind=lib.EPI_GRU, aux=aux, z=z)
    assert maxdiff(got, (1 - z) * aux + z * torch.tanh(ref)) < 5e-5, name


@pytest.mark.parametrize("name,T,H,W,segs,cout,k3", [("3x3_m192_final", 2, 13, 45, [320], 190, (1, 3, 3)), ("3x3_m192_full_blocks", 1, 16, 56, [256], 192, (1, 3, 3)),
                                                   ("x15_m192", 2, 9, 40, [64, 32], 160, (1, 1, 15)), ("y5_m192", 1, 24, 32, [64], 129, (1, 5, 1)),
                                                   ("3x3x3_m192_T3", 3, 10, 40, [64], 192, (3, 3, 3))])
def test_conv_gemm6_three_cout_blocks_vs_torch(lib, name, T, H, W, segs, cout, k3):
    """M = 192: four waves x 48 couts (three 16-cout MFMA blocks each) x 13 pixel blocks -- convc2's 192 and final_conv's 190 couts."""
    P = T * H * W
    xs = [hash_normal((P, c), 100 + i) for i, c in enumerate(segs)]
    cin = sum(segs)
    wt = hash_normal((cout, cin, *k3), 200) / math.sqrt(cin * k3[0] * k3[1] * k3[2])
    bs = hash_normal((cout,), 201) * 0.1
    ref = _ref_conv(xs, wt, bs, k3, T, H, W)
    seg_pad = [((c + 31) // 32) * 32 for c in segs]
    got = _run_conv(lib, xs, wt, bs, k3, T, H, W, version=8, seg_pad=seg_pad, m_pad=192)
    assert maxdiff(got, ref) < 3e-5 * max(1.0, ref.abs().max().item()), name
    assert torch.equal(got, _run_conv(lib, xs, wt, bs, k3, T, H, W, version=8, seg_pad=seg_pad, m_pad=192))
    assert torch.equal(got, _run_conv(lib, xs, wt, bs, k3, T, H, W, version=8, seg_pad=seg_pad, m_pad=256)), "same bits as the 256-row layout"
    aux = hash_normal((P, cout), 303)
    z = torch.sigmoid(hash_normal((P, cout), 304))
    got = _run_conv(lib, xs, wt, bs, k3, T, H, W, version=8, seg_pad=seg_pad, kind=lib.EPI_GRU, aux=aux, z=z, m_pad=192)
    assert maxdiff(got, (1 - z) * aux + z * torch.tanh(ref)) < 5e-5, name


@pytest.mark.parametrize("name,T,H,W,segs,cout,k3,lz", [
    ("gru_1x15", 2, 12, 128, [128, 256], 256, (1, 1, 15), 256), ("q_1x5_m128", 2, 12, 128, [128, 256], 128, (1, 1, 5), 256),
    ("y_1x5x1", 2, 40, 32, [128, 256], 256, (1, 5, 1), 256), ("t_3x3x3_two_phases", 4, 16, 40, [128, 256], 256, (3, 3, 3), 256),
    ("3x3_m192_one_segment", 2, 13, 45, [320], 190, (1, 3, 3), 192), ("t5_stream_gru", 5, 12, 40, [128, 256], 256, (5, 1, 1), 256),
    ("t5_stream_q_m128", 5, 12, 40, [128, 256], 128, (5, 1, 1), 256)])
def test_conv_gemm6_skips_products_with_a_zero_lo_plane(lib, name, T, H, W, segs, cout, k3, lz):
    """ppms_conv.lo_zero_from on conv_gemm6: the windows whose lo plane is all zero run in a second phase of the K loop whose step body has no
    hi x lo MFMAs.  Same bits as the full product where both visit the windows in the same order (one temporal tap); with temporal taps the
    two-phase order (all full windows of every tap, then all skipped ones) sums in another order than the one-phase loop, and so does the K-split form
    of M = 128 (each wave pair takes half of the windows with and half of those without a lo plane): equal to fp32 rounding."""
    P = T * H * W
    xs = [hash_normal((P, c), 100 + i) for i, c in enumerate(segs)]
    cin = sum(segs)
    xcat = torch.cat(xs, 1)
    xcat[:, lz:] = xcat[:, lz:].to(torch.bfloat16).float()                  # bf16-exact from channel lz on
    xs = list(torch.split(xcat, segs, 1))
    wt = hash_normal((cout, cin, *k3), 200) / math.sqrt(cin * k3[0] * k3[1] * k3[2])
    bs = hash_normal((cout,), 201) * 0.1
    ref = _ref_conv(xs, wt, bs, k3, T, H, W)
    m_pad = 128 if cout <= 128 else 192 if cout <= 192 else 256
    full = _run_conv(lib, xs, wt, bs, k3, T, H, W, version=8, m_pad=m_pad)
    skip = _run_conv(lib, xs, wt, bs, k3, T, H, W, version=8, m_pad=m_pad, lo_zero_from=lz)
    if k3[0] == 1 and m_pad != 128:
        assert torch.equal(full, skip), name
    else:        # (M = 128 with a spatial sweep runs K-split: the two wave pairs' shares of the windows depend on where the zero-lo windows begin)
        assert maxdiff(full, skip) < 1e-5 * max(1.0, ref.abs().max().item()), name
    assert torch.equal(skip, _run_conv(lib, xs, wt, bs, k3, T, H, W, version=8, m_pad=m_pad, lo_zero_from=lz)), "bit-reproducible"
    assert maxdiff(skip, ref) < 3e-5 * max(1.0, ref.abs().max().item()), name


@pytest.mark.parametrize("cout", [128, 256, 190])
def test_conv_gemm6_epilogue_classes(lib, cout):
    """Every epilogue class of conv_epilogue.h through conv_gemm6's two-pass staging (accumulators of <= 7 pixel blocks -> wave-private LDS patch ->
    8 couts of a pixel per lane), at 128 / 256 / ragged 190 (M = 192) couts."""
    L = lib
    T, H, W, k3 = 2, 12, 64, (1, 3, 3)
    P = T * H * W
    xs = [hash_normal((P, 64), 700), hash_normal((P, 32), 701)]
    wt = hash_normal((cout, 96, *k3), 702) / math.sqrt(96 * 9)
    bs = hash_normal((cout,), 703) * 0.1
    lin = _ref_conv(xs, wt, bs, k3, T, H, W)
    aux, z, pre = hash_normal((P, cout), 704), torch.sigmoid(hash_normal((P, cout), 705)), hash_normal((P, cout), 706)
    m_pad = 128 if cout <= 128 else 192 if cout <= 192 else 256
    run = lambda **k: _run_conv(L, xs, wt, bs, k3, T, H, W, version=8, m_pad=m_pad, **k)
    tol = lambda ref: 5e-5 * max(1.0, ref.abs().max().item())
    for act, f in ((L.ACT_NONE, lambda t: t), (L.ACT_RELU, F.relu), (L.ACT_GELU, F.gelu), (L.ACT_SIGMOID, torch.sigmoid), (L.ACT_TANH, torch.tanh)):
        assert maxdiff(run(act=act), f(lin)) < tol(f(lin)), ("plain", act)
    if cout % 4 == 0:
        assert maxdiff(run(act=L.ACT_GELU, pre=pre), F.gelu(lin + pre)) < tol(lin), "pre"
        assert maxdiff(run(kind=L.EPI_GRU, aux=aux, z=z, pre=pre), (1 - z) * aux + z * torch.tanh(lin + pre)) < tol(lin), "gru + pre"
        assert maxdiff(run(kind=L.EPI_RH, aux=aux, pre=pre), torch.sigmoid(lin + pre) * aux) < tol(lin), "rh + pre"
    assert maxdiff(run(kind=L.EPI_RESID, act=L.ACT_GELU, aux=aux), F.gelu(aux + lin)) < tol(lin), "resid"
    assert maxdiff(run(kind=L.EPI_RH, aux=aux), torch.sigmoid(lin) * aux) < tol(lin), "rh"
    assert maxdiff(run(kind=L.EPI_GRU, aux=aux, z=z), (1 - z) * aux + z * torch.tanh(lin)) < tol(lin), "gru"
    assert torch.equal(run(kind=L.EPI_GRU, aux=aux, z=z), run(kind=L.EPI_GRU, aux=aux, z=z))


def test_conv_gemm6_full_map_and_rating(lib):
    """The shapes the 1/4 scale of BASELINE config 2 really runs (5 x 80 x 128 pixels = 250 tiles of 16 x 13): the GRU (1,1,15) conv to 256 couts, a 3x3
    conv to 256, the q-gate conv to 128 couts, final_conv on the 192-row layout, the temporal pass; the library rates all of them 1 there and 0 on
    a 1/8-scale map (too few tiles)."""
    T, H, W = 5, 80, 128
    P = T * H * W
    for segs, cout, k3, m_pad in (([128, 256], 256, (1, 1, 15), 256), ([128], 256, (1, 3, 3), 256), ([128, 256], 128, (1, 1, 5), 128),
                                  ([320], 190, (1, 3, 3), 192), ([128], 256, (3, 3, 3), 256), ([128, 256], 256, (5, 1, 1), 256), ([128, 256], 128, (5, 1, 1), 128)):
        xs = [hash_normal((P, c), 100 + i) for i, c in enumerate(segs)]
        cin = sum(segs)
        wt = hash_normal((cout, cin, *k3), 200) / math.sqrt(cin * k3[0] * k3[1] * k3[2])
        bs = hash_normal((cout,), 201) * 0.1
        ref = _ref_conv(xs, wt, bs, k3, T, H, W)
        got = _run_conv(lib, xs, wt, bs, k3, T, H, W, version=8, m_pad=m_pad)
        assert maxdiff(got, ref) < 3e-5 * max(1.0, ref.abs().max().item()), (segs, k3)
    d = lib.Conv()
    sp = lib.SPTensor(16, 128, DEV)
    d.seg[0], d.nseg, d.kt, d.kh, d.kw, d.M, d.m_split = sp.view(), 1, 1, 3, 3, 256, 256
    d.T, d.H, d.W = 5, 80, 128
    assert lib.load().ppms_conv_gemm6_applicable(C.byref(d)) == 1
    d.T, d.H, d.W = 5, 40, 64
    assert lib.load().ppms_conv_gemm6_applicable(C.byref(d)) == 0
    d.T, d.H, d.W, d.kh, d.kw, d.kt = 5, 80, 128, 1, 1, 5               # no spatial taps: the STREAM form
    assert lib.load().ppms_conv_gemm6_applicable(C.byref(d)) == 1


def test_conv_gemm6_two_epilogue_halves(lib):
    """m_split: the z | r convs (M = 256, m_split = 128: waves 0-1 / 2-3) and final_conv (M = 192, m_split = 128: wave 2's 48 couts straddle the
    split, its cout groups take their half's epilogue in two passes)."""
    from ppmstereo_amd.engine import ConvOp, epilogue
    from ppmstereo_amd.packing import pack_conv6
    T, H, W, k3 = 2, 20, 30, (1, 3, 3)
    P = T * H * W
    for M, n0, n1 in ((256, 128, 128), (192, 126, 64)):
        cmap = list(range(n0)) + list(range(128, 128 + n1))
        x = hash_normal((P, 64), 810)
        wt = hash_normal((n0 + n1, 64, *k3), 811) / math.sqrt(64 * 9)
        bs = hash_normal((n0 + n1,), 812) * 0.1
        ref = _ref_conv([x], wt, bs, k3, T, H, W)
        xin = lib.SPTensor(P, 64, DEV)
        xin.set_f32(x.to(DEV))
        w5 = wt[:, :, None]
        packed, b, meta = pack_conv6(w5.reshape(n0 + n1, 64, 1, 1, 9).to(DEV), bs.to(DEV), [64], [64], cmap, M)
        o0, o1 = lib.SPTensor(P, 128, DEV), torch.zeros(P, 64, device=DEV)
        d = lib.Conv()
        d.seg[0], d.nseg, d.w, d.bias = xin.view(), 1, packed.data_ptr(), b.data_ptr()
        d.T, d.H, d.W, d.kt, d.kh, d.kw, d.M, d.m_split = T, H, W, 1, 3, 3, M, 128
        d.epi[0] = epilogue(act=lib.ACT_RELU, n_valid=n0, out_sp=o0.view())
        d.epi[1] = epilogue(act=lib.ACT_TANH, n_valid=min(n1, 64), out_f32=o1, out_f32_ld=64)
        ConvOp(d, [xin, packed, b, o0, o1], 8)()
        torch.cuda.synchronize()
        assert maxdiff(o0.to_f32()[:, :n0].cpu(), torch.relu(ref[:, :n0])) < 3e-5 * max(1.0, ref.abs().max().item()), M
        assert maxdiff(o1[:, :min(n1, 64)].cpu(), torch.tanh(ref[:, n0:n0 + min(n1, 64)])) < 3e-5, M
        assert (o0.to_f32()[:, n0:] == 0).all()


@pytest.mark.parametrize("T,H,W,kw", [(2, 20, 30, 5), (1, 16, 13, 5), (3, 33, 40, 3), (5, 80, 128, 5)])
def test_conv_gemm6_grouped_two_tails_in_one_launch(lib, T, H, W, kw):
    """ppms_conv.groups = 2: the two 128 -> 128 (1,1,5) tails of convz1 / convr1 (ppmtereo_update.py:254-312) as ONE conv_gemm6 launch -- segment 0 (the
    z branch's gelu output) feeds couts 0..127 with a sigmoid -> fp32 epilogue, segment 1 (the r branch's) feeds couts 128..255 with the r * h epilogue --
    against torch conv3d of each tail and against the two separate M = 128 launches (fp32 rounding apart: those run K-split); bit-reproducible; the other
    convolution entry points refuse a grouped descriptor."""
    from ppmstereo_amd.engine import ConvOp, epilogue
    from ppmstereo_amd.packing import pack_conv6, pack_conv6_grouped
    L = lib
    P = T * H * W
    k3 = (1, 1, kw)
    xz, xr, h = hash_normal((P, 128), 820), hash_normal((P, 128), 821), hash_normal((P, 128), 822)
    wz, wr = hash_normal((128, 128, *k3), 823) / math.sqrt(128 * kw), hash_normal((128, 128, *k3), 824) / math.sqrt(128 * kw)
    bz, br = hash_normal((128,), 825) * 0.1, hash_normal((128,), 826) * 0.1
    zt, rt, ht = L.SPTensor(P, 128, DEV), L.SPTensor(P, 128, DEV), L.SPTensor(P, 128, DEV)
    zt.set_f32(xz.to(DEV)), rt.set_f32(xr.to(DEV)), ht.set_f32(h.to(DEV))
    packed, b, meta = pack_conv6_grouped([wz.to(DEV), wr.to(DEV)], [bz.to(DEV), br.to(DEV)], 128)
    Z, RH = torch.zeros(P, 128, device=DEV), L.SPTensor(P, 128, DEV)
    d = L.Conv()
    d.seg[0], d.seg[1], d.nseg, d.groups = zt.view(), rt.view(), 2, 2
    d.w, d.bias = packed.data_ptr(), b.data_ptr()
    d.T, d.H, d.W, d.kt, d.kh, d.kw, d.M, d.m_split = T, H, W, 1, 1, kw, 256, 128
    d.epi[0] = epilogue(act=L.ACT_SIGMOID, n_valid=128, out_f32=Z, out_f32_ld=128)
    d.epi[1] = epilogue(L.EPI_RH, n_valid=128, out_sp=RH.view(), aux_sp=ht.view())
    # (the library RATES only maps with enough tiles for the chip -- config 2's 1/4 scale is the last case; the kernel itself serves every size)
    assert L.load().ppms_conv_gemm6_applicable(C.byref(d)) == (1 if P >= 40000 else 0)
    ConvOp(d, [zt, rt, ht, packed, b, Z, RH], 8)()
    torch.cuda.synchronize()
    ref_z = torch.sigmoid(_ref_conv([xz], wz, bz, k3, T, H, W))
    ref_r = torch.sigmoid(_ref_conv([xr], wr, br, k3, T, H, W)) * h
    assert maxdiff(Z.cpu(), ref_z) < 3e-5 and maxdiff(RH.to_f32().cpu(), ref_r) < 3e-5 * max(1.0, h.abs().max().item())
    # the two separate launches (M = 128 layout): the same bits
    Z2, RH2 = torch.zeros(P, 128, device=DEV), L.SPTensor(P, 128, DEV)
    for x_, w_, b_, e_ in ((zt, wz, bz, epilogue(act=L.ACT_SIGMOID, n_valid=128, out_f32=Z2, out_f32_ld=128)),
                           (rt, wr, br, epilogue(L.EPI_RH, n_valid=128, out_sp=RH2.view(), aux_sp=ht.view()))):
        p1, b1, m1 = pack_conv6(w_.to(DEV), b_.to(DEV), [128], None, None, 128)
        d1 = L.Conv()
        d1.seg[0], d1.nseg, d1.w, d1.bias = x_.view(), 1, p1.data_ptr(), b1.data_ptr()
        d1.T, d1.H, d1.W, d1.kt, d1.kh, d1.kw, d1.M, d1.m_split = T, H, W, 1, 1, kw, 128, 128
        d1.epi[0] = e_
        ConvOp(d1, [x_, p1, b1, Z2, RH2, ht], 8)()
    torch.cuda.synchronize()
    # (the single M = 128 launches run K-split -- (first half of K) + (second half) per cout --, the grouped launch sums K in one pass: fp32 rounding apart)
    assert maxdiff(Z, Z2) < 1e-6 and maxdiff(RH.to_f32(), RH2.to_f32()) < 1e-5 * max(1.0, h.abs().max().item())
    # repeatable, and refused elsewhere
    Zc = Z.clone()
    ConvOp(d, [zt, rt, ht, packed, b, Z, RH], 8)()
    torch.cuda.synchronize()
    assert torch.equal(Z, Zc)
    for ver in (2, 5):
        with pytest.raises(RuntimeError, match="grouped"):
            ConvOp(d, [zt, rt, ht, packed, b, Z, RH], ver)()
    assert L.load().ppms_conv_stream_applicable(C.byref(d)) == 0 and L.load().ppms_gemm1_applicable(C.byref(d)) == 0


@pytest.mark.parametrize("version", [5, 2])
def test_conv_gemm_hoisted_input_share(lib, version):
    """conv([h | inp | rest]) == conv_h_rest([h | rest]) + pre, pre = conv_inp(inp) + bias computed by another launch
    (the engine hoists the inp share of the GRU gates out of the iteration loop): every epilogue adds pre_f32 to
    acc + bias before its activation."""
    L = lib
    T, H, W, k3 = 2, 6, 128, (1, 1, 5)
    P = T * H * W
    h, inp, rest = hash_normal((P, 128), 400), hash_normal((P, 128), 401), hash_normal((P, 64), 402)
    wt = hash_normal((128, 320, *k3), 403) / math.sqrt(320 * 5)
    bs = hash_normal((128,), 404) * 0.1
    full = _ref_conv([h, inp, rest], wt, bs, k3, T, H, W)
    pre = _run_conv(L, [inp], wt[:, 128:256].contiguous(), bs, k3, T, H, W, version=version)
    w_h = torch.cat([wt[:, :128], wt[:, 256:]], 1).contiguous()
    aux = hash_normal((P, 128), 405)
    z = torch.sigmoid(hash_normal((P, 128), 406))
    run = lambda **k: _run_conv(L, [h, rest], w_h, None, k3, T, H, W, version=version, pre=pre, **k)
    assert maxdiff(run(), full) < 5e-5
    assert maxdiff(run(act=L.ACT_GELU), F.gelu(full)) < 5e-5
    assert maxdiff(run(kind=L.EPI_RH, aux=aux), torch.sigmoid(full) * aux) < 5e-5
    assert maxdiff(run(kind=L.EPI_GRU, aux=aux, z=z), (1 - z) * aux + z * torch.tanh(full)) < 5e-5


@pytest.mark.parametrize("name,T,H,W,segs,cout,k3,nslice", [
    ("gru_1x15", 5, 20, 32, [128, 384], 256, (1, 1, 15), 8), ("q_1x5", 2, 10, 18, [128, 256], 128, (1, 1, 5), 4),
    ("t_5x1x1", 5, 6, 10, [128, 64], 128, (5, 1, 1), 3), ("3x3x3", 4, 7, 9, [128], 190, (3, 3, 3), 2), ("3x3_320", 2, 9, 13, [320], 190, (1, 3, 3), 5)])
def test_conv_gemm_k_sliced(lib, name, T, H, W, segs, cout, k3, nslice):
    """Grid-level K slicing for small maps: partial tiles + deterministic reduce kernel with the fused epilogue."""
    L = lib
    P = T * H * W
    xs = [hash_normal((P, c), 100 + i) for i, c in enumerate(segs)]
    cin = sum(segs)
    wt = hash_normal((cout, cin, *k3), 200) / math.sqrt(cin * k3[0] * k3[1] * k3[2])
    bs = hash_normal((cout,), 201) * 0.1
    ref = _ref_conv(xs, wt, bs, k3, T, H, W)
    got = _run_conv(L, xs, wt, bs, k3, T, H, W, nslice=nslice)
    assert maxdiff(got, ref) < 3e-5 * max(1.0, ref.abs().max().item()), name
    again = _run_conv(L, xs, wt, bs, k3, T, H, W, nslice=nslice)
    assert torch.equal(got, again), "slice reduction must be order-deterministic"
    aux = hash_normal((P, cout), 303)
    z = torch.sigmoid(hash_normal((P, cout), 304))
    got = _run_conv(L, xs, wt, bs, k3, T, H, W, nslice=nslice, kind=L.EPI_GRU, aux=aux, z=z)
    assert maxdiff(got, (1 - z) * aux + z * torch.tanh(ref)) < 5e-5, name


@pytest.mark.parametrize("name,T,H,W,segs,cout,k3,nslice", [
    ("gru_1x5x1", 5, 40, 64, [128, 256], 256, (1, 5, 1), 1), ("gru_sliced", 5, 20, 32, [128, 384], 128, (1, 5, 1), 4),
    ("ragged", 2, 11, 9, [64], 64, (1, 5, 1), 1), ("w80", 1, 23, 80, [64, 32], 192, (1, 5, 1), 3), ("3x_t", 3, 10, 18, [32], 64, (3, 3, 1), 1),
    ("2d_3x3", 2, 9, 13, [128, 128], 128, (1, 3, 3), 1), ("2d_3x3x3_sliced", 4, 20, 32, [128], 190, (3, 3, 3), 4), ("2d_w80", 2, 23, 80, [320], 190, (1, 3, 3), 5),
    ("2d_5x3", 1, 12, 40, [64], 64, (1, 5, 3), 2)])
def test_conv_gemm2_y_sweep(lib, name, T, H, W, segs, cout, k3, nslice):
    """conv_gemm2's one-window forms: y-swept (kt, kh, 1) convs (column-major patch / window) and the 2-D window for
    kh, kw > 1, with and without K slicing, vs torch conv3d."""
    L = lib
    P = T * H * W
    xs = [hash_normal((P, c), 100 + i) for i, c in enumerate(segs)]
    cin = sum(segs)
    wt = hash_normal((cout, cin, *k3), 200) / math.sqrt(cin * k3[0] * k3[1] * k3[2])
    bs = hash_normal((cout,), 201) * 0.1
    ref = _ref_conv(xs, wt, bs, k3, T, H, W)
    got = _run_conv(L, xs, wt, bs, k3, T, H, W, nslice=nslice, ysweep=True)
    assert maxdiff(got, ref) < 3e-5 * max(1.0, ref.abs().max().item()), name
    aux = hash_normal((P, cout), 303)
    z = torch.sigmoid(hash_normal((P, cout), 304))
    got = _run_conv(L, xs, wt, bs, k3, T, H, W, nslice=nslice, ysweep=True, kind=L.EPI_GRU, aux=aux, z=z)
    assert maxdiff(got, (1 - z) * aux + z * torch.tanh(ref)) < 5e-5, name


def test_conv_gemm_rejects_bad_descriptors(lib):
    L = lib
    d = L.Conv()
    with pytest.raises(RuntimeError):
        L.check(L.load().ppms_conv_gemm2(C.byref(d), None, 0, None))


# ------------------------------------------------------------------------------------------------ small ops
def test_convex_upsample(lib):
    from ppmstereo_amd.ppmstereo import convex_upsample
    fl, mk = hash_normal((3, 2, 6, 10), 31), hash_normal((3, 144, 6, 10), 32)
    out = convex_upsample(fl.to(DEV), mk.to(DEV))
    assert maxdiff(out, O.convex_upsample(fl, mk)) < 5e-6
    Golden("convex_upsample").check("out", out, 5e-6)


@pytest.mark.parametrize("align", [True, False])
@pytest.mark.parametrize("shape,size", [((2, 3, 5, 7), (10, 14)), ((1, 2, 8, 32), (32, 128)), ((2, 1, 6, 5), (9, 11)), ((1, 1, 4, 4), (4, 4))])
def test_bilinear(lib, align, shape, size):
    from ppmstereo_amd.engine import bilinear
    x = hash_normal(shape, 400)
    if align:
        ref = F.interpolate(x, size=size, mode="bilinear", align_corners=True)
    elif size[0] % shape[2] == 0 and size[0] // shape[2] == size[1] // shape[3]:
        ref = F.interpolate(x, scale_factor=size[0] // shape[2], mode="bilinear")
    else:
        ref = F.interpolate(x, size=size, mode="bilinear", align_corners=False)
    assert maxdiff(bilinear(x.to(DEV), size, align, 1.0), ref) < 2e-6
    assert maxdiff(bilinear(x.to(DEV), size, align, -0.5), -0.5 * ref) < 2e-6


def test_layout_converters_roundtrip(lib):
    L = lib
    x = hash_normal((3, 36, 5, 7), 500)
    t = L.SPTensor(3 * 35, 64, DEV)
    xd = x.to(DEV)
    L.check(L.load().ppms_nchw_to_sp(xd.data_ptr(), t.view(0, 36), 3, 36, 35, L.stream_ptr()))
    back = torch.empty(3, 36, 5, 7, device=DEV)
    L.check(L.load().ppms_sp_to_nchw(t.view(0, 36), back.data_ptr(), 3, 36, 35, L.stream_ptr()))
    assert maxdiff(back, x) < 2e-5 and (t.to_f32()[:, 36:] == 0).all()
    assert maxdiff(t.to_f32()[:, :36].reshape(3, 5, 7, 36).permute(0, 3, 1, 2), x) < 2e-5
    hi = t.data[0, :, :36].float().cpu().reshape(3, 5, 7, 36).permute(0, 3, 1, 2)
    assert torch.equal(hi, x.to(torch.bfloat16).float()), "hi plane must be the RNE bf16 cast (it doubles as the attention V operand)"
    f = torch.empty(3 * 35, 40, device=DEV)
    L.check(L.load().ppms_nchw_to_nhwc(xd.data_ptr(), f.data_ptr(), 40, 3, 36, 35, L.stream_ptr()))
    b2 = torch.empty_like(back)
    L.check(L.load().ppms_nhwc_to_nchw(f.data_ptr(), 40, b2.data_ptr(), 3, 36, 35, L.stream_ptr()))
    assert maxdiff(b2, x) == 0


@pytest.mark.parametrize("T,h,w", [(5, 8, 32), (8, 20, 32), (3, 46, 80), (2, 10, 18)])
def test_qk_similarity(lib, T, h, w):
    L = lib
    q, k = hash_normal((T, 128, h, w), 600), hash_normal((T, 128, h, w), 601) + 0.3 * hash_normal((T, 128, h, w), 600)
    qk = torch.cat([q, k], 1).permute(0, 2, 3, 1).reshape(T * h * w, 256).contiguous().to(DEV)
    pooled = torch.zeros(2, T, (h // 4) * (w // 4), device=DEV)
    sim = torch.zeros(T, T, device=DEV)
    L.check(L.load().ppms_qk_similarity(qk.data_ptr(), qk.data_ptr() + 512, 256, pooled.data_ptr(), sim.data_ptr(), T, h, w, L.stream_ptr()))
    assert maxdiff(sim, O.qk_similarity(q, k)) < 2e-6


def test_qam_select_sequence(lib):
    """QAM scoring / top-5 pick / usage counter over several iterations with T = 8 > top_k (ppmstereo.py:501-513)."""
    L = lib
    T, HW = 8, 700
    nblk = (HW + 255) // 256
    sim = torch.tanh(hash_normal((T, T), 700))
    strive_ref = torch.ones(T, T)
    strive = torch.ones(T, T, device=DEV)
    sel = torch.zeros(T, 5, dtype=torch.int32, device=DEV)
    shat, score = torch.zeros(T, 5, device=DEV), torch.zeros(T, T, device=DEV)
    for it in range(6):
        unc = torch.sigmoid(hash_normal((T, HW), 710 + it))
        part = torch.zeros(T, nblk)
        for b in range(nblk):
            part[:, b] = unc[:, b * 256:(b + 1) * 256].sum(1)
        sc, mask, strive_ref = O.qam_select(sim, strive_ref, unc.mean(1))
        simd, partd = sim.to(DEV), part.to(DEV)             # keep the device copies alive across the launch
        L.check(L.load().ppms_qam_select(simd.data_ptr(), strive.data_ptr(), partd.data_ptr(), nblk, HW, sel.data_ptr(), shat.data_ptr(),
                                         score.data_ptr(), T, L.stream_ptr()))
        assert maxdiff(score, sc) < 2e-6
        for i in range(T):
            J = torch.nonzero(mask[i]).flatten()
            assert sel[i].cpu().tolist() == J.tolist(), (it, i)
            s = sc[i, J]
            assert maxdiff(shat[i], s / s.mean()) < 2e-6
        assert torch.equal(strive.cpu(), strive_ref)


# ------------------------------------------------------------------------------------------------ memory attention
P_FORMATS = [pytest.param(1, id="p_fp16"), pytest.param(0, id="p_bf16")]      # ppms_mem_attn's p_format (include/ppms.h)


def _attn_check(got, ref, operands, scale, p_format=0):
    """Per-element bound in ulps OF THE ELEMENT (not of the tensor's range), against the UNROUNDED fp32 read-out x = softmax(Q K^T scale) V on
    the bf16 operands the oracle uses (`ref` = the oracle's result = bf16(x): checked to be exactly that rounding).  The kernel's two extra
    roundings give |got - x| <= eps_P * (P |V|) (probabilities rounded to p_format before the PV product, worst case: all errors aligned;
    eps_P = 2^-8 for bf16 P~ -- 8 significant bits, half an ulp <= 2^-8 of the value --, 2^-11 for fp16 P~; fp16's subnormal tail adds at most
    2^-25 per key relative to the row maximum, covered by the slop) + 2^-8 |x| (the bf16 result: half an ulp) + 1e-6 fp32 slop; and the typical
    error must sit far inside that worst case (rounding errors do not all line up)."""
    eps_p = 2.0 ** -11 if p_format == 1 else 2.0 ** -8
    for i, (Q, K, V, _, _) in enumerate(operands):
        Qb, Kb, Vb = (t.to(torch.bfloat16).float() for t in (Q, K, V))
        P = torch.softmax((Qb @ Kb.t()) * scale, dim=-1)
        x = P @ Vb
        assert ((ref[i] - x).abs() <= 2.0 ** -8 * x.abs() + 1e-6).all(), "the oracle's result is not the bf16 rounding of this read-out"
        bound = eps_p * (P @ Vb.abs()) + 2.0 ** -8 * x.abs() + 1e-6
        err = (got[i] - x).abs()
        assert torch.isfinite(got[i]).all()
        worst = (err / bound).max().item()
        assert worst <= 1.0, f"clip {i}: element error {worst:.2f}x its bound"
        # (the final rounding alone averages a quarter ulp = ~0.35 of its half-ulp term; the P~ term must add little to that)
        assert err.mean().item() <= 0.5 * bound.mean().item(), (err.mean().item(), bound.mean().item())


# attn_frames: ppms_mem_attn's frames_per_workgroup argument (0 = the library's choice, which needs a 1/4-scale-sized grid to pick 2; 5 = all picked
# frames in one workgroup, one partial set per clip)
@pytest.mark.parametrize("p_format", P_FORMATS)
@pytest.mark.parametrize("attn_frames", [0, 2, 5])
@pytest.mark.parametrize("split", [False, True])
@pytest.mark.parametrize("T,n,ksel_frames", [(5, 256, 5), (8, 1024, 5), (2, 256, 2), (3, 180, 3), (6, 200, 5), (5, 320, 5)])      # (320: a partly filled 256-query block)
def test_mem_attn_vs_oracle(lib, T, n, ksel_frames, split, attn_frames, p_format):
    """prep_q + prep_k + mem_attn against play_inputs + flash_attn_math (ppmstereo.py:517-552)."""
    L = lib
    if attn_frames and not (split and n % 64 == 0):
        pytest.skip("frames per workgroup only concerns the 64-query kernel")
    from ppmstereo_amd.engine import softmax_scale, temporal_pe
    h, w = (n // 32, 32) if n % 32 == 0 else (n // 20, 20) if n % 20 == 0 else (n // 18, 18)
    assert h * w == n
    q, key, value = hash_normal((T, 128, h, w), 800), hash_normal((T, 128, h, w), 801), hash_normal((T, 128, h, w), 802)
    mf = hash_normal((T, 128, h, w), 803)
    pe = temporal_pe(T, 128)
    score = 1.0 + 0.3 * torch.tanh(hash_normal((T, T), 804))
    mask = torch.zeros(T, T, dtype=torch.bool)
    for i in range(T):
        mask[i, torch.argsort(score[i], descending=True)[:ksel_frames]] = True
    sel = torch.zeros(T, 5, dtype=torch.int32)
    shat = torch.zeros(T, 5)
    ref = torch.zeros(T, n, 128)
    scale = softmax_scale(128)
    for i in range(T):
        Q, K, V, J, s_hat = O.play_inputs(q, key, pe, value, score, mask, i)
        sel[i, :len(J)] = J.int()
        shat[i, :len(J)] = s_hat
        ref[i] = O.flash_attn_math(Q, K, V, scale)
    cl = lambda x: x.permute(0, 2, 3, 1).reshape(T * n, 128).contiguous()
    qk = torch.cat([cl(q), cl(key)], 1).contiguous().to(DEV)
    qb = torch.zeros(T, n, 128, dtype=torch.bfloat16, device=DEV)
    kb = torch.zeros(T, ksel_frames, n, 128, dtype=torch.bfloat16, device=DEV)
    vt = L.vt_image(value.reshape(T, 128, n), p_format).to(DEV)
    ped, seld, shatd = pe.to(DEV), sel.to(DEV), shat.to(DEV)
    lib_ = L.load()
    s = L.stream_ptr()
    L.check(lib_.ppms_attn_prep_q(qk.data_ptr(), 256, ped.data_ptr(), qb.data_ptr(), T, n, s))
    L.check(lib_.ppms_attn_prep_k(qk.data_ptr() + 512, 256, ped.data_ptr(), seld.data_ptr(), shatd.data_ptr(), kb.data_ptr(), T, ksel_frames, n, s))
    X = L.SPTensor(T * n, 256, DEV)
    X.set_f32(cl(mf).to(DEV), 0)
    beta = torch.tensor([0.5], device=DEV)
    raw = torch.zeros(T, n, 128, dtype=torch.bfloat16, device=DEV)
    ws = torch.empty(int(lib_.ppms_mem_attn_workspace_bytes(T, ksel_frames, n)), dtype=torch.uint8, device=DEV) if split else None
    L.check(lib_.ppms_mem_attn(qb.data_ptr(), kb.data_ptr(), vt.data_ptr(), seld.data_ptr(), ksel_frames, scale, beta.data_ptr(), X.view(0, 128), X.view(128, 128),
                               raw.data_ptr(), T, n, L.ptr(ws), attn_frames, p_format, s))
    torch.cuda.synchronize()
    # operands: bit-exact bf16 of the oracle's fp32 operands
    Q0, K0, _, _, _ = O.play_inputs(q, key, pe, value, score, mask, 0)
    assert torch.equal(qb[0].float().cpu(), Q0.to(torch.bfloat16).float())
    assert torch.equal(kb[0].reshape(-1, 128).float().cpu(), K0.to(torch.bfloat16).float())
    # output, per element: the kernel rounds P to p_format before the PV product (flash-attention: bf16) and the result to bf16
    # |err_d| <= eps_P * sum_k p_k |v_kd| (P rounding, worst case) + half a bf16 ulp of the element (final rounding)
    _attn_check(raw.float().cpu(), ref, [(O.play_inputs(q, key, pe, value, score, mask, i)) for i in range(T)], scale, p_format)
    mfg = X.to_f32(128, 128).cpu()
    # mfg is stored split (hi + lo): ~2^-16 relative
    assert maxdiff(mfg, cl(mf) + 0.5 * raw.float().cpu().reshape(T * n, 128)) < 1e-4
    # mf.hi == NULL: no aggregation, the view receives hid itself -- hi plane = the bf16 read-out, lo plane all zero (ppms_conv.lo_zero_from)
    X.own()[1, :, 128:] = 1.0
    raw2 = torch.zeros_like(raw)
    L.check(lib_.ppms_mem_attn(qb.data_ptr(), kb.data_ptr(), vt.data_ptr(), seld.data_ptr(), ksel_frames, scale, beta.data_ptr(), L.SP(None, None, 0, 0), X.view(128, 128),
                               raw2.data_ptr(), T, n, L.ptr(ws), attn_frames, p_format, s))
    torch.cuda.synchronize()
    assert torch.equal(raw2, raw)
    assert torch.equal(X.own()[0, :, 128:].reshape(T, n, 128), raw) and (X.own()[1, :, 128:] == 0).all()


@pytest.mark.parametrize("p_format", P_FORMATS)
@pytest.mark.parametrize("attn_frames", [0, 2])
@pytest.mark.parametrize("boost", [40.0, 3.0, 0.7])
def test_mem_attn_sharp_softmax(lib, boost, attn_frames, p_format):
    """(attn_frames = 2: the dominating key sits in the SECOND frame of the workgroup's pair, whose scores are taken relative to the first
    frame's reference.)  One key per query dominates and sits in a late tile.  boost = 40: the score jumps ~650 log2 units above the
    first keys, far beyond what the rescale-free 64-query kernel carries (2^60 with bf16 P~, 2^16 with fp16 P~): its workgroups raise their
    redo flags and the 32-query online-softmax kernel recomputes them.  boost = 3: ~50 log2 units -- carried in-kernel with bf16 P~ (P up to
    2^50), redone with fp16 P~.  boost = 0.7: ~11 log2 units, carried in-kernel by both (the flags say which path ran)."""
    L = lib
    from ppmstereo_amd.engine import softmax_scale
    T, n = 2, 512
    q = hash_normal((T, n, 128), 900)
    k = hash_normal((T, 2, n, 128), 901) * 0.1
    v = hash_normal((T, 128, n), 902)
    for i in range(n):
        k[0, 1, (i * 7 + 300) % n] += boost * q[0, i] / q[0, i].norm()
    qb, kb, vt = q.to(torch.bfloat16).to(DEV), k.to(torch.bfloat16).to(DEV), L.vt_image(v, p_format).to(DEV)
    sel = torch.tensor([[0, 1, 0, 0, 0], [0, 1, 0, 0, 0]], dtype=torch.int32, device=DEV)
    X = L.SPTensor(T * n, 256, DEV)
    beta = torch.tensor([1.0], device=DEV)
    raw = torch.zeros(T, n, 128, dtype=torch.bfloat16, device=DEV)
    scale = 1.0
    ws = torch.empty(int(L.load().ppms_mem_attn_workspace_bytes(T, 2, n)), dtype=torch.uint8, device=DEV)
    L.check(L.load().ppms_mem_attn(qb.data_ptr(), kb.data_ptr(), vt.data_ptr(), sel.data_ptr(), 2, scale, beta.data_ptr(), X.view(0, 128), X.view(128, 128),
                                   raw.data_ptr(), T, n, ws.data_ptr(), attn_frames, p_format, L.stream_ptr()))
    torch.cuda.synchronize()
    # which path ran: the redo flags of the 64-query kernel sit behind the partials in the workspace (ppms_mem_attn_workspace_bytes)
    flags = ws.view(torch.int32)[T * 2 * n * 130:].cpu()
    nsplit = 1 if attn_frames == 2 else 2
    flags = flags[:T * nsplit * 2 * 2].reshape(T, nsplit, 2, 2)[..., 0]          # [clip][split][256-query block]
    carried = boost < 1.0 or (boost == 3.0 and p_format == 0)
    hit = flags[0, nsplit - 1]                                                   # clip 0, the split that holds the boosted frame
    assert bool(hit.any()) != carried, (boost, p_format, flags.tolist())
    assert not flags[1].any(), "clip 1 has no dominating key: nothing to redo"
    vv = L.vt_values(vt, p_format)
    for i in range(T):
        K = kb[i].reshape(-1, 128).float().cpu()
        V = torch.cat([vv[0].cpu().t(), vv[1].cpu().t()], 0)
        ref = O.flash_attn_math(qb[i].float().cpu(), K, V, scale)
        _attn_check(raw[i].float().cpu()[None], ref[None], [(qb[i].float().cpu(), K, V, None, None)], scale, p_format)


# ------------------------------------------------------------------------------------------------ small fused ops
@pytest.mark.parametrize("k", [7, 1])
@pytest.mark.parametrize("BT,H,W,cv", [(2, 9, 18, 40), (1, 5, 7, 40), (3, 20, 32, 40), (1, 8, 64, 64)])
def test_dwconv_gelu(lib, k, BT, H, W, cv):
    """depthwise k x k conv + residual GELU (ppmtereo_update.py:1026-1027) on a channel view of an SP tensor."""
    L = lib
    P = BT * H * W
    x = hash_normal((P, 64), 500)
    w = torch.zeros(64, k * k)
    b = torch.zeros(64)
    w[:36] = hash_normal((36, k * k), 501) / k
    b[:36] = hash_normal((36,), 502) * 0.1
    xs, ys = L.SPTensor(P, 64, DEV), L.SPTensor(P, 64, DEV)
    xs.set_f32(x.to(DEV))
    wd, bd = w.to(DEV), b.to(DEV)
    L.check(L.load().ppms_dwconv_gelu(xs.view(0, cv), ys.view(0, cv), wd.data_ptr(), bd.data_ptr(), k, BT, H, W, L.stream_ptr()))
    torch.cuda.synchronize()
    xq = xs.to_f32().cpu()                                                    # the split-bf16 planes hold ~16 mantissa bits
    xi = xq[:, :cv].reshape(BT, H, W, cv).permute(0, 3, 1, 2)
    ref = F.gelu(xi + F.conv2d(xi, w[:cv].reshape(cv, 1, k, k), b[:cv], padding=k // 2, groups=cv)).permute(0, 2, 3, 1).reshape(P, cv)
    got = ys.to_f32().cpu()
    assert maxdiff(got[:, :cv], ref) < 2e-5 * max(1.0, ref.abs().max().item())
    assert (got[:, cv:] == 0).all(), "channels outside the view must not be written"


@pytest.mark.parametrize("T,H,W", [(3, 5, 9), (1, 4, 6), (5, 10, 16)])
def test_tap_gather_sum(lib, T, H, W):
    """FlowHead3D.conv2 (256 -> 2, 3x3x3) = 1x1 GEMM to 54 channels + shifted sum; checked against conv3d."""
    L = lib
    P = T * H * W
    x = hash_normal((P, 256), 510)
    wt = hash_normal((2, 256, 3, 3, 3), 511) / math.sqrt(256 * 27)
    bs = hash_normal((2,), 512)
    w1 = wt.permute(2, 3, 4, 0, 1).reshape(54, 256, 1, 1, 1).contiguous()       # row = tap * 2 + cout
    y = _run_conv(L, [x], w1, None, (1, 1, 1), T, H, W)                          # (P, 54)
    yd = torch.zeros(P, 64, device=DEV)
    yd[:, :54] = y.to(DEV)
    out = torch.zeros(P, 4, device=DEV)
    bd = bs.to(DEV)
    flow = hash_normal((P, 2), 513).to(DEV)
    flow0 = flow.clone()
    L.check(L.load().ppms_tap_gather_sum(yd.data_ptr(), 64, bd.data_ptr(), out.data_ptr(), 4, flow.data_ptr(), 2, 2, 3, 3, 3, T, H, W, 0, L.stream_ptr()))
    torch.cuda.synchronize()
    ref = _ref_conv([x], wt, bs, (3, 3, 3), T, H, W)
    assert maxdiff(out[:, :2], ref) < 3e-5
    assert torch.equal(flow, flow0 + out[:, :2])                                    # the fused flow += delta_flow (ppmstereo.py:571)
    out2 = torch.zeros(P, 4, device=DEV)
    L.check(L.load().ppms_tap_gather_sum(yd.data_ptr(), 64, bd.data_ptr(), out2.data_ptr(), 4, None, 0, 2, 3, 3, 3, T, H, W, 0, L.stream_ptr()))
    torch.cuda.synchronize()
    assert torch.equal(out2, out)


@pytest.mark.parametrize("P", [5 * 7 * 9 + 3, 16384 + 128 + 5])
def test_pwchain_vs_unfused_layers(lib, P):
    """Fused per-pixel chains of the correlation encoder (pwchain.hip) against fp32 torch math of
    PCBlock4_Deep_nopool_res.forward (ppmtereo_update.py:1024-1030), ragged pixel counts: 318 pixels run the small-map kernel (32-pixel
    tiles, one phase per layer), 16 517 the large-map one (128-pixel tiles, one phase per 64-cout block)."""
    from ppmstereo_amd.engine import PwChain
    from ppmstereo_amd.packing import pack_conv2
    L = lib
    x = hash_normal((P, 36), 520)
    mk = lambda co, ci, s: (hash_normal((co, ci, 1, 1), s) / math.sqrt(ci), hash_normal((co,), s + 1) * 0.1)
    (w0, b0), (w2, b2), (wp, bp), (w3, b3), (w4, b4) = mk(54, 36, 521), mk(36, 54, 523), mk(36, 36, 525), mk(54, 36, 527), mk(256, 54, 529)
    dws, dwt = hash_normal((36,), 531) * 0.5, hash_normal((36,), 532) * 0.1
    pk = lambda w, b, ci: pack_conv2(w.to(DEV), b.to(DEV), [ci], [64])
    xs, mid, out = L.SPTensor(P, 64, DEV), L.SPTensor(P, 64, DEV), L.SPTensor(P, 256, DEV)
    xp = torch.zeros(P, 64)
    xp[:, :36] = x
    xs.set_f32(xp.to(DEV))
    s64, t64 = torch.zeros(64, device=DEV), torch.zeros(64, device=DEV)
    s64[:36], t64[:36] = dws.to(DEV), dwt.to(DEV)
    PwChain(xs.view(), mid.view(), [(pk(w0, b0, 36), 54, False, None), (pk(w2, b2, 54), 36, True, (s64, t64))], P, [])()
    torch.cuda.synchronize()
    lin = lambda v, w, b: v @ w.reshape(w.shape[0], -1).t() + b
    x = xs.to_f32().cpu()[:, :36]
    x1 = F.gelu(x + lin(F.gelu(lin(x, w0, b0)), w2, b2))
    x2 = F.gelu(x1 + (x1 * dws + dwt))
    got = mid.to_f32().cpu()
    assert maxdiff(got[:, :36], x2) < 3e-5 * max(1.0, x2.abs().max().item())     # split-bf16 storage: ~2^-17 relative
    assert (got[:, 36:] == 0).all()
    # chain B on x2 (the engine runs the depthwise 7x7 in between): gelu(x + pw x) -> ffn2 -> outer gelu
    PwChain(mid.view(), out.view(), [(pk(wp, bp, 36), 36, True, None), (pk(w3, b3, 36), 54, False, None), (pk(w4, b4, 54), 256, False, None)], P, [])()
    torch.cuda.synchronize()
    x2g = got[:, :36]
    x4 = F.gelu(x2g + lin(x2g, wp, bp))
    ref = F.gelu(lin(F.gelu(lin(x4, w3, b3)), w4, b4))
    assert maxdiff(out.to_f32(), ref) < 5e-5 * max(1.0, ref.abs().max().item())


@pytest.mark.parametrize("T,n", [(5, 640), (3, 77), (2, 1000)])
def test_attn16_kernels(lib, T, n):
    """TimeAttnBlock core, LayerNorm(+residual) and LinearAttention kernels (attn16.hip) vs fp32 torch math
    (ppmtereo_update.py:593-631, attention.py:73-100)."""
    L = lib
    lb = L.load()
    P, Cc, heads, d = T * n, 384, 8, 48
    x = hash_normal((P, Cc), 540)
    lw, lbias = 1 + 0.1 * hash_normal((Cc,), 541), 0.1 * hash_normal((Cc,), 542)
    xs, o = L.SPTensor(P, Cc, DEV), L.SPTensor(P, Cc, DEV)
    xs.set_f32(x.to(DEV))
    lwd, lbd = lw.to(DEV), lbias.to(DEV)
    L.check(lb.ppms_time_attn(xs.view(), lwd.data_ptr(), lbd.data_ptr(), o.view(), T, n, heads, L.stream_ptr()))
    torch.cuda.synchronize()
    tok = x.view(T, n, Cc).transpose(0, 1)
    y = F.layer_norm(tok, (Cc,), lw, lbias, 1e-5).reshape(n, T, heads, d).permute(0, 2, 1, 3)
    att = torch.softmax((y @ y.transpose(-2, -1)) * d ** -0.5, dim=-1)
    ref = (att @ y).transpose(1, 2).reshape(n, T, Cc).transpose(0, 1).reshape(P, Cc)
    assert maxdiff(o.to_f32(), ref) < 5e-5 * max(1.0, ref.abs().max().item())
    # LayerNorm with and without residual
    xd = x.to(DEV).contiguous()
    none_sp = L.SP(None, None, 0, 0)
    L.check(lb.ppms_layernorm(xd.data_ptr(), Cc, lwd.data_ptr(), lbd.data_ptr(), none_sp, o.view(), P, Cc, L.stream_ptr()))
    torch.cuda.synchronize()
    ln = F.layer_norm(x, (Cc,), lw, lbias, 1e-5)
    rel = max(1.0, (x + ln).abs().max().item())                              # split-bf16 storage: ~2^-17 relative
    assert maxdiff(o.to_f32(), ln) < 2e-5 * rel
    o2 = L.SPTensor(P, Cc, DEV)
    L.check(lb.ppms_layernorm(xd.data_ptr(), Cc, lwd.data_ptr(), lbd.data_ptr(), xs.view(), o2.view(), P, Cc, L.stream_ptr()))
    torch.cuda.synchronize()
    assert maxdiff(o2.to_f32(), xs.to_f32().cpu() + ln) < 2e-5 * rel
    # linear attention: Q, K already elu()+1, V already / n
    Q = (F.elu(hash_normal((P, Cc), 543)) + 1).contiguous()
    K = (F.elu(hash_normal((P, Cc), 544)) + 1).contiguous()
    V = (hash_normal((P, Cc), 545) / n).contiguous()
    Qd, Kd, Vd = Q.to(DEV), K.to(DEV), V.to(DEV)
    ws = torch.zeros(int(lb.ppms_linear_attention_workspace_floats(T, n, heads, d)), device=DEV)
    L.check(lb.ppms_linear_attention(Qd.data_ptr(), Cc, Kd.data_ptr(), Cc, Vd.data_ptr(), Cc, ws.data_ptr(), o.view(), T, n, heads, d, L.stream_ptr()))
    torch.cuda.synchronize()
    q4, k4, v4 = (t.view(T, n, heads, d) for t in (Q, K, V))
    KV = torch.einsum("nshd,nshv->nhdv", k4, v4)
    Z = 1 / (torch.einsum("nlhd,nhd->nlh", q4, k4.sum(1)) + 1e-6)
    ref = (torch.einsum("nlhd,nhdv,nlh->nlhv", q4, KV, Z) * n).reshape(P, Cc)
    assert maxdiff(o.to_f32(), ref) < 5e-5 * max(1.0, ref.abs().max().item())
