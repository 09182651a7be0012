"""CPU: host-side logic of the product (weight packing, descriptor geometry, state_dict layout, windowing)."""
import math

import pytest
import torch
import torch.nn.functional as F

from oracle import ppm_oracle as O
from ppmstereo_amd import weights as Wm
from ppmstereo_amd.packing import BK, pack_conv2, pack_conv4, unpack_conv2_reference, unpack_conv4_reference
from ppmstereo_amd.weights import hash_normal


def emulate_kernel(packed, bias, meta, xs, seg_pad, T, H, W):
    """What the implicit-GEMM kernels compute, restated with torch on the CPU from the PACKED weights (brought back to
    the plain K order k = tap*Cpad + ci, tap = (kz*kh + ky)*kw + kx) and shifted zero-padded pixel rows."""
    kt, kh, kw = meta["taps"]
    Wm_ = unpack_conv2_reference(packed.cpu(), meta["M"], meta["nk"], meta["taps"], meta["cpad"] // BK)        # [M][K]
    P = T * H * W
    cols = []
    xp = []
    for x, cp in zip(xs, seg_pad):
        xp.append(F.pad(x, (0, cp - x.shape[1])))
    xcat = torch.cat(xp, 1).reshape(T, H, W, -1)
    for kz in range(kt):
        for ky in range(kh):
            for kx in range(kw):
                dt, dy, dx = kz - kt // 2, ky - kh // 2, kx - kw // 2
                sh = torch.zeros_like(xcat)
                t0, t1 = max(0, -dt), min(T, T - dt)
                y0, y1 = max(0, -dy), min(H, H - dy)
                x0, x1 = max(0, -dx), min(W, W - dx)
                if t0 < t1 and y0 < y1 and x0 < x1:
                    sh[t0:t1, y0:y1, x0:x1] = xcat[t0 + dt:t1 + dt, y0 + dy:y1 + dy, x0 + dx:x1 + dx]
                cols.append(sh.reshape(P, -1))
    A = torch.cat(cols, 1)                                                    # [P][K]
    return A @ Wm_.t() + bias.cpu()


@pytest.mark.parametrize("segs,cout,k3", [([36], 54, (1, 1, 1)), ([128, 384], 256, (1, 1, 15)), ([128, 64], 128, (5, 1, 1)),
                                          ([128], 190, (3, 3, 3)), ([320], 190, (1, 3, 3))])
def test_packing_reproduces_the_convolution(segs, cout, k3):
    packer = pack_conv2
    T, H, W = 3, 4, 6
    P = T * H * W
    cin = sum(segs)
    xs = [hash_normal((P, c), 10 + i) for i, c in enumerate(segs)]
    wt = hash_normal((cout, cin, *k3), 20) / math.sqrt(cin * k3[0] * k3[1] * k3[2])
    bs = hash_normal((cout,), 21)
    seg_pad = [((c + 31) // 32) * 32 for c in segs]
    packed, bias, meta = packer(wt, bs, segs, seg_pad)
    assert packed.dtype == torch.bfloat16 and packed.numel() == 2 * meta["M"] * meta["nk"] * BK
    got = emulate_kernel(packed, bias, meta, xs, seg_pad, T, H, W)[:, :cout]
    x5 = torch.cat(xs, 1).reshape(1, T, H, W, cin).permute(0, 4, 1, 2, 3)
    ref = F.conv3d(x5, wt, bs, padding=tuple(k // 2 for k in k3)).permute(0, 2, 3, 4, 1).reshape(P, cout)
    assert (got - ref).abs().max() < 2e-4 * (1 + ref.abs().max())          # hi+lo weights carry ~16 mantissa bits


@pytest.mark.parametrize("segs,cout,k3,m_pad", [([128, 384], 256, (1, 1, 15), None), ([48, 16], 190, (1, 3, 3), None), ([128], 256, (3, 3, 3), None),
                                                ([64], 128, (1, 5, 1), None), ([48, 16], 190, (1, 3, 3), 192), ([32], 160, (1, 1, 5), 192)])
def test_fragment_order_packing_reproduces_the_convolution(segs, cout, k3, m_pad):
    """pack_conv4 (conv_gemm5.hip: weights in MFMA-fragment order, 16-channel k-steps, taps in sweep order) unpacks to the same
    weight matrix: checked against F.conv3d, with the sweep-axis conventions the engine uses (y sweep: kh / kw swapped; 2-D
    sweep: (ky, kx) flattened into x)."""
    T, H, W = 2, 5, 6
    P = T * H * W
    cin = sum(segs)
    xs = [hash_normal((P, c), 10 + i) for i, c in enumerate(segs)]
    wt = hash_normal((cout, cin, *k3), 20) / math.sqrt(cin * k3[0] * k3[1] * k3[2])
    bs = hash_normal((cout,), 21)
    kt, kh, kw = k3
    sweep = wt
    if kh > 1 and kw > 1:
        sweep = wt.reshape(cout, cin, kt, 1, kh * kw)
    elif kh > 1:
        sweep = wt.transpose(3, 4).contiguous()
    seg_pad = [((c + 15) // 16) * 16 for c in segs]
    packed, bias, meta = pack_conv4(sweep, bs, segs, seg_pad, None, m_pad)            # (m_pad 192: conv_gemm5's three-cout-block layout)
    assert packed.dtype == torch.bfloat16 and meta["M"] == (m_pad or (cout + 127) // 128 * 128) and packed.numel() == 2 * meta["M"] * meta["nk"] * 16
    Wm_ = unpack_conv4_reference(packed, meta["M"], meta["nk"], meta["taps"], meta["cpad"] // 16)      # [M][taps of the packed view * cpad]
    # bring the packed view's tap order back to (kz, ky, kx) of the true kernel
    cp = meta["cpad"]
    Wv = Wm_.reshape(meta["M"], kt, meta["taps"][1], meta["taps"][2], cp)
    if kh > 1 and kw > 1:
        Wv = Wv.reshape(meta["M"], kt, kh, kw, cp)
    elif kh > 1:
        Wv = Wv.transpose(2, 3)
    cols, xp = [], []
    for x, c_ in zip(xs, seg_pad):
        xp.append(F.pad(x, (0, c_ - x.shape[1])))
    xcat = torch.cat(xp, 1).reshape(T, H, W, -1)
    for kz in range(kt):
        for ky in range(kh):
            for kx in range(kw):
                dt, dy, dx = kz - kt // 2, ky - kh // 2, kx - kw // 2
                sh = torch.zeros_like(xcat)
                t0, t1, y0, y1, x0, x1 = max(0, -dt), min(T, T - dt), max(0, -dy), min(H, H - dy), max(0, -dx), min(W, W - dx)
                if t0 < t1 and y0 < y1 and x0 < x1:
                    sh[t0:t1, y0:y1, x0:x1] = xcat[t0 + dt:t1 + dt, y0 + dy:y1 + dy, x0 + dx:x1 + dx]
                cols.append(sh.reshape(P, -1))
    got = (torch.cat(cols, 1) @ Wv.reshape(meta["M"], -1).t() + bias)[:, :cout]
    x5 = torch.cat(xs, 1).reshape(1, T, H, W, cin).permute(0, 4, 1, 2, 3)
    ref = F.conv3d(x5, wt, bs, padding=tuple(k // 2 for k in k3)).permute(0, 2, 3, 4, 1).reshape(P, cout)
    assert (got - ref).abs().max() < 2e-4 * (1 + ref.abs().max())


def test_packing_cout_map_routes_groups_to_aligned_blocks():
    wt, bs = hash_normal((190, 320, 3, 3), 30) / 50, hash_normal((190,), 31)
    rows = list(range(126)) + list(range(128, 192))
    packed, bias, meta = pack_conv2(wt, bs, [320], None, rows, 192)
    full = unpack_conv2_reference(packed, 192, meta["nk"], meta["taps"], meta["cpad"] // BK)
    assert meta["M"] == 192 and (full[126:128] == 0).all() and (bias[126:128] == 0).all()
    assert torch.allclose(bias[128:192], bs[126:]) and torch.allclose(bias[:126], bs[:126])


@pytest.mark.parametrize("c3d", [False, True])
def test_update_block_state_dict_is_the_reference_layout(c3d):
    from ppmstereo_amd.ppmstereo import PPMStereoHotPath
    m = PPMStereoHotPath(use_convex_3d=c3d)
    for tag, attn in (("update_block16", True), ("update_block08", False), ("update_block04", False)):
        sd = getattr(m, tag).state_dict()
        want = Wm.update_block_param_shapes(attn, c3d)
        assert ("mask_3d.0.weight" in sd) == c3d and ("mask_2d.0.weight" in sd) != c3d
        assert list(sd.keys()) == list(want.keys()), "same parameter names in the same (registration) order as the reference module"
        for k, shape in want.items():
            assert tuple(sd[k].shape) == tuple(shape), k
    assert tuple(m.att[0].state_dict()["to_qk.weight"].shape) == (256, 128, 1, 1)
    n = sum(p.numel() for p in m.update_block16.parameters())
    if not c3d:
        assert abs(n - 9.39e6) < 0.01e6                   # update_block16 has 9.39 M parameters (mask_2d variant)


def test_unsupported_configurations_raise():
    from ppmstereo_amd.ppmstereo import PPMStereoHotPath
    from ppmstereo_amd.update import SequenceUpdateBlock3D
    with pytest.raises(NotImplementedError):
        PPMStereoHotPath(use_3d_update_block=False)           # the 2-D block's signatures do not match the call sites (SURVEY hazard 7)
    with pytest.raises(NotImplementedError):
        PPMStereoHotPath(init_flow=True)                      # calls a non-existent update_block04.mask in the reference
    with pytest.raises(NotImplementedError):
        SequenceUpdateBlock3D(hidden_dim=96, cor_planes=36, mask_size=4)
    assert hasattr(SequenceUpdateBlock3D(hidden_dim=128, cor_planes=36, mask_size=4, use_convex_3d=True), "mask_3d")


def test_ops_refuse_cpu_tensors_without_touching_a_gpu():
    from ppmstereo_amd.corr import CorrBlock1D
    with pytest.raises(RuntimeError, match="GPU only"):
        CorrBlock1D(torch.zeros(1, 256, 4, 32), torch.zeros(1, 256, 4, 32))


def test_window_plan_matches_the_oracle_restatement():
    from ppmstereo_amd.ppmstereo import shard_windows, window_plan
    for n, k in ((40, 20), (5, 20), (25, 20), (150, 20), (40, 10), (21, 20), (19, 20)):
        assert window_plan(n, k) == O.window_plan(n, k), (n, k)
    plan = window_plan(150, 20)
    parts = [shard_windows(plan, r, 4) for r in range(4)]
    assert sorted(sum(parts, [])) == sorted(plan) and max(len(p) for p in parts) - min(len(p) for p in parts) <= 1


def test_temporal_encoding_and_scale_match_the_oracle():
    from ppmstereo_amd.engine import softmax_scale, temporal_pe
    from ppmstereo_amd.update import get_temporal_positional_encoding
    for T in (2, 5, 8, 40):
        assert torch.equal(temporal_pe(T, 128), O.temporal_pe(T, 128))
        assert torch.equal(get_temporal_positional_encoding(T, 128, "cpu", is_normalize=True, scale=1.0).reshape(T, 128), O.temporal_pe(T, 128))
    assert torch.isnan(temporal_pe(1, 128)).all()
    assert softmax_scale(128) == O.softmax_scale(128)


# ------------------------------------------------------------------------------------------------ launch heuristics (host code of libppms)
def _desc(T, H, W, M, k3, seg_c, out_vt=False):
    """A conv descriptor good enough for the host-side planners (they never dereference the pointers)."""
    from ppmstereo_amd import _lib as L
    d = L.Conv()
    for i, c in enumerate(seg_c):
        d.seg[i] = L.SP(0x1000, 0x2000, c, c)
    d.nseg, d.w, d.bias = len(seg_c), 0x3000, 0x4000
    d.T, d.H, d.W = T, H, W
    d.kt, d.kh, d.kw = k3
    d.M = d.m_split = M
    d.epi[0].n_valid = M
    if out_vt:
        d.epi[0].out_vt = 0x5000
    return d


def test_k_slicing_plan_for_small_maps():
    """ppms_conv_gemm2_slices: small maps (fewer than ~512 workgroups of 64 couts x 128 px) get their K loop cut into
    2..8 slices that divide kh * chunks; the 1/4 scale, V^T-writing convs and short K loops are left alone."""
    import ctypes as C
    from ppmstereo_amd import _lib as L
    lib = L.load()
    s = lambda *a, **k: lib.ppms_conv_gemm2_slices(C.byref(_desc(*a, **k)))
    assert s(5, 20, 32, 256, (1, 1, 15), [128, 384]) == 8          # GRU z/r pass W at 1/16: 100 workgroups, 16 chunks
    assert s(5, 40, 64, 256, (1, 1, 15), [128, 256]) == 2          # 1/8 with the hoisted inp share: 400 workgroups, 12 chunks
    assert s(5, 80, 128, 256, (1, 1, 15), [128, 256]) == 1         # 1/4 scale: enough workgroups
    assert s(5, 20, 32, 128, (1, 1, 1), [128], out_vt=True) == 1   # to_v writes V^T from the accumulators
    assert s(5, 20, 32, 64, (1, 1, 1), [128]) == 1                 # 4 k-steps in all: nothing to slice
    # short K loops stay unsliced when the in-workgroup K-groups leave <= 10 k-steps each and the grid is one round of >= 512 waves
    assert s(5, 40, 64, 128, (1, 1, 5), [128]) == 1                # GRU (1,1,5) tail at 1/8: 200 workgroups x 2 K-groups, 20 k-steps
    assert s(5, 20, 32, 256, (1, 3, 3), [128]) == 1                # mask head at 1/16: 100 workgroups x 4 K-groups, 36 k-steps
    assert s(5, 20, 32, 64, (1, 3, 3), [128]) > 1                  # 25 workgroups: too few waves without slices
    assert s(5, 20, 32, 768, (1, 1, 1), [384]) > 1                 # 300 workgroups: more than one round unsliced
    n = s(5, 20, 32, 192, (1, 3, 3), [320])                        # final_conv at 1/16: 30 row-steps per temporal tap
    assert n in (2, 3, 5, 6) and 30 % n == 0
    d = _desc(5, 20, 32, 256, (1, 1, 15), [128, 384])
    assert lib.ppms_conv_gemm2_slice_workspace_bytes(C.byref(d), 8) == 8 * 5 * 20 * 32 * 256 * 4
    assert lib.ppms_conv_gemm2_slice_workspace_bytes(C.byref(d), 1) == 0


def test_conv_stream_rating():
    """ppms_conv_stream_applicable: 1 = served and rated faster than the K-sliced LDS-staged kernel + reduce launch (maps of <= 4 096 pixels while
    pixels x K x M <= 4e9; larger maps only for convolutions without spatial taps and for 64-cout convs), 2 = served but the staged kernels win,
    0 = not served (input channels not a multiple of 64, M not a multiple of 64, out_vt)."""
    import ctypes as C
    from ppmstereo_amd import _lib as L
    lib = L.load()
    r = lambda *a, **k: lib.ppms_conv_stream_applicable(C.byref(_desc(*a, **k)))
    # the 1/16 scale of config 2 (3 200 pixels): everything but update_block16's 15-tap z/r conv
    assert r(5, 20, 32, 128, (1, 1, 5), [128, 384]) == 1           # q1
    assert r(5, 20, 32, 256, (1, 5, 1), [128, 384]) == 1           # zr2
    assert r(5, 20, 32, 256, (3, 3, 3), [128]) == 1                # flow head
    assert r(5, 20, 32, 192, (1, 3, 3), [320]) == 1                # final_conv
    assert r(5, 20, 32, 768, (1, 1, 1), [384, 384]) == 1           # the K = 768 Linear layer of the space attention
    assert r(5, 20, 32, 256, (1, 1, 15), [128, 384]) == 2          # 3 200 x 7 680 x 256 = 6.3e9: the staged kernel re-uses its window over 15 taps
    # the 1/8 scale (12 800 pixels): only what the staged kernel has no window re-use or too few tiles for
    assert r(5, 40, 64, 256, (5, 1, 1), [128, 256]) == 1           # temporal GRU pass
    assert r(5, 40, 64, 64, (1, 3, 3), [128]) == 1                 # convf2: 64 couts
    assert r(5, 40, 64, 128, (1, 1, 5), [128, 256]) == 2
    assert r(5, 40, 64, 256, (3, 3, 3), [128]) == 2
    # the 1/4 scale: never
    assert r(5, 80, 128, 128, (5, 1, 1), [128, 256]) == 2
    assert r(5, 80, 128, 64, (1, 3, 3), [128]) == 2
    # config 3's 1/16 scale (18 400 pixels): the tap-less ones
    assert r(5, 46, 80, 384, (1, 1, 1), [768]) == 1 and r(5, 46, 80, 128, (1, 1, 5), [128, 384]) == 2
    # not served
    assert r(5, 20, 32, 128, (1, 1, 1), [96]) == 0                 # K = 96
    assert r(5, 20, 32, 128, (1, 1, 1), [128], out_vt=True) == 0   # to_v writes V^T (gemm1 serves it)
    d = _desc(5, 20, 32, 128, (1, 2, 1), [128])
    assert lib.ppms_conv_stream_applicable(C.byref(d)) == 0        # even kernel extent


def test_large_map_kernel_applicability():
    """ppms_conv_gemm6_applicable rates a descriptor 1 where its 16 x 13-pixel tiles fill >= 85 % of the CU slots of the launch's rounds (config 2's
    and config 3's 1/4 scales), 2 where it serves but fills poorly (the caller keeps conv_gemm5 there), 0 on small maps (fewer tiles than ~CUs: the
    K-sliced / register-streamed kernels fill the chip better) and for what it does not serve; grouped descriptors (groups = 2) only in the shape of the
    GRU's two (1,1,5) tails.  (No GPU call: the rating is host arithmetic; the CU count falls back to 256 without a device.)"""
    import ctypes as C
    from ppmstereo_amd import _lib as L
    lib = L.load()

    def a(T, H, W, M, k3, segs, groups=0, m_split=None):
        d = _desc(T, H, W, M, k3, segs)
        d.groups = groups
        if m_split is not None:
            d.m_split = m_split
        return lib.ppms_conv_gemm6_applicable(C.byref(d))

    assert a(5, 80, 128, 256, (1, 1, 15), [128, 256]) == 1         # x sweep, 250 tiles on 256 CUs
    assert a(5, 80, 128, 256, (1, 5, 1), [128, 256]) == 1          # y sweep
    assert a(5, 80, 128, 256, (3, 3, 3), [128]) == 1               # 2-D sweep
    assert a(5, 80, 128, 256, (5, 1, 1), [128, 256]) == 1          # temporal only: the STREAM form
    assert a(5, 80, 128, 128, (1, 1, 5), [128, 256]) == 1          # M = 128: two cout halves x two pixel halves
    assert a(5, 184, 320, 256, (1, 1, 15), [128, 256]) == 1        # config 3's 1/4 scale
    assert a(5, 92, 160, 256, (1, 1, 15), [128, 256]) == 2         # config 3's 1/8 scale: 390 tiles = 76 % of two rounds
    assert a(5, 40, 64, 256, (1, 1, 15), [128, 256]) == 0          # 1/8 scale of config 2: too few tiles
    assert a(5, 80, 128, 256, (1, 9, 9), [128]) == 0               # kh > 5
    assert a(5, 80, 128, 64, (1, 3, 3), [128]) == 0                # M = 64
    assert a(5, 80, 128, 256, (1, 1, 5), [128, 128], groups=2, m_split=128) == 1      # the grouped z1_2 | r1_2 launch
    assert a(5, 80, 128, 256, (1, 1, 15), [128, 128], groups=2, m_split=128) == 0     # grouped: x sweeps of <= 5 taps only
    assert a(5, 80, 128, 256, (1, 1, 5), [128, 256], groups=2, m_split=128) == 0      # grouped: equal segments
    assert a(5, 80, 128, 256, (1, 1, 5), [128, 128], groups=3, m_split=128) == 0


def test_fnet_state_dict_is_the_reference_layout():
    """ppmstereo_amd.encoder.BasicEncoder exposes the reference's parameter names, shapes and ORDER (extractor.py:349-389, 303-341;
    tools/gen_golden.py additionally loads these weights into the reference module with strict=True)."""
    from ppmstereo_amd import weights as Wm
    from ppmstereo_amd.encoder import BasicEncoder
    m = BasicEncoder(output_dim=256, norm_fn="instance")
    shapes = Wm.fnet_param_shapes()
    sd = m.state_dict()
    assert list(sd.keys()) == list(shapes.keys())
    assert all(tuple(sd[k].shape) == tuple(v) for k, v in shapes.items())
    assert abs(sum(v.numel() for v in sd.values()) - 1.10e6) < 0.01e6      # fnet 1.10 M parameters
    m.load_state_dict(Wm.fnet_weights(), strict=True)


@pytest.mark.parametrize("k,pad,cin,cout,h,w", [(7, 3, 3, 8, 20, 28), (3, 1, 16, 12, 12, 16), (1, 0, 8, 8, 10, 14)])
def test_stride2_conv_as_space_to_depth_conv(k, pad, cin, cout, h, w):
    """The host-side weight re-layout behind the HIP fnet's stride-2 layers: conv(x, w, stride 2, padding p) == a stride-1 'same'
    conv of the re-laid kernel over the 2x2 space-to-depth input (channel (2 dy + dx) * cin + c), exactly (same products, fp32)."""
    import torch.nn.functional as F
    from ppmstereo_amd.encoder import _s2d_weight
    from ppmstereo_amd.weights import hash_normal
    x, wt = hash_normal((2, cin, h, w), 5).double(), hash_normal((cout, cin, k, k), 6).double()
    ref = F.conv2d(x, wt, stride=2, padding=pad)
    xs = x.reshape(2, cin, h // 2, 2, w // 2, 2).permute(0, 3, 5, 1, 2, 4).reshape(2, 4 * cin, h // 2, w // 2)     # (dy, dx, c) major
    w2 = _s2d_weight(wt, pad)
    got = F.conv2d(xs, w2, padding=w2.shape[-1] // 2)
    assert got.shape == ref.shape and (got - ref).abs().max() < 1e-12


def test_library_holds_no_packed_fp32_arithmetic():
    """The hardware condition of docs/LOG_r01_r05.md section 5: on gfx950 a v_pk_add/mul/fma_f32 whose op_sel is [0,1] (low result from src0.lo and
    src1.hi -- a form the compiler picks freely) reads src1.hi as 0 in lanes 48-63 while another wave on the same SIMD issues MFMAs, and the
    engine does run MFMA kernels beside small kernels on two streams.  The library is therefore built without packed fp32 formation; this
    test disassembles every gfx950 code object of the shipped libppms.so and checks that none of these instructions is in it."""
    import importlib.util
    import os

    from ppmstereo_amd import _lib as L

    L.load()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("check_no_packed_fp32", os.path.join(root, "tools", "check_no_packed_fp32.py"))
    chk = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(chk)
    n_obj, n_ins, n_mfma, hits = chk.scan(os.path.join(root, "ppmstereo_amd", "libppms.so"))
    assert n_obj >= 9 and n_ins > 100000 and n_mfma > 500, (n_obj, n_ins, n_mfma)       # the scan really saw the device code
    assert hits == [], hits[:5]


def test_corr_build_workgroup_order_is_a_bijection():
    """corr_build_kernel deals the x1 tiles of an epipolar line to one XCD by remapping its dispatch index (corr.hip): groups of 8 lines x
    ntx tiles, workgroup 8 t + k of a group = tile t of line k; the last nrow % 8 lines keep the plain order.  The same arithmetic here:
    every (line, tile) pair must come out exactly once for any grid."""
    def remap(ntx, nrow):
        seen = set()
        for lin in range(ntx * nrow):
            grp = lin // (8 * ntx)
            inn = lin - grp * 8 * ntx
            full = nrow // 8
            if grp < full:
                row, tx = grp * 8 + (inn & 7), inn >> 3
            else:
                rest = lin - full * 8 * ntx
                row, tx = full * 8 + rest // ntx, rest % ntx
            assert 0 <= row < nrow and 0 <= tx < ntx
            seen.add((row, tx))
        assert len(seen) == ntx * nrow
    for ntx in (1, 2, 3, 4, 5, 8, 10, 40):
        for nrow in (1, 7, 8, 9, 15, 16, 100, 230, 400):
            remap(ntx, nrow)


def test_pack_gemm1_round_trip():
    """pack_gemm1 (gemm1.hip: [M/32][K/16][hi, lo][lane][8], the MFMA A-operand image) unpacks to the same [M][K] matrix as the weights it was
    given: two segments with padding, a cout map, padded rows."""
    from ppmstereo_amd.packing import pack_gemm1, unpack_gemm1_reference
    from ppmstereo_amd.weights import hash_normal
    w = hash_normal((54, 36 + 100, 1, 1), 77)
    packed, b, meta = pack_gemm1(w, hash_normal((54,), 78), [36, 100], [48, 112], None, 64)
    assert meta["M"] == 64 and meta["nk"] == 10 and packed.numel() == 2 * 64 * 160
    full = unpack_gemm1_reference(packed, 64, 10)
    want = torch.zeros(64, 160)
    want[:54, :36] = w[:, :36, 0, 0]
    want[:54, 48:148] = w[:, 36:, 0, 0]
    assert (full - want).abs().max() < 2e-5 * want.abs().max()          # hi + lo of a bf16 split: 16 mantissa bits
    assert b.shape == (64,) and (b[54:] == 0).all()
    rows = list(range(0, 40, 2))
    packed, b, meta = pack_gemm1(hash_normal((20, 64, 1, 1), 79), None, [64], None, rows)
    full = unpack_gemm1_reference(packed, meta["M"], meta["nk"])
    assert meta["M"] == 64 and (full[1::2][:20] == 0).all() and full[0::2][:20].abs().max() > 0


def test_pack_conv6_grouped_interleaves_the_two_convolutions_per_k_step():
    """pack_conv6_grouped (ppms_conv.groups = 2): per k32-step the 8 cout blocks of group 0's pack_conv6 image, then the 8 of group 1's; both unpack
    to their own weights (unpack_conv6_reference)."""
    import torch
    from ppmstereo_amd.packing import pack_conv6, pack_conv6_grouped, unpack_conv6_reference
    from ppmstereo_amd.weights import hash_normal
    k3 = (1, 1, 5)
    wz, wr = hash_normal((128, 128, *k3), 31), hash_normal((128, 128, *k3), 32)
    bz, br = hash_normal((128,), 33), hash_normal((128,), 34)
    packed, bias, meta = pack_conv6_grouped([wz, wr], [bz, br], 128)
    assert meta["M"] == 256 and meta["groups"] == 2 and meta["seg_padded"] == [128, 128] and meta["nk"] == 4 * 5
    assert torch.equal(bias, torch.cat([bz, br]))
    g = packed.reshape(meta["nk"], 16, 2, 64, 8)
    for i, (w, b) in enumerate(((wz, bz), (wr, br))):
        single, _, m1 = pack_conv6(w, b, [128], None, None, 128)
        assert torch.equal(g[:, 8 * i:8 * i + 8].reshape(-1), single)
        back = unpack_conv6_reference(g[:, 8 * i:8 * i + 8].contiguous().reshape(-1), 128, meta["nk"], k3, 4)        # [128][tap * 128 + ci]
        want = w[:, :, 0, 0].permute(0, 2, 1).reshape(128, 5 * 128)
        assert (back - want).abs().max().item() < 1e-4 * want.abs().max().item()


def test_pack_stream_round_trip():
    """pack_stream (conv_stream.hip: [M/32][tap][K/16][hi, lo][lane][8], taps in the natural (kz, ky, kx) order) unpacks to the [M][tap][K]
    matrix of the weights it was given: two segments with padding, a cout map with padded rows, 3-D taps."""
    from ppmstereo_amd.packing import pack_stream, unpack_stream_reference
    from ppmstereo_amd.weights import hash_normal
    w = hash_normal((54, 36 + 10, 3, 1, 5), 77)
    packed, b, meta = pack_stream(w, hash_normal((54,), 78), [36, 10], [48, 16], None, 64)
    assert meta["M"] == 64 and meta["nk"] == 15 * 4 and meta["taps"] == (3, 1, 5) and packed.numel() == 2 * 64 * 15 * 64
    full = unpack_stream_reference(packed, 64, 15, 4).reshape(64, 15, 64)
    want = torch.zeros(64, 15, 64)
    wk = w.permute(0, 2, 3, 4, 1).reshape(54, 15, 46)
    want[:54, :, :36] = wk[:, :, :36]
    want[:54, :, 48:58] = wk[:, :, 36:]
    assert (full - want).abs().max() < 2e-5 * want.abs().max()          # hi + lo of a bf16 split: 16 mantissa bits
    assert b.shape == (64,) and (b[54:] == 0).all()
    rows = list(range(126)) + list(range(128, 192))                      # final_conv's cout map
    packed, b, meta = pack_stream(hash_normal((190, 64, 3, 3), 79), None, [64], None, rows, 192)
    full = unpack_stream_reference(packed, 192, 9, 4)
    assert meta["M"] == 192 and (full[126:128] == 0).all() and full[128:].abs().max() > 0
    with pytest.raises(AssertionError):
        pack_stream(hash_normal((8, 32, 1, 1), 80), None, [32])          # K = 32: not a multiple of 64


@pytest.mark.parametrize("tool,header", [("gen_conv5_asm", "conv5_asm.h"), ("gen_attn_asm", "attn64_asm.h")])
def test_committed_asm_headers_are_the_generators_default_output(tool, header):
    """The hand-scheduled loops are generated files and the ablation scripts (tools/abl_*_phase.sh) rewrite them in place with wrong-results
    settings: the committed header must be exactly what the generator emits with no knob set."""
    import importlib.util
    import os
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    saved = {k: os.environ.pop(k) for k in list(os.environ) if k.startswith(("PPMS_CONV5_", "PPMS_ATTN_"))}
    try:
        spec = importlib.util.spec_from_file_location(f"_{tool}", os.path.join(root, "tools", tool + ".py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        assert mod.gen() == open(os.path.join(root, "ppmstereo_amd", "csrc", header)).read()
    finally:
        os.environ.update(saved)


def test_drop_in_constructor_keeps_the_reference_defaults():
    """ppmstereo.py:45-55: attention_type=None, use_3d_update_block=False, different_update_blocks=False -- a configuration outside the hot
    path, refused loudly; the wrapper's arguments (models/ppm_stereo_model.py:27-33) build the model."""
    import inspect

    from ppmstereo_amd.ppmstereo import PPMStereo
    d = {k: v.default for k, v in inspect.signature(PPMStereo.__init__).parameters.items() if v.default is not inspect.Parameter.empty}
    assert (d["max_disp"], d["mixed_precision"], d["num_frames"], d["attention_type"], d["use_3d_update_block"], d["different_update_blocks"],
            d["use_convex_3d"], d["init_flow"]) == (192, False, 5, None, False, False, False, False)
    with pytest.raises(NotImplementedError, match="use_3d_update_block=True"):
        PPMStereo()
    assert PPMStereo.WRAPPER_CONFIG == dict(mixed_precision=True, num_frames=5, attention_type="self_stereo_temporal_update_time_update_space",
                                            use_3d_update_block=True, different_update_blocks=True)
