"""Stress: a conv_gemm6 launch repeated back to back (no synchronisation between the launches) at a geometry with several workgroups per CU in
sequence (a workgroup then starts on LDS that still holds its predecessor's fp32 staging data: a window row that has not landed when it is read
shows as garbage), every output compared bit for bit with the first.  usage: tools/conv6_stress.py [reps]"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import ctypes as C
import torch
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import ab_switches  # noqa: F401  (PPMS_LIB: another build of the library)
from ppmstereo_amd import _lib as L
from ppmstereo_amd.engine import ConvOp, epilogue
from ppmstereo_amd.packing import pack_conv4, pack_conv6
from ppmstereo_amd.weights import hash_normal
DEV = "cuda:0"
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
VER = int(os.environ.get("VERSION", "8"))          # 8: conv_gemm6 (default), 5: conv_gemm5 (the same stress on the older kernel)
T, H, W = (int(x) for x in os.environ.get("PROBE_SHAPE", "5,184,320").split(","))
P = T * H * W
L.load()
cases = [("x5_m128", [128], 128, (1, 1, 5)), ("3x3_m256", [128], 256, (1, 3, 3)), ("y5_m128", [128, 256], 128, (1, 5, 1)), ("x15_m256", [128, 256], 256, (1, 1, 15)),
         ("3x3_m192", [320], 190, (1, 3, 3)), ("3x3x3_m256", [128], 256, (3, 3, 3)), ("x5_m256_two_segs", [128, 256], 256, (1, 1, 5)), ("3x3_m128", [128, 128], 128, (1, 3, 3))]
if os.environ.get("GEMM", "1") != "0":    # convolutions without spatial taps: the STREAM form (one k32-step per window, three window buffers)
    cases += [("1x1_m128_K128", [128], 128, (1, 1, 1)), ("1x1_m128_K256", [256], 128, (1, 1, 1)), ("t5_m128", [128], 128, (5, 1, 1)), ("1x1_m192_K128", [128], 192, (1, 1, 1)),
              ("1x1_m256_K256", [256], 256, (1, 1, 1))]
only = os.environ.get("CASES")
for name, segs, cout, k3 in cases:
    if only and name not in only.split(","):
        continue
    xs = [L.SPTensor(P, c, DEV) for c in segs]
    for i, t in enumerate(xs):
        t.set_f32(hash_normal((P, t.channels), 100 + i).to(DEV))
    cin = sum(segs)
    wt = hash_normal((cout, cin, *k3), 200) / math.sqrt(cin * k3[0] * k3[1] * k3[2])
    w5 = wt
    if k3[1] > 1 and k3[2] > 1:
        w5 = wt.reshape(cout, cin, k3[0], 1, k3[1] * k3[2])
    elif k3[1] > 1:
        w5 = wt.transpose(3, 4).contiguous()
    packed, b, meta = (pack_conv6 if VER == 8 else pack_conv4)(w5.to(DEV), (hash_normal((cout,), 201) * 0.1).to(DEV), segs, segs, None, 128 if cout <= 128 else 192 if cout <= 192 else 256)
    outs = [L.SPTensor(P, meta['M'], DEV) for _ in range(2)]
    ops = []
    for o in outs:
        d = L.Conv()
        for i, t in enumerate(xs):
            d.seg[i] = t.view()
        d.nseg, d.w, d.bias = len(xs), packed.data_ptr(), b.data_ptr()
        d.T, d.H, d.W = T, H, W
        d.kt, d.kh, d.kw = k3
        d.M = d.m_split = meta["M"]
        d.epi[0] = epilogue(act=L.ACT_RELU, n_valid=cout, out_sp=o.view())
        ops.append(ConvOp(d, [packed, b, o] + xs, VER))
    ops[0]()
    torch.cuda.synchronize()
    ref = outs[0].data.clone()
    bad = nan = 0
    flags = []
    for r in range(reps):
        ops[1]()
        flags.append(torch.stack([(outs[1].data.view(torch.int16) != ref.view(torch.int16)).sum(), (~torch.isfinite(outs[1].data.float())).sum()]))
        outs[1].data.zero_()
    torch.cuda.synchronize()
    if os.environ.get("WHERE"):          # against torch: which launches are right?
        import torch.nn.functional as F
        xcat = torch.cat([t.to_f32() for t in xs], 1)
        bias = (hash_normal((cout,), 201) * 0.1).to(DEV)
        x5 = xcat.reshape(1, T, H, W, cin).permute(0, 4, 1, 2, 3)
        want = torch.relu(F.conv3d(x5, wt.to(DEV), bias, padding=tuple(k // 2 for k in k3))).permute(0, 2, 3, 4, 1).reshape(P, cout)
        r32 = ref[0].float() + ref[1].float()
        d0 = (r32 - want).abs().nan_to_num(1e30)
        ops[1]()
        torch.cuda.synchronize()
        o32 = outs[1].data[0].float() + outs[1].data[1].float()
        d1 = (o32 - want).abs().nan_to_num(1e30)
        print(f"   vs torch: first launch {int((d0 > 1e-3).any(1).sum())} wrong pixels (max diff {float(d0.max()):.3g}), another launch {int((d1 > 1e-3).any(1).sum())} wrong pixels (max {float(d1.max()):.3g})")
    if os.environ.get("WHERE"):            # where does the LAST launch differ?
        ops[1]()
        torch.cuda.synchronize()
        d = (outs[1].data.view(torch.int16) != ref.view(torch.int16)).any(0)          # (P, cout)
        idx = d.nonzero()
        if len(idx):
            px = idx[:, 0].unique()
            tiles = set()
            for p_ in px.tolist():
                t_, y_, x_ = p_ // (H * W), p_ % (H * W) // W, p_ % W
                tiles.add((x_ % 13, y_ % 16))
            print("   differing pixels", len(px), "(x % 13 = block n, y % 16 = li) sample:", sorted(tiles)[:24], "couts:", sorted(set((idx[:, 1] // 16).tolist())))
    f = torch.stack(flags).cpu()
    print(f"{name:16s} {reps} launches: {int((f[:, 0] > 0).sum())} differ from the first ({int(f[:, 0].sum())} elements), {int((f[:, 1] > 0).sum())} with non-finite values")
