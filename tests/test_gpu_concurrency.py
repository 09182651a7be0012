"""GPU robustness: the small kernels must give bit-identical results whether or not another stream keeps the CUs busy with a
conv kernel (the engine runs independent branches of an iteration on two HIP streams).  Regression test for the packed-fp32
issue described in ppmstereo_amd/build.py: built with v_pk_*_f32, ppms_bilinear dropped one tap in lanes 48-63 of some waves
under exactly this load.  The library is now built without packed fp32 code generation in every source; this test keeps
watch (200 repetitions per small kernel: the original failure showed in a few percent of the launches)."""
import pytest
import torch

from ppmstereo_amd import weights as Wm
from ppmstereo_amd.weights import hash_normal

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def eng():
    assert torch.cuda.is_available()
    from ppmstereo_amd.ppmstereo import PPMStereoHotPath
    m = PPMStereoHotPath().load_hot_path_weights(Wm.hot_path_weights()).to(DEV).eval()
    e = m.update_block04.engine(5, 80, 128, torch.device(DEV))
    for t in (e.X, e.Hb[0], e.M1, e.FH1):
        t.set_f32(0.3 * hash_normal((t.pixels, t.channels), 1).to(DEV))
    return e


def _mismatches(eng, fn, heavy=("m1", "m2"), n=200):
    side = torch.cuda.Stream()
    ref = fn().clone()
    torch.cuda.synchronize()
    bad = 0
    for _ in range(n):
        ev = torch.cuda.Event()
        ev.record()
        side.wait_event(ev)
        with torch.cuda.stream(side):
            for k in heavy:
                eng.op[k]()
        out = fn()
        torch.cuda.synchronize()
        bad += int(not torch.equal(out, ref))
    return bad


def test_small_kernels_are_unaffected_by_a_concurrent_conv(eng):
    from ppmstereo_amd import _lib as L
    from ppmstereo_amd.engine import bilinear
    T, h, w = eng.T, eng.h, eng.w
    ones = torch.ones(T, 1, h, w, device=DEV)
    rnd = torch.sigmoid(hash_normal((T, 1, h, w), 7)).to(DEV)
    flow = hash_normal((eng.P, 2), 8).to(DEV)
    mask = hash_normal((eng.P, 144), 9).to(DEV)
    lib = L.load()

    def cvx():
        out = torch.empty(T, 2, 4 * h, 4 * w, device=DEV)
        L.check(lib.ppms_convex_upsample(flow.data_ptr(), mask.data_ptr(), 144, out.data_ptr(), T, h, w, L.stream_ptr()))
        return out

    def to_nchw():
        out = torch.empty(T, 144, h, w, device=DEV)
        L.check(lib.ppms_nhwc_to_nchw(mask.data_ptr(), 144, out.data_ptr(), T, 144, h * w, L.stream_ptr()))
        return out

    def dw():
        (_, _, _), (w7, b7, _) = eng.pk.dw
        L.check(lib.ppms_dwconv_gelu(eng.X.view(0, 40), eng.C1.view(0, 40), w7.data_ptr(), b7.data_ptr(), 7, T, h, w, L.stream_ptr()))
        return eng.C1.to_f32()

    def ln():
        x = hash_normal((eng.P, 384), 11).to(DEV)
        wgt, bias = torch.ones(384, device=DEV), torch.zeros(384, device=DEV)
        out = L.SPTensor(eng.P, 384, DEV)
        L.check(lib.ppms_layernorm(x.data_ptr(), 384, wgt.data_ptr(), bias.data_ptr(), L.SP(None, None, 0, 0), out.view(), eng.P, 384, L.stream_ptr()))
        return out.to_f32()

    # the memory attention kernels (hand-scheduled loop; scalar fp32 ops): same stress
    n = eng.n
    qb = hash_normal((T, n, 128), 950).to(torch.bfloat16).to(DEV)
    kb = hash_normal((T, 5, n, 128), 951).to(torch.bfloat16).to(DEV)
    vt = hash_normal((T, 128, n), 952).to(torch.bfloat16).to(DEV)
    sel = torch.arange(5, dtype=torch.int32)[None].expand(T, 5).contiguous().to(DEV)
    Xa = L.SPTensor(T * n, 256, DEV)
    beta = torch.tensor([0.5], device=DEV)
    raw = torch.zeros(T, n, 128, dtype=torch.bfloat16, device=DEV)
    ws = torch.empty(int(lib.ppms_mem_attn_workspace_bytes(T, 5, n)), dtype=torch.uint8, device=DEV)

    def attn(split):
        def f():
            L.check(lib.ppms_mem_attn(qb.data_ptr(), kb.data_ptr(), vt.data_ptr(), sel.data_ptr(), 5, 0.0522, beta.data_ptr(), Xa.view(0, 128),
                                      Xa.view(128, 128), raw.data_ptr(), T, n, ws.data_ptr() if split else None, 0, L.ATTN_P_BF16, L.stream_ptr()))
            return raw.float()
        return f

    got = bilinear(ones, (4 * h, 4 * w), False)
    assert (got == 1.0).all(), "bilinear weights must sum to one"
    cases = {"bilinear(ones)": lambda: bilinear(ones, (4 * h, 4 * w), False), "bilinear(random)": lambda: bilinear(rnd, (4 * h, 4 * w), False),
             "bilinear(align_corners)": lambda: bilinear(rnd, (4 * h, 4 * w), True), "convex_upsample": cvx, "nhwc_to_nchw": to_nchw,
             "dwconv_gelu": dw, "layernorm": ln, "conv q1 (other conv concurrent)": lambda: (eng.op["q1"](), eng.Hb[1].to_f32())[1]}
    # the register-streamed small-map conv (conv_stream.hip: four K-groups summed through LDS in wave order) on a 1/16-scale engine, under the
    # 1/4 scale's convs on the other stream
    from ppmstereo_amd.ppmstereo import PPMStereoHotPath
    e16 = PPMStereoHotPath().load_hot_path_weights(Wm.hot_path_weights()).to(DEV).eval().update_block16.engine(5, 20, 32, torch.device(DEV))
    for t in (e16.X, e16.XA, e16.Hb[0], e16.RH):
        t.set_f32(0.3 * hash_normal((t.pixels, t.channels), 2).to(DEV))
    assert e16.op["q1"].version == 7 and e16.op["fh1"].version == 7
    cases["conv_stream q1 (1/16 scale)"] = lambda: (e16.op["q1"](), e16.Hb[1].to_f32())[1]
    cases["conv_stream fh1 (1/16 scale, 64-pixel tiles)"] = lambda: (e16.op["fh1"](), e16.FH1.to_f32())[1]
    bad = {name: _mismatches(eng, fn) for name, fn in cases.items()}
    bad["mem_attn 64-query + combine"] = _mismatches(eng, attn(True), heavy=("zr1_0", "m1", "zr2"), n=30)
    bad["mem_attn 32-query fused"] = _mismatches(eng, attn(False), heavy=("zr1_0", "m1", "zr2"), n=12)
    assert all(v == 0 for v in bad.values()), f"results change under a concurrent conv kernel: {bad}"


def test_conv_gemm6_back_to_back_launches_are_bit_identical():
    """tools/conv6_stress.py: every mode conv_gemm6 serves, launched 30 times back to back (no synchronisation in between) at BASELINE config 3's 1/4-scale
    geometry -- six workgroups per CU in sequence, so a workgroup starts on LDS that still holds its predecessor's fp32 staging data and a window
    row that was not (or wrongly) filled by its LDS-DMA piece shows as a changed output.  Regression test for the LDS-DMA spacing rule of
    conv_gemm6.hip (two pieces issued back to back left rows stale: an LDS-DMA reads M0 / its address register after it has issued)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tools", "conv6_stress.py"), "30"], capture_output=True, text=True, cwd=root, timeout=900)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [ln for ln in r.stdout.splitlines() if "launches:" in ln]
    assert len(lines) >= 8, r.stdout[-2000:]
    for ln in lines:
        assert " 0 differ from the first" in ln and " 0 with non-finite" in ln, ln
