"""Runs tools/probe/bilinear_pk_probe.hip (the library's bilinear resize compiled WITH packed fp32 ops, plain / instrumented / control)
alone and beside a library convolution on another stream.  Build the probe first:
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -shared -fPIC -o tools/probe/libbilinear_pk_probe.so tools/probe/bilinear_pk_probe.hip"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppmstereo_amd import weights as Wm
from ppmstereo_amd.ppmstereo import PPMStereoHotPath
from ppmstereo_amd.weights import hash_normal
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe", "libbilinear_pk_probe.so"))
lib.bilinear_probe_launch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
dev = torch.device("cuda:0")
T, h, w = 5, 80, 128
m = PPMStereoHotPath().load_hot_path_weights(Wm.hot_path_weights()).to(dev).eval()
eng = m.update_block04.engine(T, h, w, dev)
for t in (eng.X, eng.Hb[0], eng.M1, eng.FH1):
    t.set_f32(0.3 * hash_normal((t.pixels, t.channels), 1).to(dev))
side = torch.cuda.Stream()
OH, OW = 4 * h, 4 * w
n = T * OH * OW
src = torch.ones(T, h, w, device=dev)
dst = torch.zeros(n, device=dev)
taps = torch.zeros(n, 4, device=dev)
reps = int(os.environ.get("REPS", "40"))

lib.bilinear_asm_launch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]

lib.bilinear_bisect_launch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]

def launch(variant):
    st = torch.cuda.current_stream().cuda_stream
    if variant >= 50:
        return lib.bilinear_bisect_launch(variant, src.data_ptr(), dst.data_ptr(), taps.data_ptr(), T, h, w, OH, OW, st)
    if variant >= 10:
        return lib.bilinear_asm_launch(variant, src.data_ptr(), dst.data_ptr(), T, h, w, OH, OW, st)
    return lib.bilinear_probe_launch(variant, src.data_ptr(), dst.data_ptr(), taps.data_ptr(), T, h, w, OH, OW, st)

def run(variant, heavy):
    bad_runs, bad_elems, quarters, bad_taps, first = 0, 0, [0, 0, 0, 0], 0, None
    for _ in range(reps):
        dst.zero_(); taps.fill_(1.0 if variant < 50 else 0.0)
        ev = torch.cuda.Event(); ev.record(); side.wait_event(ev)
        with torch.cuda.stream(side):
            if heavy: heavy()
        rc = launch(variant)
        assert rc == 0
        torch.cuda.synchronize()
        nz = (dst != 1.0).nonzero().flatten()
        if len(nz):
            bad_runs += 1; bad_elems += len(nz)
            q = ((nz % 64) // 16).bincount(minlength=4).tolist()
            quarters = [a + b for a, b in zip(quarters, q)]
            bad_taps += int((taps[nz] != 1.0).any(dim=1).sum()) if variant < 50 else 0
            if first is None:
                first = (int(nz[0]), [round(float(x), 4) for x in dst[nz[:3]]], taps[nz[0]].tolist())
                if variant == 50:      # products (X.lo, X.hi, Y.lo, Y.hi) of a few bad elements next to what they should be: (hy, ly, hy, ly)
                    oy = (nz[:6] // OW) % OH
                    fy = ((oy.float() + 0.5) * (h / OH) - 0.5).clamp(min=0)
                    lyv = fy - fy.floor()
                    first = first + ("products", [[round(float(v), 4) for v in taps[i]] for i in nz[:6]], "expected hy/ly", [(round(1 - float(l), 4), round(float(l), 4)) for l in lyv])
    return dict(bad_runs=bad_runs, bad_elems=bad_elems, lane_quarters=quarters, elems_with_a_wrong_stored_tap=bad_taps, first=first)

VARIANTS = ((0, "as in the library"), (1, "+ taps stored"), (2, "control: packed formation defeated"),
            (10, "asm: v_pk_mul right behind the waits"), (11, "asm: + s_nop 0"), (12, "asm: + s_nop 1"), (14, "asm: + s_nop 3"), (18, "asm: + s_nop 7"),
            (20, "asm: two v_mul_f32 instead of each v_pk_mul"),
            (30, "asm: the compiler's whole tail (pk_mul x2, swizzled pk_add, pk_mul, add; s_nop 0 after each packed op)"),
            (31, "asm: whole tail, the swizzled pk_add replaced by two v_add_f32"), (32, "asm: whole tail with s_nop 3"), (33, "asm: whole tail without nops"),
            (34, "asm: whole tail, s_nop 7 between vmcnt(0) and its multiply"),
            (40, "asm: variant 0's stream register for register (v2-v7, v12-v15)"), (41, "asm: the same stream 32 registers higher"),
            (50, "bisect: sentinel 2.0 in the load destinations, the four products dumped"), (51, "bisect: s_nop 7 behind both waits"),
            (52, "bisect: multiplies of the loaded pairs scalar"), (53, "bisect: swizzled add scalar"), (54, "bisect: last multiply scalar"),
            (55, "bisect: all scalar"))
only = [int(x) for x in os.environ["VARIANTS"].split(",")] if "VARIANTS" in os.environ else None
for variant, what in VARIANTS:
    if only is not None and variant not in only:
        continue
    for name, hv in (("alone", None), ("beside conv m1", lambda: eng.op["m1"]())):
        print(f"variant {variant} ({what}), {name}: {run(variant, hv)}", flush=True)
