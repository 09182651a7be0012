"""Runs tools/probe/vmcnt_order_probe.hip (do a wave's loads land in the order s_waitcnt vmcnt counts them?) alone, beside a library
convolution on another stream, and beside torch GEMMs.  Build the probe first:
    hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o tools/probe/libvmcnt_order_probe.so tools/probe/vmcnt_order_probe.hip"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppmstereo_amd import weights as Wm
from ppmstereo_amd.ppmstereo import PPMStereoHotPath
from ppmstereo_amd.weights import hash_normal
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe", "libvmcnt_order_probe.so"))
lib.vmcnt_order_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_void_p]
dev = torch.device("cuda:0")
T, h, w = 5, 80, 128
m = PPMStereoHotPath().load_hot_path_weights(Wm.hot_path_weights()).to(dev).eval()
eng = m.update_block04.engine(T, h, w, dev)
for t in (eng.X, eng.Hb[0], eng.M1, eng.FH1):
    t.set_f32(0.3 * hash_normal((t.pixels, t.channels), 1).to(dev))
side = torch.cuda.Stream()
A = torch.randn(8192, 8192, device=dev); B = torch.randn(8192, 8192, device=dev)
Ab, Bb = A.bfloat16(), B.bfloat16()
n, W = 1 << 22, 2048
src = torch.rand(n, device=dev)
out = torch.zeros(n, device=dev)

def victim():
    rc = lib.vmcnt_order_launch(src.data_ptr(), out.data_ptr(), n, W, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    return out

lib.pk_after_wait_launch.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_int, C.c_void_p]
MODE = {"fn": None}

def victim_pk():
    rc = lib.pk_after_wait_launch(src.data_ptr(), out.data_ptr(), n, W, 0, torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    return out

def run(heavy, reps=40):
    bad_runs, bad_elems, quarters = 0, 0, [0, 0, 0, 0]
    for _ in range(reps):
        ev = torch.cuda.Event(); ev.record(); side.wait_event(ev)
        with torch.cuda.stream(side):
            if heavy: heavy()
        o = (MODE['fn'] or victim)()
        torch.cuda.synchronize()
        nz = o.nonzero().flatten()
        if len(nz):
            bad_runs += 1; bad_elems += len(nz)
            q = ((nz % 64) // 16).bincount(minlength=4).tolist()
            quarters = [a + b for a, b in zip(quarters, q)]
    return bad_runs, bad_elems, quarters

CASES = (("alone", None), ("beside conv (m1, conv_gemm5)", lambda: eng.op["m1"]()), ("beside conv (zr1_0)", lambda: eng.op["zr1_0"]()),
         ("beside torch.mm fp32", lambda: torch.mm(A, B)), ("beside torch.mm bf16", lambda: torch.mm(Ab, Bb)))
print("== v_pk_mul_f32 right behind the counted waits")
MODE["fn"] = victim_pk
for name, hv in CASES:
    print(f"{name:32s}: runs with a wrong packed product / 40, elements, by lane quarter: {run(hv)}")
MODE["fn"] = None
print("== plain v_mov behind vmcnt(1)")
for name, hv in (("alone", None), ("beside conv (m1, conv_gemm5)", lambda: eng.op["m1"]()), ("beside conv (zr1_0)", lambda: eng.op["zr1_0"]()),
                 ("beside torch.mm fp32", lambda: torch.mm(A, B)), ("beside torch.mm bf16", lambda: torch.mm(Ab, Bb))):
    print(f"{name:32s}: runs with an early-read mismatch / 40, elements, by lane quarter: {run(hv)}")
