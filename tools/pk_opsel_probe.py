"""Runs tools/probe/pk_opsel_probe.hip: a register-only victim (one VOP3P fp32 instruction per round, checked against scalar arithmetic)
alone, beside a library convolution and beside synthetic aggressors (MFMA / LDS reads / global loads / VALU) on another stream.
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -shared -fPIC -o tools/probe/libpk_opsel_probe.so tools/probe/pk_opsel_probe.hip"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppmstereo_amd import weights as Wm
from ppmstereo_amd.ppmstereo import PPMStereoHotPath
from ppmstereo_amd.weights import hash_normal
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "probe", "libpk_opsel_probe.so"))
lib.opsel_victim_launch.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
lib.opsel_aggressor_launch.argtypes = [C.c_int, C.c_void_p, C.c_long, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
dev = torch.device("cuda:0")
T, h, w = 5, 80, 128
m = PPMStereoHotPath().load_hot_path_weights(Wm.hot_path_weights()).to(dev).eval()
eng = m.update_block04.engine(T, h, w, dev)
for t in (eng.X, eng.Hb[0], eng.M1, eng.FH1):
    t.set_f32(0.3 * hash_normal((t.pixels, t.channels), 1).to(dev))
side = torch.cuda.Stream()
big = torch.zeros(1 << 26, device=dev)
sink = torch.zeros(4096 * 256, device=dev)
bad = torch.zeros(8, dtype=torch.int32, device=dev)
first = torch.zeros(16, device=dev)
reps = int(os.environ.get("REPS", "20"))

lib.opsel_holder_launch.argtypes = [C.c_int, C.c_void_p, C.c_long, C.c_void_p]

def holder(regs):
    return lambda: lib.opsel_holder_launch(regs, sink.data_ptr(), 30000, torch.cuda.current_stream().cuda_stream)      # 300 us

lib.opsel_mix_launch.argtypes = [C.c_int, C.c_void_p, C.c_long, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
sink_big = torch.zeros(17 << 22, device=dev)

def mix(flags, iters=1500):
    return lambda: lib.opsel_mix_launch(flags, big.data_ptr(), big.numel(), sink_big.data_ptr(), 2048, iters, torch.cuda.current_stream().cuda_stream)

def synth(kind, iters):
    return lambda: lib.opsel_aggressor_launch(kind, big.data_ptr(), big.numel(), sink.data_ptr(), 2048, iters, torch.cuda.current_stream().cuda_stream)

AGG = (("alone", None), ("conv m1", lambda: eng.op["m1"]()), ("conv zr1_0", lambda: eng.op["zr1_0"]()), ("synthetic MFMA", synth(0, 4000)),
       ("synthetic LDS reads", synth(1, 3000)), ("synthetic global loads", synth(2, 300)), ("synthetic VALU", synth(3, 20000)),
       ("holder 2 x 64 VGPRs", holder(64)), ("holder 2 x 128 VGPRs", holder(128)), ("holder 2 x 192 VGPRs", holder(192)), ("holder 2 x 224 VGPRs", holder(224)),
       ("holder 2 x 232 VGPRs", holder(232)), ("holder 2 x 240 VGPRs", holder(240)))
MIXNAMES = {1: "global loads", 2: "LDS writes", 4: "barriers", 8: "LDS reads", 16: "MFMA", 32: "bf16 split", 64: "global stores"}
for fl in (127, 126, 125, 123, 119, 111, 95, 63, 1, 2, 4, 8, 16, 32, 64, 48, 24, 30, 6):
    AGG = AGG + ((f"mix {fl:3d}: " + "+".join(v for k, v in MIXNAMES.items() if fl & k), mix(fl)),)
if "AGG" in os.environ:
    AGG = tuple(a for a in AGG if any(k in a[0] for k in os.environ["AGG"].split(",")))
OPS = ("v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0]", "v_pk_add_f32 (plain)", "v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0]",
       "v_pk_fma_f32 op_sel:[0,1,0] op_sel_hi:[1,0,1]", "v_pk_add_f32 op_sel:[1,1] op_sel_hi:[0,0]", "v_pk_add_f32 op_sel:[0,0] op_sel_hi:[0,0]",
       "v_pk_mov_b32 op_sel:[1,0]", "v_pk_add_f32 op_sel:[1,0] op_sel_hi:[0,1]", "v_pk_add_f32 op_sel:[0,1] op_sel_hi:[0,1]",
       "v_pk_add_f32 op_sel:[0,0] op_sel_hi:[1,0]", "v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,1]")
only = [int(x) for x in os.environ["OPS"].split(",")] if "OPS" in os.environ else range(len(OPS))
for op in only:
    for name, hv in AGG:
        bad_runs, tot, fst = 0, torch.zeros(4, dtype=torch.int64), None
        for _ in range(reps):
            bad.zero_()
            ev = torch.cuda.Event(); ev.record(); side.wait_event(ev)
            with torch.cuda.stream(side):
                if hv: hv()
            rc = lib.opsel_victim_launch(op, bad.data_ptr(), first.data_ptr(), 4096, 64, torch.cuda.current_stream().cuda_stream)
            assert rc == 0
            torch.cuda.synchronize()
            b = bad[:4].cpu().long()
            if int(b.sum()):
                bad_runs += 1; tot += b
                if fst is None:
                    fst = [round(float(v), 5) for v in first[:9]]
        extra = f"  first: x={fst[0:2]} y={fst[2:4]} got={fst[4:6]} expected={fst[6:8]} lane={int(fst[8])}" if fst else ""
        print(f"{OPS[op]:48s} | {name:24s}: {bad_runs:2d}/{reps} runs wrong, by lane quarter {tot.tolist()}{extra}", flush=True)
