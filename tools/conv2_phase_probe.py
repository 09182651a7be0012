"""Where a conv_gemm2 launch of a small scale spends its time (debug build: PPMS_BUILD_DEFINES=-DPPMS_CONV2_TIMING, exported for the whole
command): wall-clock stamps of wave 0 of every workgroup at entry, first operands requested, first operands in LDS, loop end, reduced, exit.
usage: tools/conv2_phase_probe.py <scale 16|8> [op1,op2,...]"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppmstereo_amd import weights as Wm, _lib as L
from ppmstereo_amd.ppmstereo import PPMStereoHotPath
from ppmstereo_amd.weights import hash_normal
dev = torch.device("cuda:0")
sc = int(sys.argv[1]) if len(sys.argv) > 1 else 16
T, H, W = 5, 320, 512
m = PPMStereoHotPath().load_hot_path_weights(Wm.hot_path_weights()).to(dev).eval()
eng = {16: m.update_block16, 8: m.update_block08, 4: m.update_block04}[sc].engine(T, H // sc, W // sc, dev)
for t in (eng.X, eng.Hb[0], eng.Hb[1], eng.Hb[2], eng.RH, eng.ZT, eng.RT, eng.FH1, eng.M1, eng.COR256, eng.CF[0], eng.FLO1, eng.VAL):
    t.set_f32(0.3 * hash_normal((t.pixels, t.channels), 1).to(dev))
lib = C.CDLL(L.lib_path())
lib.ppms_debug_conv2_timing.argtypes = [C.c_void_p]
dbg = torch.zeros(8192, 8, dtype=torch.int64, device=dev)
names = sys.argv[2].split(",") if len(sys.argv) > 2 else [k for k, v in eng.conv_ops().items() if v.version == 2]
for name in names:
    op = eng.op[name]
    for _ in range(3):
        op()
    torch.cuda.synchronize()
    lib.ppms_debug_conv2_timing(dbg.data_ptr())
    dbg.zero_()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); op(); b.record()
    torch.cuda.synchronize()
    lib.ppms_debug_conv2_timing(None)
    d = dbg[dbg[:, 0] > 0].double().cpu() * 0.01
    t0 = d[:, 0].min()
    ph = [float((d[:, k + 1] - d[:, k]).mean()) for k in range(5)]
    dsc = op.desc
    print(f"{name:10s} M={dsc.M:3d} k=({dsc.kt},{dsc.kh},{dsc.kw}) cin={[dsc.seg[i].c for i in range(dsc.nseg)]} nslice={op.nslice}: {len(d):4d} workgroups, "
          f"op (events, incl. the reduce launch) {a.elapsed_time(b) * 1e3:6.1f} us, conv kernel span {float(d[:, 5].max() - t0):6.1f} us | set-up {ph[0]:.2f}, "
          f"first operands {ph[1]:.2f}, loop {ph[2]:.2f}, K-group sum {ph[3]:.2f}, epilogue {ph[4]:.2f} us; last entry at {float((d[:, 0] - t0).max()):.1f} us")
