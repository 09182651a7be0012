"""Where a conv5 launch spends its time, phase by phase (debug build: PPMS_BUILD_DEFINES=-DPPMS_CONV5_TIMING python -m ppmstereo_amd.build):
wall-clock stamps of wave 0 (and wave 4: the second K-group of M = 128 convs) of every workgroup at: kernel entry, loop start, loop end,
after the K-group reduction, exit.  usage: tools/conv5_phase_probe.py op1,op2,..."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppmstereo_amd import weights as Wm, _lib as L
from ppmstereo_amd.ppmstereo import PPMStereoHotPath
from ppmstereo_amd.weights import hash_normal
dev = torch.device("cuda:0")
T, h, w = 5, 80, 128
m = PPMStereoHotPath().load_hot_path_weights(Wm.hot_path_weights()).to(dev).eval()
eng = m.update_block04.engine(T, h, w, dev)
for t in (eng.X, eng.Hb[0], eng.Hb[1], eng.Hb[2], eng.RH, eng.ZT, eng.RT, eng.FH1, eng.M1, eng.COR256, eng.CF[0], eng.FLO1, eng.VAL):
    t.set_f32(0.3 * hash_normal((t.pixels, t.channels), 1).to(dev))
lib = C.CDLL(L.lib_path())
lib.ppms_debug_conv5_timing.argtypes = [C.c_void_p]
dbg = torch.zeros(1024, 16, dtype=torch.int64, device=dev)
for name in sys.argv[1].split(","):
    op = eng.op[name]
    for _ in range(3):
        op()
    torch.cuda.synchronize()
    lib.ppms_debug_conv5_timing(dbg.data_ptr())
    dbg.zero_()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); op(); b.record()
    torch.cuda.synchronize()
    lib.ppms_debug_conv5_timing(None)
    d = dbg[dbg[:, 0] > 0].double() * 0.01                       # us (100 MHz)
    t0 = d[:, 0].min()
    ph = lambda k, base=0: (d[:, base + k] - t0)
    names = ("entry", "loop start", "loop end", "reduced", "exit")
    print(f"{name}: {len(d)} workgroups, kernel (events) {a.elapsed_time(b) * 1e3:.1f} us; stamps relative to the first workgroup's entry, mean [min..max] us")
    for k, nm in enumerate(names):
        v = ph(k)
        v4 = ph(k, 8)
        print(f"   {nm:10s} wave0 {v.mean():7.1f} [{v.min():6.1f}..{v.max():6.1f}]   wave4 {v4.mean():7.1f} [{v4.min():6.1f}..{v4.max():6.1f}]")
    dur = [(d[:, k + 1] - d[:, k]).mean() for k in range(4)]
    print("   phase means: prologue %.1f, loop %.1f, reduction %.1f, epilogue %.1f us" % tuple(float(x) for x in dur))
    e = [float((d[:, b] - d[:, a]).mean()) for a, b in ((3, 5), (5, 6), (6, 7), (7, 4))]
    print("   inside the epilogue (wave 0): block 0 staged after %.2f us, its first 8-row step %.2f, its other three steps %.2f, the remaining blocks %.2f us" % tuple(e))
