"""Where a conv6 launch spends its time (debug build: PPMS_BUILD_DEFINES=-DPPMS_CONV6_TIMING python -m ppmstereo_amd.build): wall-clock and
cycle-counter stamps of wave 0 of every workgroup at kernel entry, loop start, loop end, exit -> phase times and the clock held in the K loop.
usage: tools/conv6_phase_probe.py op1,op2,..."""
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
lib.ppms_debug_conv6_timing.argtypes = [C.c_void_p]
dbg = torch.zeros(4096, 16, dtype=torch.int64, device=dev)
for name in sys.argv[1].split(","):
    op = eng.op[name]
    assert op.version == 8, (name, op.version)
    for _ in range(20):                      # warm: the clock under sustained load
        op()
    torch.cuda.synchronize()
    lib.ppms_debug_conv6_timing(dbg.data_ptr())
    dbg.zero_()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); op(); b.record()
    torch.cuda.synchronize()
    lib.ppms_debug_conv6_timing(None)
    d = dbg[dbg[:, 0] > 0].double()
    us = d[:, :4] * 0.01
    t0 = us[:, 0].min()
    cyc = d[:, 8:12]
    dur = [(us[:, k + 1] - us[:, k]) for k in range(3)]
    loop_cyc = cyc[:, 2] - cyc[:, 1]
    ghz = (loop_cyc / (dur[1] * 1e3)).median()
    print(f"{name}: {len(d)} workgroups, kernel (events) {a.elapsed_time(b) * 1e3:.1f} us; entry spread {float((us[:, 0] - t0).max()):.1f} us; "
          f"prologue {float(dur[0].mean()):.1f}, loop {float(dur[1].mean()):.1f} [{float(dur[1].min()):.1f}..{float(dur[1].max()):.1f}], epilogue {float(dur[2].mean()):.1f} us; "
          f"loop cycles {float(loop_cyc.mean()):.0f}, clock in the loop {float(ghz):.2f} GHz")
