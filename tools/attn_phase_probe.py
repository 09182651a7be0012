"""Where a memory-attention workgroup spends its time (debug build: PPMS_BUILD_DEFINES=-DPPMS_ATTN_TIMING, exported for the whole command):
wall-clock stamps of wave 0 of every workgroup of mem_attn64_kernel at entry, loop start, loop end, exit.  BASELINE config 2, 1/4 scale."""
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
for t in (eng.X, eng.Hb[0], eng.VAL):
    t.set_f32(0.3 * hash_normal((t.pixels, t.channels), 1).to(dev))
lib = C.CDLL(L.lib_path())
lib.ppms_debug_attn_timing.argtypes = [C.c_void_p]
nwg = (eng.n // 256) * T * 5
dbg = torch.zeros(nwg, 4, dtype=torch.int64, device=dev)
for _ in range(3):
    eng.attend()
torch.cuda.synchronize()
lib.ppms_debug_attn_timing(dbg.data_ptr())
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record(); eng.attend(); b.record()
torch.cuda.synchronize()
lib.ppms_debug_attn_timing(None)
d = dbg.double().cpu() * 0.01
t0 = d[:, 0].min()
print(f"{nwg} workgroups; attend() incl. prep_k / redo / combine (events): {a.elapsed_time(b) * 1e3:.1f} us; kernel span by stamps {float(d[:, 3].max() - t0):.1f} us")
pro, loop, epi = d[:, 1] - d[:, 0], d[:, 2] - d[:, 1], d[:, 3] - d[:, 2]
print(f"per workgroup: prologue {pro.mean():.2f} [{pro.min():.2f}..{pro.max():.2f}] us, loop {loop.mean():.1f} [{loop.min():.1f}..{loop.max():.1f}] us, "
      f"epilogue {epi.mean():.2f} [{epi.min():.2f}..{epi.max():.2f}] us")
start = (d[:, 0] - t0).sort().values
print("workgroup entry times (us): " + ", ".join(f"{float(start[int(q * (nwg - 1))]):.0f}" for q in (0, 0.25, 0.26, 0.5, 0.51, 0.75, 0.76, 1.0)))
end = (d[:, 3] - t0)
print(f"last exits: {sorted(end.tolist())[-5:]}")
