import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppmstereo_amd import weights as Wm
from ppmstereo_amd.corr import CorrBlock1D
from ppmstereo_amd.ppmstereo import PPMStereoHotPath
from ppmstereo_amd.synth import synth_scale_inputs
dev = "cuda:0"
m = PPMStereoHotPath().load_hot_path_weights(Wm.hot_path_weights()).to(dev).eval()
d = synth_scale_inputs(1, 8, 32, seed=81)
g = lambda x: x.to(dev)
blk = m.update_block04
eng = blk.engine(1, 8, 32, torch.device(dev))
eng.set_inp(g(d["inp"])); eng.set_net(g(d["net"])); eng.set_flow(g(d["flow"])); eng.set_mhs(g(d["mhs"]))
eng.begin(CorrBlock1D(g(d["fmap1"]), g(d["fmap2"])).levels, m.att[2].packed(torch.device(dev)))
nan = lambda t: int(torch.isnan(t.float()).sum().item())
print("PE nan", nan(eng.PE), "QB nan", nan(eng.QB), "SIM", eng.SIM.cpu().tolist())
eng.lookup(); eng.motion_and_value(); eng.uncertainty(); eng.pick()
print("SEL", eng.SEL.cpu().tolist(), "SHAT", eng.SHAT.cpu().tolist(), "SCORE", eng.SCORE.cpu().tolist())
raw = torch.zeros(1, 256, 128, dtype=torch.bfloat16, device=dev)
eng.attend(raw)
print("KB nan", nan(eng.KB), "raw nan", nan(raw), "X nan per part", [nan(eng.X.to_f32(c, 128)) for c in (0, 128, 256)])
eng.update()
print("H nan", [nan(h.to_f32()) for h in eng.Hb], "Z", nan(eng.Z), "DFLOW", nan(eng.DFLOW), "FLOW", nan(eng.FLOW), "MASK", nan(eng.MASK))
fo = eng.upsample()
print("flow_out nan", nan(fo))
