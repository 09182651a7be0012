"""Runs ONE conv op of the 1/4-scale engine (BASELINE config 2: T=5, 80x128) a few times: target for rocprofv3 --pmc passes.
    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE \\
        --output-format csv -d gpurun_out/pmc -- /usr/bin/python3 tools/conv_pmc_probe.py zr1_0 5
(put the interpreter binary itself after `--`: no env / bash / shebang hop under the profiler)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppmstereo_amd import weights as Wm
from ppmstereo_amd.ppmstereo import PPMStereoHotPath
from ppmstereo_amd.synth import synth_cascade_feats

name = sys.argv[1] if len(sys.argv) > 1 else "zr1_0"
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
dev = torch.device("cuda:0")
T, H, W = 5, 320, 512
model = PPMStereoHotPath().load_hot_path_weights(Wm.hot_path_weights()).to(dev).eval()
feats = {k: v.to(dev) for k, v in synth_cascade_feats(T, H, W).items()}
model.cascade(feats, 2, T)
eng = model.update_block04.engine(T, H // 4, W // 4, dev)
torch.cuda.synchronize()
for _ in range(reps):
    eng.op[name]()
torch.cuda.synchronize()
print("done", name, "version", eng.op[name].version)
