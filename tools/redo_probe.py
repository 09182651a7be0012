"""How many 256-query tiles of the rescale-free attention kernel go to its fix-up pass (a score more than 2^16 -- fp16 P~ -- above the query's softmax
reference) on the synthetic inputs of the BASELINE configurations?  usage: tools/redo_probe.py [T H W iters]   (GPU box)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppmstereo_amd import weights as Wm
from ppmstereo_amd.ppmstereo import PPMStereoHotPath
from ppmstereo_amd.synth import synth_cascade_feats
T, H, W, iters = (int(x) for x in sys.argv[1:5]) if len(sys.argv) > 4 else (5, 320, 512, 10)
dev = torch.device("cuda:0")
model = PPMStereoHotPath().load_hot_path_weights(Wm.hot_path_weights()).to(dev).eval()
feats = {k: v.to(dev) for k, v in synth_cascade_feats(T, H, W).items()}
model.cascade(feats, iters, T, test_mode=True)
torch.cuda.synchronize()
for s, blk in ((16, model.update_block16), (8, model.update_block08), (4, model.update_block04)):
    eng = blk.engine(T, H // s, W // s, dev)
    print(f"T={T} {H}x{W} iters={iters} scale 1/{s}: (tiles, flagged) of the last attention call = {eng.attn_redo_count()}")
