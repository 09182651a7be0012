"""Which convolution kernel generation serves which launch of an iteration, per scale (engine.ConvOp.version: 2 conv_gemm2, 5 conv_gemm5,
6 gemm1, 7 conv_stream, 8 conv_gemm6).  usage: tools/kernel_versions.py [T H W]   (GPU box)"""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ppmstereo_amd import weights as Wm
from ppmstereo_amd.engine import ConvOp
from ppmstereo_amd.ppmstereo import PPMStereoHotPath
T, H, W = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (5, 320, 512)
dev = torch.device("cuda:0")
model = PPMStereoHotPath().load_hot_path_weights(Wm.hot_path_weights()).to(dev).eval()
names = {2: "conv_gemm2", 5: "conv_gemm5", 6: "gemm1", 7: "conv_stream", 8: "conv_gemm6"}
for s, blk in ((16, model.update_block16), (8, model.update_block08), (4, model.update_block04)):
    eng = blk.engine(T, H // s, W // s, dev)
    by = collections.defaultdict(list)
    for k, op in eng.op.items():
        if isinstance(op, ConvOp):
            by[names[op.version] + (" (K-sliced + reduce)" if op.nslice > 1 else "")].append(k)
    print(f"T={T} {H}x{W} scale 1/{s} ({H // s}x{W // s}):")
    for v, ks in sorted(by.items()):
        print(f"   {v:32s} {len(ks):3d}: {' '.join(sorted(ks))}")
