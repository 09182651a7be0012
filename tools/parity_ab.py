"""Parity table of the HIP cascade against the REFERENCE's fixtures (tests/golden/cascade_it10.npz, cascade_it20.npz: PPMStereo.forward(test_mode=False),
ppmstereo.py:601-791, all predictions) for both formats of the attention's P~ (TUNING["attn_p"]: fp16 / bf16), on one box in one process.
    python tools/parity_ab.py > profiles/rNN_parity_ab.txt
Also counts the 256-query tiles the rescale-free 64-query kernel handed to its fix-up pass (scores above its softmax reference by more than the
format carries: 2^16 for fp16 P~, 2^60 for bf16) in the last attention call of every scale."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch

from golden_util import Golden
from oracle import ppm_oracle as O          # (input generation only: pre_loop_glue of the hash features, as the fixture generator does)
from ppmstereo_amd import engine as E
from ppmstereo_amd import weights as Wm
from ppmstereo_amd.ppmstereo import PPMStereoHotPath
from ppmstereo_amd.weights import hash_normal

DEV = "cuda:0"


def inputs():
    T, H, Wd = 5, 64, 256
    fm1 = hash_normal((T, 256, H // 4, Wd // 4), 171)
    fm2 = 0.8 * torch.roll(fm1, shifts=-3, dims=3) + 0.6 * hash_normal((T, 256, H // 4, Wd // 4), 172)
    ctx = [hash_normal((T, 256, H // s, Wd // s), 173 + i) for i, s in enumerate((4, 8, 16))]
    return T, O.pre_loop_glue(fm1, fm2, *ctx)


def run(fmt, iters, name):
    E.TUNING["attn_p"] = fmt
    model = PPMStereoHotPath().load_hot_path_weights(Wm.hot_path_weights()).to(DEV).eval()
    gd = Golden(name)
    T, feats = inputs()
    preds, uncs = [], []
    disp, unc = model.cascade({k: v.to(DEV) for k, v in feats.items()}, iters, T, preds, uncs)
    P = torch.stack(preds).float().cpu().numpy()
    k, step = gd.keys["predictions"]
    got, ref = P.reshape(-1)[::step], gd.raw("predictions")
    which = np.arange(0, P.size, step) // P[0].size
    epe = [float(np.abs(got - ref)[which == i].mean()) for i in range(len(preds))]
    mx = [float(np.abs(got - ref)[which == i].max()) for i in range(len(preds))]
    k, step = gd.keys["disparity"]
    e = np.abs(disp[None].float().cpu().numpy().reshape(-1)[::step] - gd.raw("disparity"))
    redo = {}
    for tag in ("update_block16", "update_block08", "update_block04"):
        for eng in getattr(model, tag)._engines.values():
            redo[tag[-2:]] = eng.attn_redo_count()
    return epe, mx, float(e.mean()), float(e.max()), redo


if __name__ == "__main__":
    assert torch.cuda.is_available()
    for iters, name in ((10, "cascade_it10"), (20, "cascade_it20")):
        res = {fmt: run(fmt, iters, name) for fmt in ("fp16", "bf16")}
        n = len(res["fp16"][0])
        print(f"== {name}: HIP cascade vs the reference's fixture, mean |d disparity| (px) per prediction [max in brackets]; budget 1e-3 px")
        print(f"{'prediction':>10}  {'scale':>5}  {'fp16 P~':>22}  {'bf16 P~':>22}")
        b = (n // 4, n // 2)
        for i in range(n):
            sc = "1/16" if i < b[0] else "1/8" if i < b[1] else "1/4"
            print(f"{i:>10}  {sc:>5}  {res['fp16'][0][i]:.3e} [{res['fp16'][1][i]:.2e}]  {res['bf16'][0][i]:.3e} [{res['bf16'][1][i]:.2e}]")
        for fmt in ("fp16", "bf16"):
            epe, mx, fe, fm, redo = res[fmt]
            print(f"{name} {fmt} P~: final disparity EPE {fe:.3e} px (max {fm:.2e}); worst prediction EPE {max(epe):.3e} px; fix-up tiles in the last call per scale: {redo}")
        print()
