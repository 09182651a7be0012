"""Where does the loop lose accuracy?  One iteration of forward_update_block per scale on the cascade_it10 inputs, every stage output of
the engine against the oracle's trace of the same iteration (errors relative to the rms of the oracle's tensor).  GPU box.

    python tools/stage_error_probe.py
"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from oracle import ppm_oracle as O                          # noqa: E402
from ppmstereo_amd import weights as Wm                     # noqa: E402
from ppmstereo_amd.corr import CorrBlock1D                  # noqa: E402
from ppmstereo_amd.ppmstereo import PPMStereoHotPath        # noqa: E402
from ppmstereo_amd.synth import synth_scale_inputs          # noqa: E402

DEV = torch.device("cuda:0")


def rel(name, got, ref):
    got, ref = got.detach().float().cpu(), ref.detach().float()
    e = got - ref
    rms = ref.pow(2).mean().sqrt().item()
    print(f"    {name:12s} rms err / rms = {e.pow(2).mean().sqrt().item() / rms:.3e}   max err / rms = {e.abs().max().item() / rms:.3e}   (rms {rms:.3f})")


def main():
    torch.set_num_threads(16)
    W = Wm.hot_path_weights()
    model = PPMStereoHotPath().load_hot_path_weights(W).to(DEV).eval()
    for tag, ai, T, h, w, isc, seed in (("update_block16", 0, 5, 4, 16, 4, 7), ("update_block16", 0, 5, 20, 32, 4, 8), ("update_block04", 2, 5, 16, 64, 1, 9)):
        print(f"{tag}: T={T} {h}x{w}")
        d = synth_scale_inputs(T, h, w, seed=seed, with_mhs=(tag != "update_block16"))
        Wb, Wa = W[tag], W[f"att.{ai}"]
        tr, rp, ru = [], [], []
        pyr = O.corr_pyramid(d["fmap1"], d["fmap2"])
        O.forward_update_block(Wb, Wa, pyr, d["flow"], d["net"], d["inp"], d["mhs"], 1, isc, T, tag == "update_block16", rp, ru, tr)
        t = tr[0]
        g = lambda x: None if x is None else x.to(DEV)
        blk = getattr(model, tag)
        with torch.cuda.device(DEV):
            eng = blk.engine(T, h, w, DEV)
            eng.set_inp(g(d["inp"])), eng.set_net(g(d["net"])), eng.set_flow(g(d["flow"])), eng.set_mhs(g(d["mhs"]))
            eng.begin(CorrBlock1D(g(d["fmap1"]), g(d["fmap2"])).levels, model.att[ai].packed(DEV))
            eng.lookup()
            rel("corr lookup", eng.store_nchw(eng.CORR.view(0, 36), 36), t["corr"])
            eng.motion_and_value()
            rel("mf", eng.get_mf(), t["mf"]), rel("value", eng.get_value(), t["value"])
            eng.uncertainty()
            rel("unc", eng.get_unc(), t["unc"])
            eng.pick()
            eng.attend()
            rel("mfg", eng.get_mfg(), t["mfg"])
            rel("mfg - mf", eng.get_mfg() - eng.get_mf(), t["mfg"] - t["mf"])
            # the same stage from the ORACLE's inputs (isolates the update block from what came before)
            eng.set_mf(g(t["mf"])), eng.set_mfg(g(t["mfg"]))
            eng.update()
            if tag == "update_block16":
                x = torch.cat([d["inp"], t["mf"], t["mfg"]], 1)
                xt = O.time_attn(Wb, x, T)
                rel("x time+space", eng.store_nchw(eng.XA.view(), 384), O.space_attn(Wb, xt))
            rel("net", eng.get_net(), t["net"]), rel("dflow", eng.get_dflow(), t["dflow"]), rel("mask", eng.get_mask(), t["mask"])


if __name__ == "__main__":
    main()
