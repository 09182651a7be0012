"""Do the latency-bound small scales hide behind the 1/4 scale of ANOTHER clip?  (GPU box)

Runs the 1/16 + 1/8 scales of config 2 (5 + 5 iterations) and its 1/4 scale (10 iterations) N times each: alone, back to back on one
stream (today's cascade), and concurrently on two streams (what a software pipeline over consecutive clips / windows would do).

    python tools/overlap_probe.py
"""
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from ppmstereo_amd import weights as Wm                                   # noqa: E402
from ppmstereo_amd.corr import CorrBlock1D                                # noqa: E402
from ppmstereo_amd.ppmstereo import PPMStereoHotPath, _run_iterations    # noqa: E402
from ppmstereo_amd.synth import synth_cascade_feats                       # noqa: E402

DEV = torch.device("cuda:0")


def main():
    T, H, W, N = 5, 320, 512, 20
    model = PPMStereoHotPath().load_hot_path_weights(Wm.hot_path_weights()).to(DEV).eval()
    feats = {k: v.to(DEV) for k, v in synth_cascade_feats(T, H, W).items()}

    def scale(s_, blk, ai, n_it, isc):
        f1, f2 = feats[f"f1_{s_}"], feats[f"f2_{s_}"]
        h, w = f1.shape[2:]
        eng = blk.engine(T, h, w, DEV)
        eng.set_inp(feats[f"inp_{s_}"]), eng.set_net(feats[f"net_{s_}"]), eng.set_flow(model.zero_init(f1)), eng.set_mhs(None)
        eng.begin(CorrBlock1D(f1, f2).levels, model.att[ai].packed(DEV))
        _run_iterations(eng, n_it, isc, T, h, w, [], [], "last" if s_ == 4 else "none")

    small = lambda: (scale(16, model.update_block16, 0, 5, 4), scale(8, model.update_block08, 1, 5, 2))
    large = lambda: scale(4, model.update_block04, 2, 10, 1)
    with torch.cuda.device(DEV):
        for _ in range(2):
            small(), large()
        torch.cuda.synchronize()

        def timed(fn):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / N * 1e3

        t_small = timed(lambda: [small() for _ in range(N)])
        t_large = timed(lambda: [large() for _ in range(N)])
        t_seq = timed(lambda: [(small(), large()) for _ in range(N)])
        sA, sB = torch.cuda.Stream(device=DEV), torch.cuda.Stream(device=DEV)

        def both():
            for _ in range(N):                      # host order small, large, small, ...: both queues stay fed
                with torch.cuda.stream(sA):
                    small()
                with torch.cuda.stream(sB):
                    large()

        t_both = timed(both)
    print(f"per clip: small scales alone {t_small:.2f} ms, 1/4 scale alone {t_large:.2f} ms, back to back on one stream {t_seq:.2f} ms, "
          f"concurrently on two streams {t_both:.2f} ms ({100 * (1 - t_both / t_seq):.1f} % less)")


if __name__ == "__main__":
    main()
