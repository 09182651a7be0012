"""the HIP fnet (ppmstereo_amd/encoder.py) against the reference's own BasicEncoder outputs
(tests/golden/fnet_*.npz, written by tools/gen_golden.py from /root/reference/models/core/extractor.py:348-423) and against the CPU oracle at
the benchmark's image size."""
import time

import pytest
import torch

from golden_util import Golden
from ppmstereo_amd import weights as Wm

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def fnet():
    assert torch.cuda.is_available()
    from ppmstereo_amd.encoder import BasicEncoder
    m = BasicEncoder(output_dim=256, norm_fn="instance")
    m.load_state_dict(Wm.fnet_weights(), strict=True)
    return m.to(DEV).eval()


@pytest.mark.parametrize("name,n,hh,ww", [("fnet_small", 2, 64, 96), ("fnet_odd", 1, 40, 72)])
def test_fnet_vs_reference_golden(fnet, name, n, hh, ww):
    g = Golden(name)
    i1, i2 = Wm.hash_uniform((n, 3, hh, ww), 600 + hh).to(DEV), Wm.hash_uniform((n, 3, hh, ww), 700 + hh).to(DEV)
    f1, f2 = fnet([i1, i2])
    assert f1.shape == (n, 256, hh // 4, ww // 4) and f2.shape == f1.shape
    # fp32-accurate convolutions (bf16x3 split) through 14 conv + InstanceNorm layers: a few 1e-5 of the feature range
    g.check("fmap1", f1, 2e-4, 2e-4)
    g.check("fmap2", f2, 2e-4, 2e-4)
    # single-tensor call (extractor.py:396-397: no list) == first half of the pair call; and bit-reproducible
    assert torch.equal(fnet(i1), fnet(i1))
    assert (fnet(i1) - f1).abs().max() < 1e-6


def test_fnet_full_size_vs_oracle(fnet):
    """BASELINE config 2 images: T = 5 frames of 320 x 512, left + right = 10 images in one call (ppmstereo.py:618)."""
    from oracle import ppm_oracle as O
    T, H, W = 5, 320, 512
    i1, i2 = Wm.hash_uniform((T, 3, H, W), 611), Wm.hash_uniform((T, 3, H, W), 612)
    torch.set_num_threads(16)
    r1, r2 = O.basic_encoder(Wm.fnet_weights(), [i1, i2])
    d1, d2 = i1.to(DEV), i2.to(DEV)
    f1, f2 = fnet([d1, d2])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        f1, f2 = fnet([d1, d2])
    torch.cuda.synchronize()
    ms = (time.perf_counter() - t0) / 5 * 1e3
    print(f"fnet, 10 images of 320x512 (326 GFLOP): {ms:.2f} ms per call")
    for f, r in ((f1, r1), (f2, r2)):
        err = (f.cpu() - r).abs()
        assert torch.isfinite(f).all()
        assert err.max() < 3e-4 * max(1.0, r.abs().max().item()), (err.max().item(), r.abs().max().item())
        assert err.mean() < 2e-5 * max(1.0, r.abs().mean().item() * 10), err.mean().item()


def test_fnet_rejects_what_it_does_not_support(fnet):
    from ppmstereo_amd.encoder import BasicEncoder
    with pytest.raises(NotImplementedError):
        BasicEncoder(norm_fn="batch")
    with pytest.raises(RuntimeError):
        fnet(torch.zeros(1, 3, 64, 64))                       # CPU tensor: no fallback
    with pytest.raises(ValueError):
        fnet(torch.zeros(1, 3, 66, 64, device=DEV))


def test_encoders_config3_geometry_properties(fnet):
    """BASELINE config 3 images (T = 5, 736 x 1280 after InputPadder): too large for the CPU oracle in test time, so fnet and cnet are checked
    through properties there -- shapes, finite, bit-reproducible, per-(sample, channel) statistics of an InstanceNorm'ed layer --
    and the planners / launch lists are exercised at 184 x 320 .. 23 x 40 maps."""
    from ppmstereo_amd.cnet import Feature
    T, H, W = 5, 736, 1280
    i1, i2 = Wm.hash_uniform((T, 3, H, W), 711).to(DEV), Wm.hash_uniform((T, 3, H, W), 712).to(DEV)
    f1, f2 = fnet([i1, i2])
    g1, g2 = fnet([i1, i2])
    assert f1.shape == (T, 256, H // 4, W // 4) and torch.isfinite(f1).all() and torch.isfinite(f2).all()
    assert torch.equal(f1, g1) and torch.equal(f2, g2)
    assert 0.05 < f1.std().item() < 50.0                     # a live signal, not zeros / blow-up
    cnet = Feature("tiny", 256)
    cnet.load_state_dict(Wm.cnet_weights(), strict=True)
    cnet = cnet.to(DEV).eval()
    c4, c8, c16 = cnet(i1)
    d4, d8, d16 = cnet(i1)
    assert c4.shape == (T, 256, H // 4, W // 4) and c8.shape == (T, 256, H // 8, W // 8) and c16.shape == (T, 256, H // 16, W // 16)
    assert all(torch.isfinite(t).all() for t in (c4, c8, c16))
    assert torch.equal(c4, d4) and torch.equal(c8, d8) and torch.equal(c16, d16)


def test_large_batches_run_and_match_small_ones(fnet):
    """The reference's evaluation default forward_batch_test(kernel_size=20) hands the encoders 2 x 20 = 40 images at once (ppmstereo.py:
    277-294, 614-618): at 544 x 960 the half-resolution stem is 40 x 272 x 480 = 5.2 M pixels, beyond the 2^22-pixel limit the implicit-GEMM
    kernel used to have.  fnet's InstanceNorm and every other layer are per image, so the features of an image must not depend on the
    batch it travels in: the last frames of the 40-image call against a 4-image call (size-independent property; the launch plans --
    K slices, tile counts -- differ between the two, hence a tolerance).  Then forward_batch_test itself on the 20-frame video."""
    m = fnet
    N, H, W = 20, 544, 960
    i1, i2 = Wm.hash_uniform((N, 3, H, W), 650).to(DEV), Wm.hash_uniform((N, 3, H, W), 651).to(DEV)
    f1, f2 = m([i1, i2])
    assert f1.shape == (N, 256, H // 4, W // 4) and torch.isfinite(f1).all() and torch.isfinite(f2).all()
    g1, g2 = m([i1[-2:], i2[-2:]])
    scale = f1[-2:].abs().max().item()
    assert (f1[-2:] - g1).abs().max().item() < 2e-4 * scale and (f2[-2:] - g2).abs().max().item() < 2e-4 * scale
    del f1, f2, g1, g2
    from ppmstereo_amd.cnet import Feature
    c = Feature("tiny", 256)
    c.load_state_dict(Wm.cnet_weights(), strict=True)
    c = c.to(DEV).eval()
    a4, a8, a16 = c(i1)
    b4, b8, b16 = c(i1[-2:])
    for a, b in ((a4, b4), (a8, b8), (a16, b16)):
        assert torch.isfinite(a).all() and (a[-2:] - b).abs().max().item() < 5e-4 * a[-2:].abs().max().item()
    del a4, a8, a16, b4, b8, b16, c
    torch.cuda.empty_cache()
    from ppmstereo_amd.ppmstereo import PPMStereo
    whole = PPMStereo.shipped()
    whole.load_hot_path_weights(Wm.hot_path_weights())
    whole.fnet.load_state_dict(Wm.fnet_weights(), strict=True), whole.cnet.load_state_dict(Wm.cnet_weights(), strict=True)
    sd = whole.state_dict()
    sd.update(Wm.sst_weights())
    whole.load_state_dict(sd, strict=True)
    whole = whole.to(DEV).eval()
    video = torch.stack([(i1.cpu() + 1) * 127.5, (i2.cpu() + 1) * 127.5], 1)               # (20, 2, 3, H, W) in [0, 255]
    out = whole.forward_batch_test({"stereo_video": video}, kernel_size=20, iters=2)      # windows [0, 20) and [10, 20): 40 and 20 images
    assert tuple(out["disparity"].shape) == (N, 1, H, W) and torch.isfinite(out["disparity"]).all()
    assert (out["uncertainties"] > 0).all() and (out["uncertainties"] < 1).all()
