"""the HIP cnet (ppmstereo_amd/cnet.py) against the reference's own Feature("tiny", 256) outputs
(tests/golden/cnet_*.npz, tools/gen_golden.py from /root/reference/models/core/convnext.py:202-264) and against the CPU oracle at the
benchmark's image size."""
import time

import pytest
import torch

from golden_util import Golden
from ppmstereo_amd import weights as Wm

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def cnet():
    assert torch.cuda.is_available()
    from ppmstereo_amd.cnet import Feature
    m = Feature("tiny", 256)
    assert list(m.state_dict().keys()) == list(Wm.cnet_param_shapes().keys())
    m.load_state_dict(Wm.cnet_weights(), strict=True)
    return m.to(DEV).eval()


@pytest.mark.parametrize("name,n,hh,ww", [("cnet_small", 2, 64, 96), ("cnet_32", 1, 32, 64)])
def test_cnet_vs_reference_golden(cnet, name, n, hh, ww):
    g = Golden(name)
    img = Wm.hash_uniform((n, 3, hh, ww), 900 + hh).to(DEV)
    c4, c8, c16 = cnet(img)
    assert c4.shape == (n, 256, hh // 4, ww // 4) and c8.shape == (n, 256, hh // 8, ww // 8) and c16.shape == (n, 256, hh // 16, ww // 16)
    # 18 ConvNeXt blocks + 3 decoder levels of fp32-accurate GEMMs; LayerNorm / GRN / InstanceNorm in fp32
    g.check("c4", c4, 5e-4, 3e-4), g.check("c8", c8, 5e-4, 3e-4), g.check("c16", c16, 5e-4, 3e-4)
    d4, d8, d16 = cnet(img)
    assert torch.equal(c4, d4) and torch.equal(c8, d8) and torch.equal(c16, d16)


def test_cnet_full_size_vs_oracle(cnet):
    """BASELINE config 2: the T = 5 left images of 320 x 512 in one call (ppmstereo.py:624)."""
    from oracle import ppm_oracle as O
    T, H, W = 5, 320, 512
    img = Wm.hash_uniform((T, 3, H, W), 911)
    torch.set_num_threads(16)
    refs = O.feature_cnet(Wm.cnet_weights(), img)
    d = img.to(DEV)
    outs = cnet(d)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        outs = cnet(d)
    torch.cuda.synchronize()
    print(f"cnet, 5 images of 320x512 (325 GFLOP): {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per call")
    for o, r in zip(outs, refs):
        err = (o.cpu() - r).abs()
        assert torch.isfinite(o).all()
        assert err.max() < 5e-4 * max(1.0, r.abs().max().item()), (err.max().item(), r.abs().max().item())


def test_cnet_rejects_what_it_does_not_support(cnet):
    from ppmstereo_amd.cnet import Feature
    with pytest.raises(NotImplementedError):
        Feature("base")
    with pytest.raises(RuntimeError):
        cnet(torch.zeros(1, 3, 64, 64))
    with pytest.raises(ValueError):
        cnet(torch.zeros(1, 3, 48, 64, device=DEV))
