"""the HIP SST block (ppmstereo_amd/sst.py) against the reference's own forward_sst_block outputs
(tests/golden/sst_*.npz, tools/gen_golden.py from /root/reference/models/core/ppmstereo.py:322-395) and against the CPU oracle at BASELINE
config 2's 1/16 map."""
import time

import pytest
import torch

from golden_util import Golden
from ppmstereo_amd import weights as Wm
from ppmstereo_amd.weights import hash_normal

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def sst():
    assert torch.cuda.is_available()
    from ppmstereo_amd.sst import SSTBlock
    m = SSTBlock()
    assert list(m.state_dict().keys()) == list(Wm.sst_param_shapes().keys())
    m.load_state_dict(Wm.sst_weights(), strict=True)
    return m.to(DEV).eval()


@pytest.mark.parametrize("name,T,h,w", [("sst_T5", 5, 8, 12), ("sst_T3", 3, 6, 10)])
def test_sst_vs_reference_golden(sst, name, T, h, w):
    g = Golden(name)
    a, b = hash_normal((T, 256, h, w), 810 + T).to(DEV), hash_normal((T, 256, h, w), 820 + T).to(DEV)
    o1, o2 = sst(a, b, T)
    # 4 x (3 LoFTR layers + a temporal block) of fp32-accurate GEMMs, LayerNorms in fp32; features are O(1..10)
    g.check("f1", o1, 5e-4, 2e-4)
    g.check("f2", o2, 5e-4, 2e-4)
    p1, p2 = sst(a, b, T)
    assert torch.equal(o1, p1) and torch.equal(o2, p2)


def test_sst_config2_size_vs_oracle(sst):
    from oracle import ppm_oracle as O
    T, h, w = 5, 20, 32                                       # 320 x 512 at 1/16
    a, b = hash_normal((T, 256, h, w), 831), hash_normal((T, 256, h, w), 832)
    r1, r2 = O.sst_block(Wm.sst_weights(), a, b, T)
    da, db = a.to(DEV), b.to(DEV)
    o1, o2 = sst(da, db, T)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        o1, o2 = sst(da, db, T)
    torch.cuda.synchronize()
    print(f"SST block, T=5, 20x32 tokens per frame (76 GFLOP): {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms per call")
    for o, r in ((o1, r1), (o2, r2)):
        err = (o.cpu() - r).abs()
        assert torch.isfinite(o).all()
        assert err.max() < 3e-4 * max(1.0, r.abs().max().item()), (err.max().item(), r.abs().max().item())
