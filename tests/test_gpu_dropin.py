"""The drop-in import path: ``models.core.*`` of this repository resolves the imports of the reference's wrapper
(/root/reference/models/ppm_stereo_model.py:12, models/core/ppmstereo.py:17-33), and a model built, loaded and called exactly the way
that wrapper does it (ppm_stereo_model.py:27-50; pytorch3d's Configurable base aside) gives the oracle's disparities."""
import pytest
import torch

from oracle import ppm_oracle as O
from ppmstereo_amd import weights as Wm
from ppmstereo_amd.weights import hash_normal

pytestmark = pytest.mark.gpu


def _procedural_checkpoint():
    """state_dict of the whole model (the reference's keys, in its order) filled with the procedural weights."""
    from ppmstereo_amd.ppmstereo import PPMStereo as Donor
    m = Donor.shipped()
    m.load_hot_path_weights(Wm.hot_path_weights())
    m.fnet.load_state_dict(Wm.fnet_weights(), strict=True)
    m.cnet.load_state_dict(Wm.cnet_weights(), strict=True)
    sd = m.state_dict()
    sd.update(Wm.sst_weights())
    return {k: v.clone() for k, v in sd.items()}


def test_model_built_loaded_and_called_like_the_reference_wrapper(tmp_path):
    from models.core.ppmstereo import PPMStereo                               # models/ppm_stereo_model.py:12
    ckpt = str(tmp_path / "ppmstereo_final.pth")
    torch.save({"model": _procedural_checkpoint(), "total_steps": 200000}, ckpt)  # as train.py:283-299 writes it
    # ---- models/ppm_stereo_model.py:26-44, line for line in effect
    model = PPMStereo(mixed_precision=True, num_frames=5, attention_type="self_stereo_temporal_update_time_update_space",
                      use_3d_update_block=True, different_update_blocks=True)
    state_dict = torch.load(ckpt, map_location="cpu")
    if "model" in state_dict:
        state_dict = state_dict["model"]
    if "state_dict" in state_dict:
        state_dict = state_dict["state_dict"]
        state_dict = {"module." + k: v for k, v in state_dict.items()}
    res = model.load_state_dict(state_dict, strict=False)
    assert not res.missing_keys and not res.unexpected_keys, (res.missing_keys[:3], res.unexpected_keys[:3])
    model.to("cuda")
    model.eval()
    # ---- ppm_stereo_model.py:47-50: forward(batch_dict, iters) -> forward_batch_test(batch_dict, kernel_size=20, iters=iters)
    N, H0, W0 = 6, 60, 250
    video = (torch.sigmoid(hash_normal((N, 2, 3, H0, W0), 961)) * 255.0).contiguous()
    out = model.forward_batch_test({"stereo_video": video}, kernel_size=20, iters=4)
    assert set(out) == {"disparity", "uncertainties"} and tuple(out["disparity"].shape) == (N, 1, H0, W0) and not out["disparity"].is_cuda
    W = Wm.hot_path_weights()
    Wf, Ws, Wc = Wm.fnet_weights(), Wm.sst_weights(), Wm.cnet_weights()
    ref = O.forward_batch_test(W, lambda x: O.basic_encoder(Wf, x), lambda im: O.feature_cnet(Wc, im), video, 20, 4,
                               sst_fn=lambda a, b: O.sst_block(Ws, a, b, N))
    err = (out["disparity"] - ref["disparity"]).abs()
    print(f"wrapper-style model: EPE vs oracle {err.mean().item():.3e} px, max {err.max().item():.3e} px")
    assert err.mean().item() < 1e-3 and (out["uncertainties"] - ref["uncertainties"]).abs().max().item() < 2e-2      # measured 6.2e-4 px
    # a checkpoint in the other format the wrapper accepts ({"state_dict": ...} gets a "module." prefix there and, with strict=False,
    # then loads NOTHING: reference behaviour, ppm_stereo_model.py:37-41): every key is reported as unexpected, no exception
    other = PPMStereo(mixed_precision=True, num_frames=5, attention_type="self_stereo_temporal_update_time_update_space",
                      use_3d_update_block=True, different_update_blocks=True)
    res = other.load_state_dict({"module." + k: v for k, v in state_dict.items()}, strict=False)
    assert len(res.unexpected_keys) == len(state_dict) and len(res.missing_keys) == len(state_dict)


def test_hot_path_imports_of_the_reference_resolve():
    """models/core/ppmstereo.py:17-33 of the reference, the names on the hot path."""
    from models.core.corr import CorrBlock1D
    from models.core.ppmtereo_update import Attention_qk, SequenceUpdateBlock3D, get_temporal_positional_encoding
    from models.core.utils.utils import InputPadder, interp
    import ppmstereo_amd.corr
    import ppmstereo_amd.update
    assert CorrBlock1D is ppmstereo_amd.corr.CorrBlock1D and SequenceUpdateBlock3D is ppmstereo_amd.update.SequenceUpdateBlock3D
    pe = get_temporal_positional_encoding(5, 128, "cuda:0", is_normalize=True, scale=1.0)
    assert tuple(pe.shape) == (5, 1, 1, 128)
    x = torch.randn(1, 2, 10, 12, device="cuda:0")
    assert tuple(interp(x, (20, 24)).shape) == (1, 2, 20, 24)
    p = InputPadder((1, 3, 60, 250), divis_by=32)
    assert tuple(p.pad(torch.zeros(1, 3, 60, 250))[0].shape) == (1, 3, 64, 256)
    assert Attention_qk(num_heads=1, dim_head=128).to_qk.weight.shape == (256, 128, 1, 1)
