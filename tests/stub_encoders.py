"""Stand-ins for the reference's encoders (fnet = BasicEncoder, cnet = Feature: outside the hot path) used by the
forward / forward_batch_test fixtures: frame f of the synthetic video is a constant image of value f, and the stubs return
hash features keyed on that frame id, so every sliding window sees its own frames.  tools/gen_golden.py drives the
reference with the same stubs (G8)."""
import torch

from ppmstereo_amd.weights import hash_normal


def frame_video(N: int, H0: int, W0: int) -> torch.Tensor:
    """(N,2,3,H0,W0) float32: both views of frame f filled with the value f."""
    return torch.arange(N, dtype=torch.float32)[:, None, None, None, None].expand(N, 2, 3, H0, W0).contiguous()


def _ids(img: torch.Tensor):
    H, W = img.shape[-2:]
    return [int(round(float((v + 1.0) * 255.0 / 2.0))) for v in img[:, 0, H // 2, W // 2].cpu()]


class StubFNet:
    def __call__(self, x):
        img = x[0]
        H, W = img.shape[-2:]
        ids = _ids(img)
        f1 = torch.stack([hash_normal((256, H // 4, W // 4), 2000 + f) for f in ids])
        f2 = torch.stack([hash_normal((256, H // 4, W // 4), 3000 + f) for f in ids])
        return f1.to(img.device), f2.to(img.device)


class StubCNet:
    def __call__(self, img):
        H, W = img.shape[-2:]
        ids = _ids(img)
        return tuple(torch.stack([hash_normal((256, H // s, W // s), 4000 + 100 * i + f) for f in ids]).to(img.device) for i, s in enumerate((4, 8, 16)))
