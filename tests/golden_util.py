"""Reader for tests/golden/*.npz (written by tools/gen_golden.py from the reference itself)."""
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Golden:
    def __init__(self, name):
        self.z = np.load(os.path.join(GOLDEN, name + ".npz"))
        self.keys = {}
        for k in self.z.files:
            if k.endswith("__shape"):
                continue
            if "__s" in k:
                base, step = k.rsplit("__s", 1)
                self.keys[base] = (k, int(step))
            else:
                self.keys[k] = (k, 1)

    def raw(self, key):
        return self.z[self.keys[key][0]]

    def shape(self, key):
        k, step = self.keys[key]
        return tuple(self.z[key + "__shape"]) if step > 1 else self.z[k].shape

    def diff(self, key, value):
        """max |value - golden| over the stored (possibly strided) sample, after a shape check."""
        k, step = self.keys[key]
        v = value.detach().float().cpu().numpy() if torch.is_tensor(value) else np.asarray(value, dtype=np.float32)
        assert tuple(v.shape) == tuple(self.shape(key)), (key, v.shape, self.shape(key))
        g = self.z[k]
        s = v.reshape(-1)[::step] if step > 1 else v
        assert np.isfinite(s).all(), key + ": non-finite values"
        return float(np.abs(s - g).max()), float(np.abs(g).max())

    def check(self, key, value, atol, rtol=0.0):
        d, scale = self.diff(key, value)
        assert d <= atol + rtol * scale, f"{key}: max|diff|={d:.3e} (tol {atol + rtol * scale:.3e}, |golden|max={scale:.3e})"
        return d
