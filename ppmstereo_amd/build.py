"""Builds libppms.so (the HIP kernels + C ABI) in-tree with hipcc for gfx950.

    python -m ppmstereo_amd.build          # or __graft_entry__.build()

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the repo snapshot to the GPU box.
"""
from __future__ import annotations

import hashlib
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libppms.so")
STAMP = LIB + ".stamp"
SOURCES = ["corr.hip", "conv_gemm.hip", "conv_gemm2.hip", "conv_gemm3.hip", "small_ops.hip", "mem_attn.hip", "attn16.hip", "pwchain.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off",
         "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]


def _digest() -> str:
    h = hashlib.sha256(" ".join(FLAGS).encode())
    for f in sorted(os.listdir(CSRC)) + ["../../include/ppms.h"]:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + fh.read())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP) and open(STAMP).read() == dig:
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    cmd = [hipcc] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print("[ppmstereo_amd.build]", " ".join(cmd), file=sys.stderr)
    subprocess.run(cmd, check=True)
    with open(STAMP, "w") as fh:
        fh.write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
