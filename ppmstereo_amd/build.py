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
# v_pk_mul_f32 / v_pk_add_f32 (packed fp32) gave wrong results in lanes 48-63 of a wave of the bilinear resize kernel
# whenever an MFMA-heavy kernel of another stream shared the SIMD (tools/race_probe.py, tests/test_gpu_concurrency.py: one of
# the four taps was lost in 16-element runs; waits and nops around the loads did not help, disabling packed fp32 formation
# did).  Every source except the memory attention is therefore compiled with -packed-fp32-ops; mem_attn.hip keeps the packed
# forms (its exp / sum stream is 0.19 ms per 1/4-scale call faster with them; it is checked against the CPU restatement, by the
# convex-combination property and under the same concurrency stress).
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
NO_PK = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
PACKED_FP32_SOURCES = ("mem_attn.hip",)
FLAGS = COMMON + NO_PK                      # (kept for callers that print the flags)


def _digest() -> str:
    h = hashlib.sha256(" ".join(COMMON + NO_PK + list(PACKED_FP32_SOURCES)).encode())
    for f in sorted(os.listdir(CSRC)) + ["../../include/ppms.h"]:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(f.encode() + fh.read())
    return h.hexdigest()


def build(force: bool = False, verbose: bool = True) -> str:
    dig = _digest()
    if not force and os.path.exists(LIB) and os.path.exists(STAMP) and open(STAMP).read() == dig:
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    # the target feature is meant for the device pass only; the host pass of the same clang invocation reports it as unknown
    noise = "'-packed-fp32-ops' is not a recognized feature for this target (ignoring feature)"

    def run(cmd):
        if verbose:
            print("[ppmstereo_amd.build]", " ".join(cmd), file=sys.stderr)
        res = subprocess.run(cmd, stderr=subprocess.PIPE, text=True)
        err = "\n".join(ln for ln in res.stderr.splitlines() if noise not in ln)
        if err.strip():
            print(err, file=sys.stderr)
        if res.returncode != 0:
            raise subprocess.CalledProcessError(res.returncode, cmd)

    objdir = os.path.join(HERE, "_obj")
    os.makedirs(objdir, exist_ok=True)
    objs = []
    for src in SOURCES:
        obj = os.path.join(objdir, src.replace(".hip", ".o"))
        run([hipcc] + COMMON + ([] if src in PACKED_FP32_SOURCES else NO_PK) + ["-c", os.path.join(CSRC, src), "-o", obj])
        objs.append(obj)
    run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", LIB])
    with open(STAMP, "w") as fh:
        fh.write(dig)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
