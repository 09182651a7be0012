"""Builds libppms.so (the HIP kernels + C ABI) in-tree with hipcc for gfx950.

    python -m ppmstereo_amd.build          # or __graft_entry__.build()

hipcc cross-compiles without a GPU; the .so is git-ignored but travels with the repo snapshot to the GPU box.

Concurrency: several ranks of one torchrun job may import the package at once.  The whole build runs under an exclusive
``flock`` on ``libppms.so.lock``; objects are compiled into a private temporary directory and the finished library and
its stamp are moved into place with ``os.replace`` (atomic), so a process either maps the old complete library or the
new complete one.  Ranks that lose the race wait on the lock, re-check the stamp and return without compiling.
"""
from __future__ import annotations

import fcntl
import hashlib
import os
import shutil
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libppms.so")
STAMP = LIB + ".stamp"
LOCK = LIB + ".lock"
SOURCES = ["corr.hip", "conv_gemm2.hip", "conv_gemm5.hip", "conv_gemm6.hip", "gemm1.hip", "conv_stream.hip", "small_ops.hip", "encoder_ops.hip", "mem_attn.hip", "attn16.hip", "pwchain.hip"]
HEADERS = ["common.h", "corr_lookup.h", "conv_epilogue.h", "conv5_asm.h", "conv6_asm.h", "attn64_asm.h", os.path.join("..", "..", "include", "ppms.h")]
# Packed fp32 VALU forms (v_pk_mul_f32 / v_pk_add_f32 / v_pk_fma_f32) are disabled (NO_PK).  Measured on MI355X: such an instruction
# with op_sel:[0,1] (low result = src0.lo op src1.hi -- the compiler picks that form freely, e.g. in the bilinear resize kernel) reads
# src1.hi as 0 in lanes 48-63 whenever ANOTHER wave on the same SIMD is issuing MFMAs, and the engine runs MFMA kernels beside small
# kernels on two streams.  Register-only reproducer: tools/pk_opsel_probe.py; evidence: profiles/r02_pk_opsel_probe.txt; static guard:
# tools/check_no_packed_fp32.py (tests/test_host_logic.py); docs/LOG_r01_r05.md section 5.  Sources listed in PACKED_FP32_SOURCES keep the packed forms.
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-I" + os.path.join(ROOT, "include"), "-I" + CSRC]
NO_PK = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
EXTRA = [x for x in os.environ.get("PPMS_BUILD_DEFINES", "").split() if x]        # build-time A/B only, e.g. "-DPPMS_CONV5_TIMING"
PACKED_FP32_SOURCES: tuple = tuple(x for x in os.environ.get("PPMS_BUILD_PACKED_FP32", "").split(",") if x)   # build-time A/B only
FLAGS = COMMON + NO_PK + EXTRA


def _digest() -> str:
    """sha256 over the flags and exactly the files that go into the library (editor temp files do not count).  The checkout's own location is
    taken out of the -I paths first: the library built here travels with the tree to the GPU box, where the tree sits under another path -- with
    absolute paths in the digest every fresh box found the stamp "stale" and spent its first minute recompiling identical sources (rounds 1-5)."""
    h = hashlib.sha256(" ".join(x.replace(ROOT, "$ROOT") for x in FLAGS + list(PACKED_FP32_SOURCES)).encode())
    for f in SOURCES + HEADERS:
        with open(os.path.join(CSRC, f), "rb") as fh:
            h.update(os.path.basename(f).encode() + fh.read())
    return h.hexdigest()


def _fresh(dig: str) -> bool:
    try:
        return os.path.exists(LIB) and open(STAMP).read() == dig
    except OSError:
        return False


def build(force: bool = False, verbose: bool = True) -> str:
    dig = _digest()
    if not force and _fresh(dig):
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

    with open(LOCK, "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        try:
            if not force and _fresh(dig):          # another process built it while this one waited
                return LIB
            tmp = tempfile.mkdtemp(prefix="_build_", dir=HERE)
            try:
                objs = []
                procs = []
                for src in SOURCES:                # compile the translation units in parallel (independent hipcc processes)
                    obj = os.path.join(tmp, src.replace(".hip", ".o"))
                    cmd = [hipcc] + (COMMON + EXTRA if src in PACKED_FP32_SOURCES else FLAGS) + ["-c", os.path.join(CSRC, src), "-o", obj]
                    if verbose:
                        print("[ppmstereo_amd.build]", " ".join(cmd), file=sys.stderr)
                    procs.append((cmd, subprocess.Popen(cmd, stderr=subprocess.PIPE, text=True)))
                    objs.append(obj)
                for cmd, pr in procs:
                    _, err = pr.communicate()
                    err = "\n".join(ln for ln in err.splitlines() if noise not in ln)
                    if err.strip():
                        print(err, file=sys.stderr)
                    if pr.returncode != 0:
                        raise subprocess.CalledProcessError(pr.returncode, cmd)
                out = os.path.join(tmp, "libppms.so")
                run([hipcc, "--offload-arch=gfx950", "-shared", "-fPIC"] + objs + ["-o", out])
                stamp = os.path.join(tmp, "stamp")
                with open(stamp, "w") as fh:
                    fh.write(dig)
                if os.path.exists(STAMP):
                    os.remove(STAMP)               # no window in which a new library sits beside an old stamp that matches
                os.replace(out, LIB)
                os.replace(stamp, STAMP)
            finally:
                shutil.rmtree(tmp, ignore_errors=True)
        finally:
            fcntl.flock(lock, fcntl.LOCK_UN)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv))
