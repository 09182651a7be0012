"""Static guard for the packed-fp32 hardware condition (docs/LOG_r01_r05.md section 5): disassembles every gfx950 code object inside
ppmstereo_amd/libppms.so and fails if a v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 instruction is present.
usage: tools/check_no_packed_fp32.py [library.so]   (also imported by tests/test_host_logic.py)"""
import os
import re
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"
BANNED = re.compile(r"\bv_pk_(add|mul|fma)_f32\b")


def device_disassembly(lib_path: str):
    """yields (bundle index, disassembly text) for every gfx950 code object in the library's .hip_fatbin section"""
    with tempfile.TemporaryDirectory() as tmp:
        fat = os.path.join(tmp, "fatbin")
        subprocess.run([f"{LLVM}/llvm-objcopy", f"--dump-section=.hip_fatbin={fat}", lib_path, os.path.join(tmp, "copy.so")], check=True)
        blob = open(fat, "rb").read()
        starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
        assert starts, "no offload bundle in .hip_fatbin"
        for i, (a, b) in enumerate(zip(starts, starts[1:] + [len(blob)])):
            part, co = os.path.join(tmp, f"bundle{i}"), os.path.join(tmp, f"code{i}.co")
            open(part, "wb").write(blob[a:b])
            subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={part}", f"--output={co}"],
                           check=True)
            yield i, subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], check=True, capture_output=True, text=True).stdout


def scan(lib_path: str):
    """-> (code objects, instructions, MFMA instructions, list of banned instruction lines)"""
    n_obj = n_ins = n_mfma = 0
    hits = []
    for i, text in device_disassembly(lib_path):
        n_obj += 1
        for line in text.splitlines():
            if "\t" not in line:
                continue
            n_ins += 1
            n_mfma += "v_mfma_" in line
            if BANNED.search(line):
                hits.append(f"code object {i}: {line.strip()}")
    return n_obj, n_ins, n_mfma, hits


if __name__ == "__main__":
    path = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ppmstereo_amd", "libppms.so")
    n_obj, n_ins, n_mfma, hits = scan(path)
    print(f"{path}: {n_obj} code objects, {n_ins} instruction lines, {n_mfma} MFMA instructions, {len(hits)} packed fp32 arithmetic instructions")
    for h in hits[:20]:
        print("  ", h)
    sys.exit(1 if hits else 0)
