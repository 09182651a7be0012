"""Generates ppmstereo_amd/csrc/conv6_asm.h: the hand-ordered k32-step of conv_gemm6.hip (one wave per SIMD, v_mfma_f32_16x16x32_bf16).

A wave owns MB blocks of 16 couts x NBW blocks of 16 pixels.  One k32-step (32 input channels of one tap) is, per pixel block n,
    acc[m][n] += W_hi[m] x X_hi[n]     (MB MFMAs)
    acc[m][n] += W_lo[m] x X_hi[n]     (MB MFMAs)
    acc[m][n] += W_hi[m] x X_lo[n]     (MB MFMAs; left out when the window's lo plane is known to be zero)
with the 2 MB weight fragments of the step resident in registers (loaded from L2 one step ahead, two register stages) and the two activation
fragments of a block read from the LDS window two blocks ahead into a ring of slots.  Every instruction is its own `asm volatile` statement:
volatile statements keep their order, so the lists below ARE the issue order; the compiler only allocates registers and places the
scalar / address arithmetic of the C code around the calls.  Accumulators live in the AGPR half of the register file ("+a").

usage: python tools/gen_conv6_asm.py
"""
import os

MF = "v_mfma_f32_16x16x32_bf16"
ABL = int(os.environ.get("PPMS_CONV6_ABL", "0"))     # timing experiments only (wrong results): 1 drops the LDS fragment reads and their waits,
                                                     # 2 the weight-fragment loads, 4 the MFMAs, 8 the LDS-DMA pieces of the loop, 32 the waits for the LDS fragment reads (the reads stay)
VARIANTS = [(4, 13), (3, 13), (4, 7), (4, 6)]        # (MB, NBW): M = 256, M = 192, the two pixel halves of M = 128


def slot_of(n, nbw):
    """Ring slot of pixel block n: blocks cycle through slots 0..2; the NBW % 3 leftover blocks at the end of a step get slots of their own, so
    that the ring phase is the same in every step (the next step's blocks 0 and 1 are requested into slots 0 and 1 during this step's last two
    blocks)."""
    full = nbw - nbw % 3
    return n % 3 if n < full else 3 + (n - full)


def a_blocks(mb, nbw):
    """Which weight-fragment loads (k = 0 .. 2 MB - 1) of the NEXT step are issued in which block: two per block over the first blocks of the step (the
    step ends with a wait for them: the earlier they leave, the less of their L2 round trip is exposed)."""
    nab = min(mb, max(1, nbw - 3))
    per = {}
    for k in range(2 * mb):
        per.setdefault(k * nab // (2 * mb), []).append(k)
    return nab, per


class Emit:
    def __init__(self):
        self.lines = []

    def asm(self, text, outs=(), ins=(), clob='"memory"'):
        op = text.split()[0]
        if (ABL & 32 and "lgkmcnt" in text) or (ABL & 1 and (op == "ds_read_b128" or "lgkmcnt" in text)) or (ABL & 2 and op == "global_load_dwordx4") or (ABL & 4 and MF in text):
            return
        ops = list(outs) + list(ins)
        for i, (nm, _, _) in enumerate(ops):
            text = text.replace("{" + nm + "}", "%" + str(i))
        o = ", ".join(f'"{c}"({e})' for _, c, e in outs)
        i = ", ".join(f'"{c}"({e})' for _, c, e in ins)
        self.lines.append(f'        asm volatile("{text}" : {o} : {i} : {clob});' if (o or i) else f'        asm volatile("{text}" ::: {clob});')

    def c(self, text):
        self.lines.append("        " + text)


def step(mb, nbw, skip):
    """skip: the step body without the hi x lo products (windows whose lo plane is all zero).  Returns (text, number of DMA slots)."""
    assert nbw % 3 <= 1, "the ring has one spare slot"
    E = Emit()
    nab, aper = a_blocks(mb, nbw)
    slots = 0
    pending = 2 * mb                      # weight loads of the next step not yet issued

    def slot(group):
        # a DMA slot: hook(K) issues at most ONE LDS-DMA piece.  Slots come behind the last weight load (everything a hook issues must be YOUNGER
        # than the weight loads on the in-order vmcnt counter) and at least four MFMAs (64 cycles) apart: an LDS-DMA instruction reads M0 and its
        # address register some time AFTER it has issued -- the next s_mov m0 / a write of that register within a few cycles of it misdirects the
        # transfer (tools/conv6_stress.py: back-to-back pieces left window rows stale; 64 cycles of distance: 300 launches clean)
        nonlocal slots
        if pending == 0 and not (ABL & 8) and (mb >= 4 or group == 1):
            E.c(f"hook({slots});")
            slots += 1

    for n in range(nbw):
        s = slot_of(n, nbw)
        # the fragments of block n were requested two blocks ago; the requests of block n + 1 may stay in flight
        E.asm("s_waitcnt lgkmcnt(2)")
        # the requests issued in this block: block n + 2 of this step, or block n + 2 - NBW of the next one; the block's column offset inside
        # the window is an immediate (CR = rows per window column is a template parameter)
        if n + 2 < nbw:
            bh, bl, blk, ts = "bh", "bl", n + 2, slot_of(n + 2, nbw)
        else:
            bh, bl, blk, ts = "bhn", "bln", n + 2 - nbw, slot_of(n + 2 - nbw, nbw)
        loads = list(aper.get(n, []))
        for m in range(mb):                                   # hi x hi
            E.asm(f"{MF} {{c}}, {{a}}, {{b}}, {{c}}", [("c", "+a", f"acc[{m}][{n}]")], [("a", "v", f"a[{2 * m}]"), ("b", "v", f"ring[{s}][0]")])
            if m == 0:
                E.asm("ds_read_b128 {d}, {p} offset:{o}", [("d", "+v", f"ring[{ts}][0]")], [("p", "v", bh), ("o", "n", f"{blk} * CR * 128")])
            elif m == 1:
                E.asm("ds_read_b128 {d}, {p} offset:{o}", [("d", "+v", f"ring[{ts}][1]")], [("p", "v", bl), ("o", "n", f"{blk} * CR * 128")])
            elif loads:
                k = loads.pop(0)
                pending -= 1
                E.asm(f"global_load_dwordx4 {{d}}, {{o}}, {{sb}} offset:{(k & 3) * 1024}", [("d", "+v", f"an[{k}]")],
                      [("o", "v", "avoff0" if k < 4 else "avoff1"), ("sb", "s", "sb_next")])
        if not loads:
            slot(0)
        for m in range(mb):                                   # lo x hi
            E.asm(f"{MF} {{c}}, {{a}}, {{b}}, {{c}}", [("c", "+a", f"acc[{m}][{n}]")], [("a", "v", f"a[{2 * m + 1}]"), ("b", "v", f"ring[{s}][0]")])
            if loads:
                k = loads.pop(0)
                pending -= 1
                E.asm(f"global_load_dwordx4 {{d}}, {{o}}, {{sb}} offset:{(k & 3) * 1024}", [("d", "+v", f"an[{k}]")],
                      [("o", "v", "avoff0" if k < 4 else "avoff1"), ("sb", "s", "sb_next")])
        assert not loads
        slot(1)
        if not skip:
            for m in range(mb):                               # hi x lo
                E.asm(f"{MF} {{c}}, {{a}}, {{b}}, {{c}}", [("c", "+a", f"acc[{m}][{n}]")], [("a", "v", f"a[{2 * m}]"), ("b", "v", f"ring[{s}][1]")])
            slot(2)
    return "\n".join(E.lines), slots


def prime(nbw):
    E = Emit()
    for n in range(2):
        E.asm("ds_read_b128 {d}, {p} offset:{o}", [("d", "+v", f"ring[{slot_of(n, nbw)}][0]")], [("p", "v", "bh"), ("o", "n", f"{n} * CR * 128")])
        E.asm("ds_read_b128 {d}, {p} offset:{o}", [("d", "+v", f"ring[{slot_of(n, nbw)}][1]")], [("p", "v", "bl"), ("o", "n", f"{n} * CR * 128")])
    return "\n".join(E.lines)


def gen():
    out = ['''// GENERATED by tools/gen_conv6_asm.py -- do not edit.
// The k32-step of conv_gemm6.hip: per pixel block 3 x MB v_mfma_f32_16x16x32_bf16 (hi x hi, lo x hi, hi x lo) with this step's weight fragments
// in registers, the LDS requests for the activation fragments of the block two ahead and the global loads of the NEXT step's weight fragments
// placed between them by hand, one memory instruction per MFMA gap.  hook(K) (C code: at most ONE LDS-DMA piece of the next window) is called
// behind MFMA groups once the last weight load is out -- everything it issues is YOUNGER than the weight loads on the in-order vmcnt counter --
// and never less than four MFMAs after the previous call (see the generator: an LDS-DMA reads M0 and its address register after it has issued).
// Ring slots: see tools/gen_conv6_asm.py:slot_of().
#pragma once
''']
    out.append("template <int MB, int NBW, bool SKIP> struct conv6_shape;")
    for mb, nbw in VARIANTS:
        for skip in (False, True):
            _, nslots = step(mb, nbw, skip)
            out.append(f"template <> struct conv6_shape<{mb}, {nbw}, {'true' if skip else 'false'}> {{ static constexpr int SLOTS = {nslots}; }};   // hook calls of a step")
    out.append("")
    out.append("// CR: rows of a window column (16 + y halo): the column offsets of the fragment reads are immediates.  bh / bl: LDS address of this lane's hi / lo\n"
               "// fragment of the wave's first block at the current tap (bl = bh ^ 64); bhn / bln: the same for the next step.  SKIP: the body without the\n"
               "// hi x lo products (windows whose lo plane is all zero).\n"
               "template <int MB, int NBW, int CR, bool SKIP, class Hook>\n__device__ __forceinline__ void conv6_step(f32x4 (&acc)[4][13], const u32x4 (&a)[8], u32x4 (&an)[8], u32x4 (&ring)[4][2],\n"
               "        unsigned bh, unsigned bl, unsigned bhn, unsigned bln, unsigned avoff0, unsigned avoff1, const char* sb_next, Hook&& hook) {")
    first = True
    for mb, nbw in VARIANTS:
        for skip in (False, True):
            b0, _ = step(mb, nbw, skip)
            out.append(("    if constexpr (" if first else "    } else if constexpr (") + f"MB == {mb} && NBW == {nbw} && {'SKIP' if skip else '!SKIP'}) {{\n" + b0)
            first = False
        first = False
    out.append("    }\n}\n")
    out.append("// the requests for blocks 0 and 1 of a step (the loop's first step, and the first step behind every window switch: the tail requests of the\n"
               "// step in front of the barrier read a window that may not have landed yet and are simply issued again)\n"
               "template <int NBW, int CR>\n__device__ __forceinline__ void conv6_prime(u32x4 (&ring)[4][2], unsigned bh, unsigned bl) {")
    first = True
    for nbw in sorted({v[1] for v in VARIANTS}, reverse=True):
        out.append(("    if constexpr (" if first else "    } else if constexpr (") + f"NBW == {nbw}) {{\n" + prime(nbw))
        first = False
    out.append("    }\n}\n")
    return "\n".join(out)


if __name__ == "__main__":
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ppmstereo_amd", "csrc", "conv6_asm.h")
    open(path, "w").write(gen())
    print("wrote", path)
