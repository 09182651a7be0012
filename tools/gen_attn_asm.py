#!/usr/bin/env python3
"""Generates ppmstereo_amd/csrc/attn64_asm.h: the hand-scheduled inner loop of mem_attn64_kernel (mem_attn.hip).

One "substep" = one 32-key sub-tile of the 64-key KV tile, for the wave's two 32-query blocks: 32 MFMAs (32 cycles of the matrix pipe
each), and in their shadow the softmax VALU work of 32 scores per lane (fma, exp2, add, bf16 pack) plus 16 LDS fragment reads.
Every instruction is its own `asm volatile` statement (never reordered among themselves): the lists below ARE the issue order, the
compiler only allocates registers.  Schedule of substep k (slot = one MFMA and what issues behind it):

  MFMA   slots  0.. 7   S_{k+1} += K Q^T, k-steps 0..3           VALU  slots  0..15  P = exp2(S_k * scale - m), keys  0..15 of the sub-tile
         slots  8..15   O += V P   for keys 16..31 of sub-tile k-1      slots 16..31  the same for keys 16..31
         slots 16..23   S_{k+1}, k-steps 4..7
         slots 24..31   O += V P   for keys 0..15 of sub-tile k

so every consumer sits >= 8 slots behind its producer (S -> exp: MFMA result to VALU; packed P -> MFMA) and the matrix pipe never waits
for the VALU.  K / V^T fragments rotate through four 16-B register buffers, requested three uses ahead; the V^T fragments of the
"previous sub-tile" group are read into their own registers early (slots 17..23), i.e. before the barrier that hands the KV stage
back to the LDS-DMA ring.
"""
import os

MF = "v_mfma_f32_32x32x16_bf16"


ABL = int(os.environ.get("PPMS_ATTN_ABL", "0"))     # timing experiments only (wrong results): 1 drops the softmax VALU work, 2 the LDS
                                                    # requests and waits, 4 the MFMAs, 8 the address upkeep


class Emit:
    def __init__(self):
        self.lines = []

    def asm(self, text, outs=(), ins=()):
        op = text.split()[0]
        if ((ABL & 1 and op in ("v_exp_f32", "v_fma_f32", "v_add_f32", "v_cvt_pk_bf16_f32")) or (ABL & 2 and (op == "ds_read_b128" or "lgkmcnt" in text)) or
                (ABL & 4 and op == MF) or (ABL & 8 and op == "v_add_u32")):
            return
        if ABL & 16 and op == "v_exp_f32":
            text = text.replace("v_exp_f32", "v_mov_b32")
        if ABL & 32 and op in ("v_fma_f32", "v_add_f32", "v_cvt_pk_bf16_f32"):
            return
        ops = list(outs) + list(ins)
        for i, (nm, _, _) in enumerate(ops):
            text = text.replace("{" + nm + "}", "%" + str(i))
        o = ", ".join(f'"{c}"({e})' for _, c, e in outs)
        i = ", ".join(f'"{c}"({e})' for _, c, e in ins)
        self.lines.append(f'    asm volatile("{text}" : {o} : {i} : "memory");' if (o or i) else f'    asm volatile("{text}" ::: "memory");')


def elem(e):
    """element e of a substep -> (half, b, g): half 0 = keys 0..15 (S^T registers 0..7), half 1 = keys 16..31 (registers 8..15)"""
    half, idx = e >> 4, e & 15
    b, g8 = idx >> 3, idx & 7
    return half, b, half * 8 + g8


def substep(par):
    E = Emit()
    # ---- LDS queue simulation: ids in issue order; entry state = this substep's ring units 0, 1, 2 in flight -------------------
    queue = ["u0", "u1", "u2"]

    def wait_for(tag):
        younger = len(queue) - 1 - queue.index(tag)
        E.asm(f"s_waitcnt lgkmcnt({younger})")

    def ring_read(u):                 # u in 0..14: >= 12 -> next substep's unit u - 12
        nxt = u >= 12
        uu = u - 12 if nxt else u
        buf = u % 4
        if uu < 8:                    # K fragment, k-step uu, for S of sub-tile k+1 (own) / k+2 (prefetch)
            imm = (0 if par == 0 else 8192) if nxt else (8192 if par == 0 else 0)
            E.asm(f"ds_read_b128 {{d}}, {{a}} offset:{imm}", [("d", "+v", f"ring[{buf}]")], [("a", "v", f"kaddr[{uu}]")])
        else:                         # V^T fragment, d block uu - 8, keys 0..15 of sub-tile k
            E.asm(f"ds_read_b128 {{d}}, {{a}} offset:{(uu - 8) * 4096}", [("d", "+v", f"ring[{buf}]")], [("a", "v", f"vaddr[{par * 2}]")])
        queue.append(("n" if nxt else "u") + str(uu))

    cons = {0: 0, 2: 1, 4: 2, 6: 3, 16: 4, 18: 5, 20: 6, 22: 7, 24: 8, 26: 9, 28: 10, 30: 11}     # slot -> ring unit consumed there
    for s in range(32):
        # ---- the MFMA of this slot ------------------------------------------------------------------------------------------
        if s in cons:
            wait_for("u" + str(cons[s]))
        b = s & 1
        if s < 8 or 16 <= s < 24:
            u = (s >> 1) if s < 8 else 4 + ((s - 16) >> 1)
            if u == 0:
                E.asm(f"{MF} {{c}}, {{a}}, {{b}}, 0", [("c", "=&v", f"nxt[{b}]")], [("a", "v", f"ring[{u % 4}]"), ("b", "v", f"qf[{b}][{u}]")])
            else:
                E.asm(f"{MF} {{c}}, {{a}}, {{b}}, {{c}}", [("c", "+v", f"nxt[{b}]")], [("a", "v", f"ring[{u % 4}]"), ("b", "v", f"qf[{b}][{u}]")])
        elif s < 16:
            dblk = (s - 8) >> 1
            E.asm(f"{MF} {{c}}, {{a}}, {{b}}, {{c}}", [("c", "+a", f"o[{dblk}][{b}]")], [("a", "v", f"vh1[{dblk}]"), ("b", "v", f"pf1[{b}]")])
        else:
            u = 8 + ((s - 24) >> 1)
            E.asm(f"{MF} {{c}}, {{a}}, {{b}}, {{c}}", [("c", "+a", f"o[{u - 8}][{b}]")], [("a", "v", f"ring[{u % 4}]"), ("b", "v", f"pf0[{b}]")])
        # ---- LDS requests behind the second MFMA of a fragment ------------------------------------------------------------------
        if (s - 1) in cons:
            ring_read(cons[s - 1] + 3)
        if s in (17, 19, 21, 23):
            dblk = (s - 17) >> 1
            E.asm(f"ds_read_b128 {{d}}, {{a}} offset:{dblk * 4096}", [("d", "+v", f"vh1[{dblk}]")], [("a", "v", f"vaddr[{par * 2 + 1}]")])
            queue.append(f"h{dblk}")
        # ---- VALU: exp of element s + 1 (its argument was formed one slot earlier: an fma feeding the exp directly costs a wait
        #      state), argument of element s + 2, sum of element s, pack of the pair before ---------------------------------------
        E.asm("v_exp_f32 {p}, {t}", [("p", "=v", f"pt[{(s + 1) & 3}]")], [("t", "v", f"tt[{(s + 1) & 1}]")])
        if s < 30:
            _, lb, lg = elem(s + 2)
            src = f"cur[{lb}][{lg}]"
        else:
            lb, src = 0, f"nxt[0][{s - 30}]"
        E.asm("v_fma_f32 {t}, {x}, {sc}, {m}", [("t", "=v", f"tt[{s & 1}]")], [("x", "v", src), ("sc", "s", "scale"), ("m", "v", f"negm[{lb}]")])
        _, cb, _ = elem(s)
        E.asm("v_add_f32 {l}, {l}, {p}", [("l", "+v", f"lsum[{cb}]")], [("p", "v", f"pt[{s & 3}]")])
        if s % 2 == 0:                # pack the pair (s - 2, s - 1); slot 0: the previous substep's last pair
            e1 = (s - 1) % 32
            half, pb, g = elem(e1)
            w = (g & 7) >> 1
            E.asm("v_cvt_pk_bf16_f32 {d}, {p0}, {p1}", [("d", "+v" if True else "=v", f"pf{half}[{pb}][{w}]")],
                  [("p0", "v", f"pt[{(s - 2) & 3}]"), ("p1", "v", f"pt[{(s - 1) & 3}]")])
        # ---- address upkeep: K addresses move to the next stage once this sub-tile's own K requests are out (even substeps);
        #      V addresses after the tile's last V request (odd substeps) ---------------------------------------------------------
        if par == 0 and 18 <= s < 26:
            E.asm("v_add_u32 {a}, {a}, {dl}", [("a", "+v", f"kaddr[{s - 18}]")], [("dl", "s", "delta")])
        if par == 1 and 26 <= s < 30:
            E.asm("v_add_u32 {a}, {a}, {dl}", [("a", "+v", f"vaddr[{s - 26}]")], [("dl", "s", "delta")])
    assert queue[-3:] == ["n0", "n1", "n2"], queue
    return "\n".join(E.lines)


SIG = ("f32x16 (&cur)[2], f32x16 (&nxt)[2], const bf16x8 (&qf)[2][8], f32x16 (&o)[4][2], u32x4 (&ring)[4], u32x4 (&vh1)[4],\n"
       "        u32x4 (&pf0)[2], u32x4 (&pf1)[2], float (&pt)[4], float (&tt)[2], float (&lsum)[2], const float (&negm)[2], float scale,\n"
       "        unsigned (&kaddr)[8], unsigned (&vaddr)[4], int delta")


def gen():
    out = ['''// GENERATED by tools/gen_attn_asm.py -- do not edit.  (Schedule and register roles: see the generator's docstring.)
#pragma once
''']
    out.append(f"template <int PAR>\n__device__ __forceinline__ void attn64_substep({SIG}) {{")
    out.append("    if constexpr (PAR == 0) {\n" + substep(0) + "\n    } else {\n" + substep(1) + "\n    }\n}\n")
    # prime: ring units 0..2 of substep 0 (S of sub-tile 1: tile 0, keys 32..63), the look-ahead exp of element 0 and argument of element 1
    E = Emit()
    for u in range(3):
        E.asm("ds_read_b128 {d}, {a} offset:8192", [("d", "+v", f"ring[{u}]")], [("a", "v", f"kaddr[{u}]")])
    E.asm("v_fma_f32 {t}, {x}, {sc}, {m}", [("t", "=v", "tt[0]")], [("x", "v", "cur[0][0]"), ("sc", "s", "scale"), ("m", "v", "negm[0]")])
    E.asm("v_fma_f32 {t}, {x}, {sc}, {m}", [("t", "=v", "tt[1]")], [("x", "v", "cur[0][1]"), ("sc", "s", "scale"), ("m", "v", "negm[0]")])
    E.asm("v_exp_f32 {p}, {t}", [("p", "=v", "pt[0]")], [("t", "v", "tt[0]")])
    out.append("__device__ __forceinline__ void attn64_prime(f32x16 (&cur)[2], u32x4 (&ring)[4], float (&pt)[4], float (&tt)[2], const float (&negm)[2], float scale,\n"
               "                                             unsigned (&kaddr)[8]) {\n" + "\n".join(E.lines) + "\n}\n")
    # tail: the last pair's pack; drains the LDS queue.  (The O += V P group of the last sub-tile's keys 16..31 that follows is
    # written with MFMA builtins in mem_attn.hip: outside the loop the register allocator moves accumulator tuples around with
    # v_accvgpr_* copies, and it pads wait states only around MFMAs it can see.)
    E = Emit()
    E.asm("v_cvt_pk_bf16_f32 {d}, {p0}, {p1}", [("d", "+v", "pf1[1][3]")], [("p0", "v", "pt[2]"), ("p1", "v", "pt[3]")])
    E.asm("s_waitcnt lgkmcnt(0)")
    E.asm("s_nop 1")
    out.append("__device__ __forceinline__ void attn64_tail(u32x4 (&pf1)[2], float (&pt)[4]) {\n" + "\n".join(E.lines) + "\n}\n")
    return "\n".join(out)


if __name__ == "__main__":
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ppmstereo_amd", "csrc", "attn64_asm.h")
    open(path, "w").write(gen())
    print("wrote", path)
