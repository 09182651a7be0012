#!/usr/bin/env python3
"""Generates ppmstereo_amd/csrc/attn64_asm.h: the hand-scheduled inner loop of mem_attn64_kernel (mem_attn.hip).

One "substep" = one 32-key sub-tile of the 64-key KV tile, for the wave's two 32-query blocks: 32 MFMAs (32 cycles of the matrix pipe
each), and behind them the softmax VALU work of 32 scores per lane (16 pairs: one packed fma, two exp2, one packed add, one bf16
pack each) plus 16 LDS fragment reads.
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


# 1: packed fp32 VALU ops (v_pk_fma_f32 / v_pk_add_f32) for the softmax pairs -- measured SLOWER on gfx950 (1/4-scale call 1.52 ms
# against 1.25 ms with scalar ops: the packed fp32 ops do not hide behind the MFMAs); needs __attribute__((target("packed-fp32-ops")))
# on the kernel, the library being built without packed-fp32 code generation
PK = int(os.environ.get("PPMS_ATTN_PK", "0"))
# fragment ring: RING buffers, a fragment is requested RING - 1 uses ahead; PAIRWAIT: one counted wait per TWO fragments (RING = 6:
# 13 instead of 25 waits per tile).  Round 2 measured 6 no faster than 4 (1.24 vs 1.22 ms); with the softmax denominator in the matrix pipe
# (fewer VALU ops between the waits) it is 1.5 % faster in situ (1.055 vs 1.071 ms per 1/4-scale call, tools/ab_attn_ring.sh): 6 since round 3
RING = int(os.environ.get("PPMS_ATTN_RING", "6"))
PAIRWAIT = RING >= 6
# 1: the softmax denominator is accumulated from the PACKED bf16 probabilities (v_dot2_f32_bf16 with (1, 1): one op per pair instead of
# two fp32 adds) -- correct (15 tests) but measured SLOWER on gfx950 (1.32 vs 1.25 ms: like the packed fp32 ops, the dot op does not
# hide behind the MFMAs), so it stays off
DOT = int(os.environ.get("PPMS_ATTN_DOT", "0"))
# Softmax denominator.  "mfma" (default since round 3): l comes out of the matrix pipe -- one more MFMA per (16 keys, query block) whose A
# operand is all ones, so l = sum of exactly the bf16-rounded probabilities the PV product uses (numerator and denominator then carry the
# SAME rounding and the same accumulation: the rounding error of P cancels to first order wherever the values of a channel share a sign;
# measured on the reference's iters=10 fixture: EPE 7.9e-4 -> see DESIGN.md section 4), and the 32 fp32 adds per substep leave the VALU
# stream.  "add": fp32 adds of the unrounded probabilities (flash-attention's form; rounds 1-2).
# "mfma16" (default): the same sum from ONE 16x16x32 MFMA per (16 keys, query block) -- half the matrix-pipe time of the 32x32x16 form.
# The P fragment (32x32x16 B layout: lane (r, h) = query r, keys 8h..8h+7) read as a 16x16x32 B operand is column r & 15, k-block
# 2 h + (r >> 4); with the constant A operand (row 0: ones on k-blocks 0 and 2, row 1: ones on k-blocks 1 and 3, other rows zero)
# the result's row 0 is the sum over all 16 keys for queries 0..15, row 1 for queries 16..31 (registers 0 and 1 of lanes 0..15).
LSUM = os.environ.get("PPMS_ATTN_LSUM", "mfma16")
assert LSUM in ("mfma", "mfma16", "add")
MF16 = "v_mfma_f32_16x16x32_bf16"
# 1: the first score argument of a pair is formed in the pair's EVEN slot (behind the exp there) instead of both in the odd slot: VALU ops
# per slot 2 / 3 instead of 1 / 4
BAL = int(os.environ.get("PPMS_ATTN_BAL", "0"))
ABL = int(os.environ.get("PPMS_ATTN_ABL", "0"))     # timing experiments only (wrong results): 1 drops the softmax VALU work, 2 the LDS
                                                    # requests and waits, 4 the MFMAs, 8 the address upkeep


class Emit:
    def __init__(self):
        self.lines = []

    def asm(self, text, outs=(), ins=()):
        op = text.split()[0]
        if ((ABL & 1 and op in ("v_exp_f32", "v_fma_f32", "v_add_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_cvt_pk_bf16_f32")) or (ABL & 2 and (op == "ds_read_b128" or "lgkmcnt" in text)) or
                (ABL & 4 and op == MF) or (ABL & 8 and op == "v_add_u32")):
            return
        if ABL & 16 and op == "v_exp_f32":
            text = text.replace("v_exp_f32", "v_mov_b32")
        if ABL & 32 and op in ("v_fma_f32", "v_add_f32", "v_pk_fma_f32", "v_pk_add_f32", "v_cvt_pk_bf16_f32"):
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
    queue = [f"u{i}" for i in range(RING - 1)]

    def wait_for(tag):
        younger = len(queue) - 1 - queue.index(tag)
        E.asm(f"s_waitcnt lgkmcnt({younger})")

    def ring_read(u):                 # u in 0..14: >= 12 -> next substep's unit u - 12
        nxt = u >= 12
        uu = u - 12 if nxt else u
        buf = u % RING
        if uu < 8:                    # K fragment, k-step uu, for S of sub-tile k+1 (own) / k+2 (prefetch)
            imm = (0 if par == 0 else 8192) if nxt else (8192 if par == 0 else 0)
            E.asm(f"ds_read_b128 {{d}}, {{a}} offset:{imm}", [("d", "+v", f"ring[{buf}]")], [("a", "v", f"kaddr[{uu}]")])
        else:                         # V^T fragment, d block uu - 8, keys 0..15 of sub-tile k
            E.asm(f"ds_read_b128 {{d}}, {{a}} offset:{(uu - 8) * 4096}", [("d", "+v", f"ring[{buf}]")], [("a", "v", f"vaddr[{par * 2}]")])
        queue.append(("n" if nxt else "u") + str(uu))

    cons = {0: 0, 2: 1, 4: 2, 6: 3, 16: 4, 18: 5, 20: 6, 22: 7, 24: 8, 26: 9, 28: 10, 30: 11}     # slot -> ring unit consumed there
    for s in range(32):
        # ---- the MFMA of this slot ------------------------------------------------------------------------------------------
        if s in cons and not (PAIRWAIT and cons[s] % 2 == 1):
            wait_for("u" + str(cons[s] + (1 if PAIRWAIT else 0)))
        b = s & 1
        if s < 8 or 16 <= s < 24:
            u = (s >> 1) if s < 8 else 4 + ((s - 16) >> 1)
            if u == 0:
                E.asm(f"{MF} {{c}}, {{a}}, {{b}}, 0", [("c", "=&v", f"nxt[{b}]")], [("a", "v", f"ring[{u % RING}]"), ("b", "v", f"qf[{b}][{u}]")])
            else:
                E.asm(f"{MF} {{c}}, {{a}}, {{b}}, {{c}}", [("c", "+v", f"nxt[{b}]")], [("a", "v", f"ring[{u % RING}]"), ("b", "v", f"qf[{b}][{u}]")])
        elif s < 16:
            dblk = (s - 8) >> 1
            E.asm(f"{MF} {{c}}, {{a}}, {{b}}, {{c}}", [("c", "+a", f"o[{dblk}][{b}]")], [("a", "v", f"vh1[{dblk}]"), ("b", "v", f"pf1[{b}]")])
            if LSUM == "mfma" and s >= 14:      # behind the group's last two MFMAs: l += 1 * P over the same 16 keys (one per query block)
                E.asm(f"{MF} {{c}}, {{a}}, {{b}}, {{c}}", [("c", "+a", f"lacc[{b}]")], [("a", "v", "ones"), ("b", "v", f"pf1[{b}]")])
            if LSUM == "mfma16" and s >= 14:
                E.asm(f"{MF16} {{c}}, {{a}}, {{b}}, {{c}}", [("c", "+a", f"lacc4[{b}]")], [("a", "v", "ones"), ("b", "v", f"pf1[{b}]")])
        else:
            u = 8 + ((s - 24) >> 1)
            E.asm(f"{MF} {{c}}, {{a}}, {{b}}, {{c}}", [("c", "+a", f"o[{u - 8}][{b}]")], [("a", "v", f"ring[{u % RING}]"), ("b", "v", f"pf0[{b}]")])
            if LSUM == "mfma" and s >= 30:
                E.asm(f"{MF} {{c}}, {{a}}, {{b}}, {{c}}", [("c", "+a", f"lacc[{b}]")], [("a", "v", "ones"), ("b", "v", f"pf0[{b}]")])
            if LSUM == "mfma16" and s >= 30:
                E.asm(f"{MF16} {{c}}, {{a}}, {{b}}, {{c}}", [("c", "+a", f"lacc4[{b}]")], [("a", "v", "ones"), ("b", "v", f"pf0[{b}]")])
        # ---- LDS requests behind the second MFMA of a fragment (the dedicated V^T fragments first: they stay older than every
        #      request made for the next substep) -------------------------------------------------------------------------------------
        if s in (17, 19, 21, 23):
            dblk = (s - 17) >> 1
            E.asm(f"ds_read_b128 {{d}}, {{a}} offset:{dblk * 4096}", [("d", "+v", f"vh1[{dblk}]")], [("a", "v", f"vaddr[{par * 2 + 1}]")])
            queue.append(f"h{dblk}")
        if (s - 1) in cons:
            ring_read(cons[s - 1] + RING - 1)
        # ---- VALU, in PAIRS of scores (registers 2p, 2p + 1 of a tile are two consecutive keys of one query): pair p owns slots 2p
        #      and 2p + 1.  even slot: exp of pair p + 1's first score, sum of pair p.  odd slot: arguments of pair p + 2 (first: an fma
        #      feeding an exp within two instructions costs a wait state), exp of pair p + 1's second score, bf16 pack of pair p.
        p = s >> 1
        half, pb, g = elem(2 * p)
        if p < 14:
            _, lb, lg = elem(2 * (p + 2))
            tile = f"cur[{lb}]"
        else:
            lb, lg, tile = 0, 2 * (p - 14), "nxt[0]"

        def arg(j):
            E.asm("v_fma_f32 {t}, {x}, {sc}, {m}", [("t", "=v", f"tt2[{p & 1}][{j}]")],
                  [("x", "v", f"{tile}[{lg + j}]"), ("sc", "v", "scale2[0]"), ("m", "v", f"negm2[{lb}][0]")])

        pf_word = f"pf{half}[{pb}][{(g & 7) >> 1}]"
        if s % 2 == 0:
            E.asm("v_exp_f32 {p}, {t}", [("p", "=v", f"pt2[{(p + 1) & 1}][0]")], [("t", "v", f"tt2[{(p + 1) & 1}][0]")])
            if BAL and not PK:
                arg(0)
            if PK:
                E.asm("v_pk_add_f32 {l}, {l}, {p}", [("l", "+v", f"lsum2[{pb}]")], [("p", "v", f"pt2[{p & 1}]")])
            elif DOT or LSUM != "add":
                pass                                   # (the pair is summed from its packed bf16 form: odd slot / the matrix pipe)
            else:
                for j in range(2):
                    E.asm("v_add_f32 {l}, {l}, {p}", [("l", "+v", f"lsum2[{pb}][{j}]")], [("p", "v", f"pt2[{p & 1}][{j}]")])
        else:
            if PK:
                E.asm("v_pk_fma_f32 {t}, {x}, {sc}, {m}", [("t", "=v", f"tt2[{p & 1}]")],
                      [("x", "v", f"__builtin_shufflevector({tile}, {tile}, {lg}, {lg + 1})"), ("sc", "v", "scale2"), ("m", "v", f"negm2[{lb}]")])
            else:
                if not BAL:
                    arg(0)
                arg(1)
            E.asm("v_exp_f32 {p}, {t}", [("p", "=v", f"pt2[{(p + 1) & 1}][1]")], [("t", "v", f"tt2[{(p + 1) & 1}][1]")])
            E.asm("v_cvt_pk_bf16_f32 {d}, {p0}, {p1}", [("d", "+v", pf_word)], [("p0", "v", f"pt2[{p & 1}][0]"), ("p1", "v", f"pt2[{p & 1}][1]")])
            if DOT:                                    # l += P0 + P1 of the bf16 values the PV product uses: one op per pair
                E.asm("v_dot2_f32_bf16 {l}, {w}, {one}, {l}", [("l", "+v", f"lsum2[{pb}][0]")], [("w", "v", pf_word), ("one", "s", "0x3f803f80u")])
        # ---- address upkeep: K addresses move to the next stage once this sub-tile's own K requests are out (even substeps);
        #      V addresses after the tile's last V request (odd substeps) ---------------------------------------------------------
        if par == 0 and 18 <= s < 26:
            E.asm("v_add_u32 {a}, {a}, {dl}", [("a", "+v", f"kaddr[{s - 18}]")], [("dl", "s", "delta")])
        if par == 1 and 26 <= s < 30:
            E.asm("v_add_u32 {a}, {a}, {dl}", [("a", "+v", f"vaddr[{s - 26}]")], [("dl", "s", "delta")])
    if par == 1:      # the tile's KV stage returns to the DMA ring at the barrier behind this substep: its last LDS reads (h3) must have landed
        wait_for("h3")
    assert queue[-(RING - 1):] == [f"n{i}" for i in range(RING - 1)], queue
    return "\n".join(E.lines)


SIG = ("f32x16 (&cur)[2], f32x16 (&nxt)[2], const bf16x8 (&qf)[2][8], f32x16 (&o)[4][2], u32x4 (&ring)[ATT_RING], u32x4 (&vh1)[4],\n"
       "        u32x4 (&pf0)[2], u32x4 (&pf1)[2], f32x2 (&pt2)[2], f32x2 (&tt2)[2], f32x2 (&lsum2)[2], f32x16 (&lacc)[2], f32x4 (&lacc4)[2], const u32x4& ones,\n"
       "        const f32x2 (&negm2)[2], f32x2 scale2, unsigned (&kaddr)[8], unsigned (&vaddr)[4], int delta")


def gen():
    out = ['''// GENERATED by tools/gen_attn_asm.py -- do not edit.  (Schedule and register roles: see the generator's docstring.)
#pragma once
constexpr int ATT_RING = %d;        // K / V^T fragment buffers in registers
constexpr int ATT_LSUM = %d;        // softmax denominator: 0 = fp32 adds (lsum2), 1 = 32x32x16 ones-row MFMA (lacc), 2 = 16x16x32 selector MFMA (lacc4)
''' % (RING, {"add": 0, "mfma": 1, "mfma16": 2}[LSUM])]
    out.append(f"template <int PAR>\n__device__ __forceinline__ void attn64_substep({SIG}) {{")
    out.append("    if constexpr (PAR == 0) {\n" + substep(0) + "\n    } else {\n" + substep(1) + "\n    }\n}\n")
    # prime: ring units 0..2 of substep 0 (S of sub-tile 1: tile 0, keys 32..63), arguments of pairs 0 and 1, exps of pair 0
    E = Emit()
    for u in range(RING - 1):
        E.asm("ds_read_b128 {d}, {a} offset:8192", [("d", "+v", f"ring[{u}]")], [("a", "v", f"kaddr[{u}]")])
    for pr in range(2):
        for j in range(2):
            E.asm("v_fma_f32 {t}, {x}, {sc}, {m}", [("t", "=v", f"tt2[{pr}][{j}]")], [("x", "v", f"cur[0][{2 * pr + j}]"), ("sc", "v", "scale2[0]"), ("m", "v", "negm2[0][0]")])
    for j in range(2):
        E.asm("v_exp_f32 {p}, {t}", [("p", "=v", f"pt2[0][{j}]")], [("t", "v", f"tt2[0][{j}]")])
    out.append("__device__ __forceinline__ void attn64_prime(f32x16 (&cur)[2], u32x4 (&ring)[ATT_RING], f32x2 (&pt2)[2], f32x2 (&tt2)[2], const f32x2 (&negm2)[2], f32x2 scale2,\n"
               "                                             unsigned (&kaddr)[8]) {\n" + "\n".join(E.lines) + "\n}\n")
    # tail: drains the LDS queue.  (The O += V P group of the last sub-tile's keys 16..31 that follows is written with MFMA builtins
    # in mem_attn.hip: outside the loop the register allocator moves accumulator tuples around with v_accvgpr_* copies, and it pads
    # wait states only around MFMAs it can see.)
    E = Emit()
    E.asm("s_waitcnt lgkmcnt(0)")
    E.asm("s_nop 1")
    out.append("__device__ __forceinline__ void attn64_tail() {\n" + "\n".join(E.lines) + "\n}\n")
    return "\n".join(out)


if __name__ == "__main__":
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ppmstereo_amd", "csrc", "attn64_asm.h")
    open(path, "w").write(gen())
    print("wrote", path)
