#!/usr/bin/env python3
"""Generates ppmstereo_amd/csrc/attn64_asm.h: the hand-scheduled inner loop of mem_attn64_kernel (mem_attn.hip), built on the
16x16x32 bf16 MFMA.

Why this shape: under dense MFMA load on random data the part holds a higher clock with v_mfma_f32_16x16x32_bf16 than with
v_mfma_f32_32x32x16_bf16 at equal cycles per FLOP (tools/probe/mfma_shape_probe.hip on the attention's 128 x 64 O^T tile per wave:
2065 vs 1854 TFLOP/s with the operands in registers, 1720 vs 1576 with every fragment re-read from LDS; on zeros both run at 2.39 GHz).

One "substep" = one 32-key sub-tile of the 64-key KV tile for the wave's four 16-query blocks: 64 MFMAs of 16 cycles (+ 4 for the
softmax denominator) and behind them the softmax VALU work of 32 scores per lane (16 pairs: two fma, two exp2, one bf16 pack each)
plus 16 LDS fragment reads.  Every instruction is its own `asm volatile` statement: the lists below ARE the issue order.

  S^T block (b, qb) = K rows of key block b (16 x 128) times Q^T of query block qb: 4 k-steps of 32 channels.  K row m of block b is
  key 8 (m >> 2) + 4 b + (m & 3) of the sub-tile, so lane (query c, g) holds keys 8g..8g+3 in block 0 and 8g+4..8g+7 in block 1:
  packed to bf16 these are the B operand (k-block g = 8 consecutive keys) of O^T += V^T P for ALL 32 keys of the sub-tile.

  MFMA slots   0..15   S_{k+1} block row 0: K fragment (0, s) x 4 query blocks, s = 0..3          VALU  slots  0..15  P of query block 0
              16..23   O^T[:, qb 3] += V^T P  of sub-tile k-1 (8 d blocks), then its denominator           16..31  query block 1
              24..39   S_{k+1} block row 1                                                                32..47  query block 2
              40..47   O^T[:, qb 0] of sub-tile k;  48..55  qb 1;  56..63  qb 2  (each + denominator)      48..63  query block 3
  (pair p owns slots 4p..4p+3: exp of pair p+1, arguments of pair p+2, bf16 pack of pair p.)

  K fragments rotate through four 16-B register buffers, requested three uses ahead; the eight V^T fragments of the sub-tile are read once
  (slots 24..31) into their own registers and serve the four O^T groups, the last of which runs in the next substep -- i.e. no LDS read
  touches a KV stage after the barrier that hands it back to the DMA ring.
"""
import os

MF = "v_mfma_f32_16x16x32_bf16"          # S^T = K' Q^T: bf16 operands, as the reference casts them (ppmstereo.py:541-550)
MF_P = {False: "v_mfma_f32_16x16x32_bf16", True: "v_mfma_f32_16x16x32_f16"}      # O^T += V^T P~ and l += 1 P~: the format of P~ (and of the V^T image)
CVT_P = {False: "v_cvt_pk_bf16_f32", True: "v_cvt_pk_f16_f32"}
RINGK = 4
ABL = int(os.environ.get("PPMS_ATTN_ABL", "0"))     # timing experiments only (wrong results): 1 drops the softmax VALU work, 2 the LDS
                                                    # requests and waits, 4 the MFMAs, 8 the address upkeep
SACC = "a" if ABL & 128 else "v"                    # (128, with 1: the S^T accumulators in the AGPR half -- what would the VGPR placement cost?)
DSLOT = [int(x) for x in os.environ.get("PPMS_ATTN_DSLOT", "18,22,50,58").split(",")]
CUNIT = [0, 4, 8, 12, 24, 28, 32, 36]               # slot at which K unit u = (block row u >> 2, k-step u & 3) is first consumed


class Emit:
    def __init__(self):
        self.lines = []

    def asm(self, text, outs=(), ins=()):
        op = text.split()[0]
        if ((ABL & 1 and op in ("v_exp_f32", "v_fma_f32", "v_cvt_pk_bf16_f32", "v_cvt_pk_f16_f32")) or (ABL & 2 and (op == "ds_read_b128" or "lgkmcnt" in text)) or
                (ABL & 4 and op.startswith("v_mfma")) or (ABL & 8 and op == "v_add_u32")):
            return
        ops = list(outs) + list(ins)
        for i, (nm, _, _) in enumerate(ops):
            text = text.replace("{" + nm + "}", "%" + str(i))
        o = ", ".join(f'"{c}"({e})' for _, c, e in outs)
        i = ", ".join(f'"{c}"({e})' for _, c, e in ins)
        self.lines.append(f'    asm volatile("{text}" : {o} : {i} : "memory");' if (o or i) else f'    asm volatile("{text}" ::: "memory");')


def pair(p):
    """pair p of a substep (p >= 16: pair p - 16 of the next one) -> (tile, qb, word of the P fragment, block row, first register)"""
    tile = "cur" if p < 16 else "nxt"
    p &= 15
    qb, w = p >> 2, p & 3
    return tile, qb, w, w >> 1, 2 * (w & 1)


def substep(par, p16):
    MFP, CVT = MF_P[p16], CVT_P[p16]
    E = Emit()
    queue = [f"u{i}" for i in range(RINGK - 1)]          # LDS requests in flight at entry, oldest first (the LDS returns in order)

    def wait_for(tag):
        i = queue.index(tag)
        E.asm(f"s_waitcnt lgkmcnt({len(queue) - 1 - i})")
        del queue[:i + 1]

    def k_read(u):                # u in 0..10: >= 8 -> next substep's unit u - 8
        nxt = u >= 8
        uu = u - 8 if nxt else u
        # even substeps compute S of the tile's second sub-tile (+ 8192 B) and prefetch for the next tile's first (the addresses have moved
        # to the next stage by then); odd substeps the other way round
        imm = (0 if par == 0 else 8192) if nxt else (8192 if par == 0 else 0)
        E.asm(f"ds_read_b128 {{d}}, {{a}} offset:{imm}", [("d", "+v", f"ring[{u % RINGK}]")], [("a", "v", f"kaddr[{uu}]")])
        queue.append(("n" if nxt else "u") + str(uu))

    for s in range(64):
        # ---- the MFMA of this slot ------------------------------------------------------------------------------------------
        if s in CUNIT:
            wait_for("u" + str(CUNIT.index(s)))
        if 40 <= s < 48 and (s & 1) == 0:
            wait_for(f"v{s - 40 + 1}")                     # one wait per two V^T fragments
        if s < 16 or 24 <= s < 40:
            u = (s >> 2) if s < 16 else 4 + ((s - 24) >> 2)
            b, ks, qb = u >> 2, u & 3, s & 3
            if ks == 0:
                E.asm(f"{MF} {{c}}, {{a}}, {{b}}, 0", [("c", "=&" + SACC, f"nxt[{b}][{qb}]")], [("a", "v", f"ring[{u % RINGK}]"), ("b", "v", f"qf[{qb}][{ks}]")])
            else:
                E.asm(f"{MF} {{c}}, {{a}}, {{b}}, {{c}}", [("c", "+" + SACC, f"nxt[{b}][{qb}]")], [("a", "v", f"ring[{u % RINGK}]"), ("b", "v", f"qf[{qb}][{ks}]")])
        else:
            qb, d = (3, s - 16) if s < 24 else (0, s - 40) if s < 48 else (1, s - 48) if s < 56 else (2, s - 56)
            E.asm(f"{MFP} {{c}}, {{a}}, {{b}}, {{c}}", [("c", "+a", f"o[{d}][{qb}]")], [("a", "v", f"vt[{d}]"), ("b", "v", f"pf[{qb}]")])
            if d == 7:            # the denominator of the same 32 keys: l += 1 * P (A operand all ones: every row of the result is the sum)
                E.asm(f"{MFP} {{c}}, {{a}}, {{b}}, {{c}}", [("c", "+a", f"lacc[{qb}]")], [("a", "v", "ones"), ("b", "v", f"pf[{qb}]")])
        # ---- LDS requests ------------------------------------------------------------------------------------------------------
        if (s - 1) in CUNIT:
            k_read(CUNIT.index(s - 1) + RINGK - 1)
        if 24 <= s < 32:
            d = s - 24
            E.asm(f"ds_read_b128 {{d}}, {{a}} offset:{d * 2048}", [("d", "+v", f"vt[{d}]")], [("a", "v", f"vaddr[{par}]")])
            queue.append(f"v{d}")
        # ---- LDS-DMA of the tile three ahead: K chunks in even substeps, V^T chunks in odd ones, one instruction per ~16 slots (issued in a
        #      burst at the top of the iteration -- no MFMA in flight behind the barrier -- the eight of them cost 6.5 % of the loop)
        if s in DSLOT:
            i = DSLOT.index(s)
            E.asm("s_mov_b32 m0, {m}\\n\\ts_nop 0\\n\\tglobal_load_lds_dwordx4 {o}, {p}", [],
                  [("o", "v", f"{'koff' if par == 0 else 'voff'}[{i}]"), ("p", "s", "kp" if par == 0 else "vp"), ("m", "s", f"dst[{par * 4 + i}]")])
        # ---- VALU: pair p owns slots 4p .. 4p+3 -------------------------------------------------------------------------------------
        p, ph = s >> 2, s & 3
        if ph in (0, 2):
            j = ph >> 1
            E.asm("v_exp_f32 {p}, {t}", [("p", "=v", f"pt[{(p + 1) & 1}][{j}]")], [("t", "v", f"tt[{(p + 1) & 1}][{j}]")])
        else:
            j = ph >> 1
            tile, qb, _, b, r0 = pair(p + 2)
            E.asm("v_fma_f32 {t}, {x}, {sc}, {m}", [("t", "=v", f"tt[{p & 1}][{j}]")],
                  [("x", "v", f"{tile}[{b}][{qb}][{r0 + j}]"), ("sc", "v", "scale"), ("m", "v", f"negm[{qb}]")])
            if ph == 3:
                _, qb, w, _, _ = pair(p)
                E.asm(CVT + " {d}, {p0}, {p1}", [("d", "+v", f"pf[{qb}][{w}]")], [("p0", "v", f"pt[{p & 1}][0]"), ("p1", "v", f"pt[{p & 1}][1]")])
        # ---- address upkeep: the K addresses move to the next stage once the sub-tile's own K requests are out (even substeps); the V^T
        #      addresses after the tile's last V^T request (odd substeps) ---------------------------------------------------------------
        if par == 0 and 26 <= s < 34:
            E.asm("v_add_u32 {a}, {a}, {dl}", [("a", "+v", f"kaddr[{s - 26}]")], [("dl", "s", "delta")])
        if par == 1 and 32 <= s < 34:
            E.asm("v_add_u32 {a}, {a}, {dl}", [("a", "+v", f"vaddr[{s - 32}]")], [("dl", "s", "delta")])
    assert queue == [f"n{i}" for i in range(1, RINGK - 1)], queue      # (n0 landed in front of the last V^T fragment; the next substep assumes <= 3 in flight)
    return "\n".join(E.lines)


SIG = ("f32x4 (&cur)[2][4], f32x4 (&nxt)[2][4], const bf16x8 (&qf)[4][4], f32x4 (&o)[8][4], u32x4 (&ring)[4], u32x4 (&vt)[8], u32x4 (&pf)[4],\n"
       "        f32x2 (&pt)[2], f32x2 (&tt)[2], f32x4 (&lacc)[4], const u32x4& ones, const float (&negm)[4], float scale, unsigned (&kaddr)[8],\n"
       "        unsigned (&vaddr)[2], int delta, const unsigned (&koff)[4], const unsigned (&voff)[4], const char* kp, const char* vp, const unsigned (&dst)[8]")


def gen():
    out = ["// GENERATED by tools/gen_attn_asm.py -- do not edit.  (Schedule and register roles: see the generator's docstring.)\n#pragma once\n"]
    # P16: P~ (and the V^T image, and the all-ones row of the denominator) in fp16 instead of bf16 -- 11 instead of 8 significand bits in the
    # P~ V product at the same MFMA rate; the kernel then tolerates 2^15 of overshoot above the softmax reference instead of 2^60 (mem_attn.hip)
    out.append(f"template <int PAR, bool P16>\n__device__ __forceinline__ void attn64_substep({SIG}) {{")
    out.append("    if constexpr (PAR == 0 && !P16) {\n" + substep(0, False) + "\n    } else if constexpr (PAR == 1 && !P16) {\n" + substep(1, False) +
               "\n    } else if constexpr (PAR == 0) {\n" + substep(0, True) + "\n    } else {\n" + substep(1, True) + "\n    }\n}\n")
    # prime: K units 0..2 of substep 0 (S of sub-tile 1: tile 0, keys 32..63), arguments of pairs 0 and 1, exps of pair 0
    E = Emit()
    for u in range(RINGK - 1):
        E.asm("ds_read_b128 {d}, {a} offset:8192", [("d", "+v", f"ring[{u}]")], [("a", "v", f"kaddr[{u}]")])
    for pr in range(2):
        _, qb, _, b, r0 = pair(pr)
        for j in range(2):
            E.asm("v_fma_f32 {t}, {x}, {sc}, {m}", [("t", "=v", f"tt[{pr}][{j}]")], [("x", "v", f"cur[{b}][{qb}][{r0 + j}]"), ("sc", "v", "scale"), ("m", "v", f"negm[{qb}]")])
    for j in range(2):
        E.asm("v_exp_f32 {p}, {t}", [("p", "=v", f"pt[0][{j}]")], [("t", "v", f"tt[0][{j}]")])
    out.append("__device__ __forceinline__ void attn64_prime(f32x4 (&cur)[2][4], u32x4 (&ring)[4], f32x2 (&pt)[2], f32x2 (&tt)[2], const float (&negm)[4], float scale,\n"
               "                                              unsigned (&kaddr)[8]) {\n" + "\n".join(E.lines) + "\n}\n")
    E = Emit()
    E.asm("s_waitcnt lgkmcnt(0)")
    E.asm("s_nop 7")
    out.append("__device__ __forceinline__ void attn64_tail() {\n" + "\n".join(E.lines) + "\n}\n")
    return "\n".join(out)


if __name__ == "__main__":
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "ppmstereo_amd", "csrc", "attn64_asm.h")
    open(path, "w").write(gen())
    print("wrote", path)
