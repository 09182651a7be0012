// Fused conv epilogue shared by the implicit-GEMM kernels: bias is already added by the caller.
#pragma once
#include "common.h"

// LD = false: the descriptor is known to be a plain STORE without pre_f32 -- the instantiation contains no load at all (see epilogue_row8)
template <bool LD = true>
__device__ __forceinline__ void epilogue_group(const ppms_epilogue& e, const float* vin, int64_t pix, int cl, int hw, bool with_vt = true) {
    // v[0..3]: acc + bias for couts cl..cl+3 (local to this half) at pixel pix.  Everything is predicated (no early
    // exits, no runtime trip counts) so that the caller's loops unroll fully and the accumulators stay in registers.
    const int nv = e.n_valid - cl;            // how many of the 4 are real
    const bool all4 = nv >= 4;
    float y[4];
    float ax[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    float v[4] = {vin[0], vin[1], vin[2], vin[3]};
    if (LD && e.pre_f32 != nullptr && nv > 0) {     // iteration-invariant part of the pre-activation (n_valid is a multiple of 4 here)
        const f32x4 p4 = gld<f32x4>(e.pre_f32 + pix * e.pre_f32_ld + cl);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += p4[j];
    }
    const int kind = LD ? e.kind : PPMS_EPI_STORE;
    if (nv > 0 && (kind == PPMS_EPI_RESID || kind == PPMS_EPI_RH || kind == PPMS_EPI_GRU)) {
        const bf16_t* ah = (const bf16_t*)e.aux_sp.hi + pix * e.aux_sp.ld + cl;
        const bf16_t* al = (const bf16_t*)e.aux_sp.lo + pix * e.aux_sp.ld + cl;
        if (all4) {
            const bf16x4 h4 = gld<bf16x4>(ah), l4 = gld<bf16x4>(al);
#pragma unroll
            for (int j = 0; j < 4; ++j) ax[j] = join_bf16(h4[j], l4[j]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < nv) ax[j] = join_bf16(gld<bf16_t>(ah + j), gld<bf16_t>(al + j));
        }
    }
    if (kind == PPMS_EPI_RESID) {
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = ax[j] + v[j];
        apply_act_n<4>(y, e.act, e.scale);
    } else if (kind == PPMS_EPI_RH) {
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = sigmoid_fast(v[j]) * ax[j];
    } else if (kind == PPMS_EPI_GRU) {
        const float* zp = e.aux_f32 + pix * e.aux_f32_ld + cl;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float z = (j < nv) ? gld<float>(zp + j) : 0.0f;
            y[j] = (1.0f - z) * ax[j] + z * tanh_fast(v[j]);
        }
    } else if (kind == PPMS_EPI_ADDF32) {
        float* op = e.out_f32 + pix * e.out_f32_ld + cl;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            y[j] = 0.0f;
            if (j < nv) gst<float>(op + j, gld<float>(op + j) + v[j]);
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = v[j];
        apply_act_n<4>(y, e.act, e.scale);
    }
    if (kind != PPMS_EPI_ADDF32 && nv > 0) {
        if (e.out_sp.hi != nullptr) {
            bf16_t* oh = (bf16_t*)e.out_sp.hi + pix * e.out_sp.ld + cl;
            bf16_t* ol = (bf16_t*)e.out_sp.lo + pix * e.out_sp.ld + cl;
            bf16x4 h4, l4;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                bf16_t hh, ll;
                split_bf16(y[j], hh, ll);
                h4[j] = hh;
                l4[j] = ll;
            }
            if (all4) {
                gst<bf16x4>(oh, h4);
                gst<bf16x4>(ol, l4);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j < nv) {
                        gst<bf16_t>(oh + j, h4[j]);
                        gst<bf16_t>(ol + j, l4[j]);
                    }
            }
        }
        if (e.out_f32 != nullptr) {
            float* op = e.out_f32 + pix * e.out_f32_ld + cl;
            if (all4) {
                f32x4 o = {y[0], y[1], y[2], y[3]};
                gst<f32x4>(op, o);
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j < nv) gst<float>(op + j, y[j]);
            }
        }
        if (with_vt && e.out_vt != nullptr) {
            const int64_t frame = pix / hw;
            const int64_t rem = pix - frame * hw;
            bf16_t* vp = (bf16_t*)e.out_vt + (frame * e.n_valid + cl) * hw + rem;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < nv) gst<bf16_t>(vp + (int64_t)j * hw, vt_enc(y[j], e.vt_f16));
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------
// Coalesced form (conv_gemm2).  In the MFMA accumulator layout a lane owns ONE pixel and 4 couts per
// register group, so every epilogue load / store of a wave touches 32 different cache lines with 8-16 useful bytes
// each; with the residual / gate / hoisted-share operands that is several thousand line requests per tile and showed
// up as 15-30 % of the short GRU convs.  The kernels therefore transpose 32-pixel x 64-cout accumulator blocks through
// a wave-private LDS patch (stage_write32 / stage_read8) and run the epilogue on 8 consecutive couts of one pixel per
// lane: 8 lanes cover 64 couts of a pixel, i.e. whole 128-B (bf16 planes) / 256-B (fp32) runs per pixel.
constexpr int STG_LD = 68;                        // dwords per staged pixel row (64 couts + 4 pad: conflict-free b128 phases)
constexpr int STG_WAVE = 32 * STG_LD * 4;         // bytes of one wave's staging patch (32 pixels)

// the transposed V operand of the memory attention (bf16 [frame][n_valid][H*W]) is pixel-major: it is written from
// the accumulator layout (lanes = adjacent pixels), before the transposition.  STORE epilogues only.
__device__ __forceinline__ void epilogue_vt4(const ppms_epilogue& e, const float* v, int frame, int rem, int cl, int hw) {
    // (frame, rem) = the pixel's frame and its offset inside the frame: the callers know both (a 64-bit division per call otherwise)
    const int nv = e.n_valid - cl;
    bf16_t* vp = (bf16_t*)e.out_vt + ((int64_t)frame * e.n_valid + cl) * hw + rem;
    float y[4] = {v[0], v[1], v[2], v[3]};
    apply_act_n<4>(y, e.act, e.scale);
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (j < nv) gst<bf16_t>(vp + (int64_t)j * hw, vt_enc(y[j], e.vt_f16));
}

// Row form: v[0..7] = acc + bias for couts cl..cl+7 (local to this half, cl % 8 == 0) at pixel pix.
// ---- row epilogue in two halves: row8_fetch issues the loads of a row's extra operands (pre_f32 share, aux_sp state, GRU gate),
// row8_finish consumes them, applies the epilogue and stores.  A row loop fetches the operands of a GROUP of rows first and finishes them
// afterwards: vmcnt is one in-order counter for loads and stores, so the wait for a row's operands also waits for every store issued
// before those loads -- one memory round trip per wait.  Grouping pays that once per group instead of once per row.
// Both are specialised on the descriptor's CLASS so that an instantiation holds exactly the loads / registers its class needs:
enum {
    EPI_CLS_PLAIN = 0,     // STORE, no pre_f32, no out_vt: no load at all
    EPI_CLS_PRE = 1,       // STORE + pre_f32
    EPI_CLS_AUX = 2,       // RESID / RH without pre_f32: aux_sp
    EPI_CLS_GRU = 3,       // GRU (aux_sp + gate, pre_f32 optional)
    EPI_CLS_ANY = 4,       // everything else (ADDF32, out_vt): all fields checked at run time
    EPI_CLS_AUXPRE = 5     // RESID / RH with pre_f32 (the r gate of the GRU passes with the hoisted input share)
};
__device__ __forceinline__ int epilogue_class(const ppms_epilogue& e) {
    if (e.out_vt != nullptr) return EPI_CLS_ANY;
    if (e.kind == PPMS_EPI_STORE) return e.pre_f32 != nullptr ? EPI_CLS_PRE : EPI_CLS_PLAIN;
    if (e.kind == PPMS_EPI_RESID || e.kind == PPMS_EPI_RH) return e.pre_f32 == nullptr ? EPI_CLS_AUX : EPI_CLS_AUXPRE;
    if (e.kind == PPMS_EPI_GRU) return EPI_CLS_GRU;
    return EPI_CLS_ANY;
}
__device__ __forceinline__ bool epilogue_is_plain(const ppms_epilogue& e) { return epilogue_class(e) == EPI_CLS_PLAIN; }

struct row8_aux {
    f32x4 p0, p1, z0, z1;
    bf16x8 h8, l8;
};
template <int CLS>
__device__ __forceinline__ void row8_fetch(const ppms_epilogue& e, int64_t pix, int cl, row8_aux& a) {
    if (CLS == EPI_CLS_PLAIN || e.n_valid - cl < 8) return;       // (ragged tail rows load inside row8_finish)
    if ((CLS == EPI_CLS_PRE || CLS == EPI_CLS_GRU || CLS == EPI_CLS_ANY || CLS == EPI_CLS_AUXPRE) && e.pre_f32 != nullptr) {
        const float* pp = e.pre_f32 + pix * e.pre_f32_ld + cl;
        a.p0 = gld<f32x4>(pp), a.p1 = gld<f32x4>(pp + 4);
    }
    const int kind = e.kind;
    if (CLS == EPI_CLS_AUX || CLS == EPI_CLS_AUXPRE || CLS == EPI_CLS_GRU ||
        (CLS == EPI_CLS_ANY && (kind == PPMS_EPI_RESID || kind == PPMS_EPI_RH || kind == PPMS_EPI_GRU))) {
        a.h8 = gld<bf16x8>((const bf16_t*)e.aux_sp.hi + pix * e.aux_sp.ld + cl);
        a.l8 = gld<bf16x8>((const bf16_t*)e.aux_sp.lo + pix * e.aux_sp.ld + cl);
    }
    if (CLS == EPI_CLS_GRU || (CLS == EPI_CLS_ANY && kind == PPMS_EPI_GRU)) {
        const float* zp = e.aux_f32 + pix * e.aux_f32_ld + cl;
        a.z0 = gld<f32x4>(zp), a.z1 = gld<f32x4>(zp + 4);
    }
}

template <int CLS>
__device__ __forceinline__ void row8_finish(const ppms_epilogue& e, const float* vin, int64_t pix, int cl, int hw, const row8_aux& a) {
    constexpr bool LD = CLS != EPI_CLS_PLAIN;
    const int nv = e.n_valid - cl;
    if (nv <= 0) return;
    if (nv < 8) {                                  // ragged tail of the valid couts: the 4-wide predicated form, twice
#pragma unroll 1
        for (int s = 0; s < 2; ++s) {
            float v4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v4[j] = s ? vin[4 + j] : vin[j];
            epilogue_group<LD>(e, v4, pix, cl + 4 * s, hw, false);
        }
        return;
    }
    float v[8], y[8], ax[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        v[j] = vin[j];
        ax[j] = 0.0f;
    }
    if ((CLS == EPI_CLS_PRE || CLS == EPI_CLS_GRU || CLS == EPI_CLS_ANY || CLS == EPI_CLS_AUXPRE) && e.pre_f32 != nullptr) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] += a.p0[j];
            v[4 + j] += a.p1[j];
        }
    }
    const int kind = (CLS == EPI_CLS_PLAIN || CLS == EPI_CLS_PRE) ? PPMS_EPI_STORE : (CLS == EPI_CLS_GRU ? PPMS_EPI_GRU : e.kind);
    if (kind == PPMS_EPI_RESID || kind == PPMS_EPI_RH || kind == PPMS_EPI_GRU) {
#pragma unroll
        for (int j = 0; j < 8; ++j) ax[j] = join_bf16(a.h8[j], a.l8[j]);
    }
    if (kind == PPMS_EPI_RESID) {
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = ax[j] + v[j];
        apply_act_n<8>(y, e.act, e.scale);
    } else if (kind == PPMS_EPI_RH) {
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = sigmoid_fast(v[j]) * ax[j];
    } else if (kind == PPMS_EPI_GRU) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float z = j < 4 ? a.z0[j & 3] : a.z1[j & 3];
            y[j] = (1.0f - z) * ax[j] + z * tanh_fast(v[j]);
        }
    } else if (kind == PPMS_EPI_ADDF32) {
        float* op = e.out_f32 + pix * e.out_f32_ld + cl;
        f32x4 o0 = gld<f32x4>(op), o1 = gld<f32x4>(op + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o0[j] += v[j];
            o1[j] += v[4 + j];
        }
        gst<f32x4>(op, o0);
        gst<f32x4>(op + 4, o1);
        return;
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = v[j];
        apply_act_n<8>(y, e.act, e.scale);
    }
    if (e.out_sp.hi != nullptr) {
        bf16x8 h8, l8;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            bf16_t hh, ll;
            split_bf16(y[j], hh, ll);
            h8[j] = hh;
            l8[j] = ll;
        }
        gst<bf16x8>((bf16_t*)e.out_sp.hi + pix * e.out_sp.ld + cl, h8);
        gst<bf16x8>((bf16_t*)e.out_sp.lo + pix * e.out_sp.ld + cl, l8);
    }
    if (e.out_f32 != nullptr) {
        float* op = e.out_f32 + pix * e.out_f32_ld + cl;
        const f32x4 o0 = {y[0], y[1], y[2], y[3]}, o1 = {y[4], y[5], y[6], y[7]};
        gst<f32x4>(op, o0);
        gst<f32x4>(op + 4, o1);
    }
}

// one row, operands fetched on the spot (callers without a row loop to group: the slice-reduce kernel)
template <bool LD = true>
__device__ __forceinline__ void epilogue_row8(const ppms_epilogue& e, const float* vin, int64_t pix, int cl, int hw) {
    constexpr int CLS = LD ? EPI_CLS_ANY : EPI_CLS_PLAIN;
    row8_aux a;
    row8_fetch<CLS>(e, pix, cl, a);
    row8_finish<CLS>(e, vin, pix, cl, hw, a);
}

// accumulator block (couts mb*32 .. +31 of the wave's 64) of 32 pixels -> the wave's staging patch [pixel][cout]
__device__ __forceinline__ void stage_write32(float* stg, int r, int h, int mb, int gq, const f32x4& a4) {
    *(f32x4*)(stg + r * STG_LD + mb * 32 + 8 * gq + 4 * h) = a4;
}

// lane -> (pixel row it*8 + lane/8, couts 8*(lane%8) .. +7) of the staged block
__device__ __forceinline__ void stage_read8(const float* stg, int prow, int q, float* v) {
    const f32x4 a = *(const f32x4*)(stg + prow * STG_LD + q * 8), b = *(const f32x4*)(stg + prow * STG_LD + q * 8 + 4);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        v[j] = a[j];
        v[4 + j] = b[j];
    }
}

// host-side check shared by the kernels that use the coalesced form
inline const char* epilogue_row8_check(const ppms_epilogue& e) {
    auto al16 = [](const void* p) { return ((uintptr_t)p & 15) == 0; };
    if (e.out_sp.hi != nullptr && !(al16(e.out_sp.hi) && al16(e.out_sp.lo) && e.out_sp.ld % 8 == 0)) return "out_sp must be 16-byte aligned with ld % 8 == 0";
    if ((e.kind == PPMS_EPI_RESID || e.kind == PPMS_EPI_RH || e.kind == PPMS_EPI_GRU) &&
        !(al16(e.aux_sp.hi) && al16(e.aux_sp.lo) && e.aux_sp.ld % 8 == 0))
        return "aux_sp must be 16-byte aligned with ld % 8 == 0";
    if (e.out_f32 != nullptr && !(al16(e.out_f32) && e.out_f32_ld % 4 == 0)) return "out_f32 must be 16-byte aligned with ld % 4 == 0";
    if (e.kind == PPMS_EPI_GRU && !(al16(e.aux_f32) && e.aux_f32_ld % 4 == 0)) return "aux_f32 must be 16-byte aligned with ld % 4 == 0";
    if (e.pre_f32 != nullptr && !(al16(e.pre_f32) && e.pre_f32_ld % 4 == 0)) return "pre_f32 must be 16-byte aligned with ld % 4 == 0";
    if (e.out_vt != nullptr && (e.kind != PPMS_EPI_STORE || e.pre_f32 != nullptr)) return "out_vt needs a STORE epilogue without pre_f32";
    return nullptr;
}
