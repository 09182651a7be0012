// Fused conv epilogue shared by the implicit-GEMM kernels: bias is already added by the caller.
#pragma once
#include "common.h"

__device__ __forceinline__ void epilogue_group(const ppms_epilogue& e, const float* vin, int64_t pix, int cl, int hw, bool with_vt = true) {
    // v[0..3]: acc + bias for couts cl..cl+3 (local to this half) at pixel pix.  Everything is predicated (no early
    // exits, no runtime trip counts) so that the caller's loops unroll fully and the accumulators stay in registers.
    const int nv = e.n_valid - cl;            // how many of the 4 are real
    const bool all4 = nv >= 4;
    float y[4];
    float ax[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    float v[4] = {vin[0], vin[1], vin[2], vin[3]};
    if (e.pre_f32 != nullptr && nv > 0) {     // iteration-invariant part of the pre-activation (n_valid is a multiple of 4 here)
        const f32x4 p4 = *(const f32x4*)(e.pre_f32 + pix * e.pre_f32_ld + cl);
#pragma unroll
        for (int j = 0; j < 4; ++j) v[j] += p4[j];
    }
    const int kind = e.kind;
    if (nv > 0 && (kind == PPMS_EPI_RESID || kind == PPMS_EPI_RH || kind == PPMS_EPI_GRU)) {
        const bf16_t* ah = (const bf16_t*)e.aux_sp.hi + pix * e.aux_sp.ld + cl;
        const bf16_t* al = (const bf16_t*)e.aux_sp.lo + pix * e.aux_sp.ld + cl;
        if (all4) {
            const bf16x4 h4 = *(const bf16x4*)ah, l4 = *(const bf16x4*)al;
#pragma unroll
            for (int j = 0; j < 4; ++j) ax[j] = join_bf16(h4[j], l4[j]);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < nv) ax[j] = join_bf16(ah[j], al[j]);
        }
    }
    if (kind == PPMS_EPI_RESID) {
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = apply_act(ax[j] + v[j], e.act) * e.scale;
    } else if (kind == PPMS_EPI_RH) {
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = sigmoid_fast(v[j]) * ax[j];
    } else if (kind == PPMS_EPI_GRU) {
        const float* zp = e.aux_f32 + pix * e.aux_f32_ld + cl;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float z = (j < nv) ? zp[j] : 0.0f;
            y[j] = (1.0f - z) * ax[j] + z * tanh_fast(v[j]);
        }
    } else if (kind == PPMS_EPI_ADDF32) {
        float* op = e.out_f32 + pix * e.out_f32_ld + cl;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            y[j] = 0.0f;
            if (j < nv) op[j] += v[j];
        }
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j) y[j] = apply_act(v[j], e.act) * e.scale;
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
                *(bf16x4*)oh = h4;
                *(bf16x4*)ol = l4;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j < nv) {
                        oh[j] = h4[j];
                        ol[j] = l4[j];
                    }
            }
        }
        if (e.out_f32 != nullptr) {
            float* op = e.out_f32 + pix * e.out_f32_ld + cl;
            if (all4) {
                f32x4 o = {y[0], y[1], y[2], y[3]};
                *(f32x4*)op = o;
            } else {
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    if (j < nv) op[j] = y[j];
            }
        }
        if (with_vt && e.out_vt != nullptr) {
            const int64_t frame = pix / hw;
            const int64_t rem = pix - frame * hw;
            bf16_t* vp = (bf16_t*)e.out_vt + (frame * e.n_valid + cl) * hw + rem;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < nv) vp[(int64_t)j * hw] = (bf16_t)y[j];
        }
    }
}


// ------------------------------------------------------------------------------------------------------------------
// Coalesced form (conv_gemm2 / conv_gemm3).  In the MFMA accumulator layout a lane owns ONE pixel and 4 couts per
// register group, so every epilogue load / store of a wave touches 32 different cache lines with 8-16 useful bytes
// each; with the residual / gate / hoisted-share operands that is several thousand line requests per tile and showed
// up as 15-30 % of the short GRU convs.  The kernels therefore transpose 32-pixel x 64-cout accumulator blocks through
// a wave-private LDS patch (stage_write32 / stage_read8) and run the epilogue on 8 consecutive couts of one pixel per
// lane: 8 lanes cover 64 couts of a pixel, i.e. whole 128-B (bf16 planes) / 256-B (fp32) runs per pixel.
constexpr int STG_LD = 68;                        // dwords per staged pixel row (64 couts + 4 pad: conflict-free b128 phases)
constexpr int STG_WAVE = 32 * STG_LD * 4;         // bytes of one wave's staging patch (32 pixels)

// the transposed V operand of the memory attention (bf16 [frame][n_valid][H*W]) is pixel-major: it is written from
// the accumulator layout (lanes = adjacent pixels), before the transposition.  STORE epilogues only.
__device__ __forceinline__ void epilogue_vt4(const ppms_epilogue& e, const float* v, int64_t pix, int cl, int hw) {
    const int nv = e.n_valid - cl;
    const int64_t frame = pix / hw;
    const int64_t rem = pix - frame * hw;
    bf16_t* vp = (bf16_t*)e.out_vt + (frame * e.n_valid + cl) * hw + rem;
#pragma unroll
    for (int j = 0; j < 4; ++j)
        if (j < nv) vp[(int64_t)j * hw] = (bf16_t)(apply_act(v[j], e.act) * e.scale);
}

// v[0..7]: acc + bias for couts cl..cl+7 (local to this half, cl % 8 == 0) at pixel pix
__device__ __forceinline__ void epilogue_row8(const ppms_epilogue& e, const float* vin, int64_t pix, int cl, int hw) {
    const int nv = e.n_valid - cl;
    if (nv <= 0) return;
    if (nv < 8) {                                  // ragged tail of the valid couts: the 4-wide predicated form, twice
#pragma unroll 1
        for (int s = 0; s < 2; ++s) {
            float v4[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v4[j] = s ? vin[4 + j] : vin[j];
            epilogue_group(e, v4, pix, cl + 4 * s, hw, false);
        }
        return;
    }
    float v[8], y[8], ax[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        v[j] = vin[j];
        ax[j] = 0.0f;
    }
    if (e.pre_f32 != nullptr) {
        const float* pp = e.pre_f32 + pix * e.pre_f32_ld + cl;
        const f32x4 p0 = *(const f32x4*)pp, p1 = *(const f32x4*)(pp + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] += p0[j];
            v[4 + j] += p1[j];
        }
    }
    const int kind = e.kind;
    if (kind == PPMS_EPI_RESID || kind == PPMS_EPI_RH || kind == PPMS_EPI_GRU) {
        const bf16x8 h8 = *(const bf16x8*)((const bf16_t*)e.aux_sp.hi + pix * e.aux_sp.ld + cl);
        const bf16x8 l8 = *(const bf16x8*)((const bf16_t*)e.aux_sp.lo + pix * e.aux_sp.ld + cl);
#pragma unroll
        for (int j = 0; j < 8; ++j) ax[j] = join_bf16(h8[j], l8[j]);
    }
    if (kind == PPMS_EPI_RESID) {
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = apply_act(ax[j] + v[j], e.act) * e.scale;
    } else if (kind == PPMS_EPI_RH) {
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = sigmoid_fast(v[j]) * ax[j];
    } else if (kind == PPMS_EPI_GRU) {
        const float* zp = e.aux_f32 + pix * e.aux_f32_ld + cl;
        const f32x4 z0 = *(const f32x4*)zp, z1 = *(const f32x4*)(zp + 4);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float z = j < 4 ? z0[j & 3] : z1[j & 3];
            y[j] = (1.0f - z) * ax[j] + z * tanh_fast(v[j]);
        }
    } else if (kind == PPMS_EPI_ADDF32) {
        float* op = e.out_f32 + pix * e.out_f32_ld + cl;
        f32x4 o0 = *(f32x4*)op, o1 = *(f32x4*)(op + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            o0[j] += v[j];
            o1[j] += v[4 + j];
        }
        *(f32x4*)op = o0;
        *(f32x4*)(op + 4) = o1;
        return;
    } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = apply_act(v[j], e.act) * e.scale;
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
        *(bf16x8*)((bf16_t*)e.out_sp.hi + pix * e.out_sp.ld + cl) = h8;
        *(bf16x8*)((bf16_t*)e.out_sp.lo + pix * e.out_sp.ld + cl) = l8;
    }
    if (e.out_f32 != nullptr) {
        float* op = e.out_f32 + pix * e.out_f32_ld + cl;
        const f32x4 o0 = {y[0], y[1], y[2], y[3]}, o1 = {y[4], y[5], y[6], y[7]};
        *(f32x4*)op = o0;
        *(f32x4*)(op + 4) = o1;
    }
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
