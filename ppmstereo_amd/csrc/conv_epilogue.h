// Fused conv epilogue shared by the implicit-GEMM kernels: bias is already added by the caller.
#pragma once
#include "common.h"

__device__ __forceinline__ void epilogue_group(const ppms_epilogue& e, const float* v, int64_t pix, int cl, int hw) {
    // v[0..3]: acc + bias for couts cl..cl+3 (local to this half) at pixel pix.  Everything is predicated (no early
    // exits, no runtime trip counts) so that the caller's loops unroll fully and the accumulators stay in registers.
    const int nv = e.n_valid - cl;            // how many of the 4 are real
    const bool all4 = nv >= 4;
    float y[4];
    float ax[4] = {0.0f, 0.0f, 0.0f, 0.0f};
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
        for (int j = 0; j < 4; ++j) y[j] = sigmoid_f(v[j]) * ax[j];
    } else if (kind == PPMS_EPI_GRU) {
        const float* zp = e.aux_f32 + pix * e.aux_f32_ld + cl;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float z = (j < nv) ? zp[j] : 0.0f;
            y[j] = (1.0f - z) * ax[j] + z * tanhf(v[j]);
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
        if (e.out_vt != nullptr) {
            const int64_t frame = pix / hw;
            const int64_t rem = pix - frame * hw;
            bf16_t* vp = (bf16_t*)e.out_vt + (frame * e.n_valid + cl) * hw + rem;
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < nv) vp[(int64_t)j * hw] = (bf16_t)y[j];
        }
    }
}

