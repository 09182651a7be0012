// Implicit-GEMM convolution on bf16 MFMA with fp32-accurate split operands (x = hi + lo):
//     D[cout][pixel] = sum_k ( Whi*Xhi + Whi*Xlo + Wlo*Xhi ),  k = (tap, input channel), fp32 accumulate.
// One kernel serves every conv of the update block (ppmtereo_update.py:292-310 GRU, :473-480 motion encoder,
// :673-674 flow head, :889-893 uncertainty, :910-914 mask, :646 to_v, :129 to_qk); the epilogue fuses bias,
// activation, residuals and the GRU gate arithmetic and writes the next op's operand format directly.
//
// Tiling (gfx950): workgroup = 4 waves = 64 couts x 256 pixels; each wave 64 x 64 = 2x2 MFMA 32x32x16 tiles;
// K step 32 (one tap, 32 input channels), two LDS stages (80 KiB -> 2 workgroups / CU), register-staged global
// loads issued before the MFMA block and written to LDS after it.  LDS rows are 64 B with the 16-B chunk index
// XOR-swizzled by (row>>2)&3, which makes every ds_read_b128 fragment read conflict free.  Workgroups that share a
// pixel tile (different cout blocks) are placed on one XCD (same id mod 8) so the tile is re-read from that L2.
#include "common.h"
#include "conv_epilogue.h"

namespace {

constexpr int BM = 64, BN = 256, BK = 32;
constexpr int A_PLANE = BM * BK * 2;   // 4 KiB
constexpr int B_PLANE = BN * BK * 2;   // 16 KiB
constexpr int STAGE = 2 * A_PLANE + 2 * B_PLANE;   // 40 KiB

__device__ __forceinline__ int swz(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

// The descriptor lives in device memory (built once per scale by the host, every pointer in it is fixed for the whole
// scale): fields are fetched by scalar loads where they are used instead of occupying ~80 SGPRs as a by-value argument.
__global__ __launch_bounds__(256, 2) void conv_gemm_kernel(const ppms_conv* __restrict__ pd, const int ntiles, const int mblocks,
                                                           const int nk, const int cpt /*k-steps per tap*/,
                                                           const int n0 /*k-steps of seg 0*/) {
    const ppms_conv& p = *pd;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    // XCD-aware id -> (pixel tile, cout block): ids equal mod 8 share an XCD (L2); keep a pixel tile's cout blocks there
    const int id = blockIdx.x;
    const int xcd = id & 7, jj = id >> 3;
    const int mblk = jj % mblocks;
    const int ntile = (jj / mblocks) * 8 + xcd;
    if (ntile >= ntiles) return;

    const int64_t P = (int64_t)p.T * p.H * p.W;
    const int HW = p.H * p.W;
    // ---- B-operand rows owned by this thread: rows (tid>>2) + 64*i, 16-B chunk tid&3 -------------------
    const int cB = tid & 3;
    int rx[4], ry[4], rt[4];
    int64_t rp[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int64_t pix = (int64_t)ntile * BN + (tid >> 2) + 64 * i;
        rp[i] = pix;
        if (pix < P) {
            const int t = (int)(pix / HW);
            const int rem = (int)(pix - (int64_t)t * HW);
            rt[i] = t;
            ry[i] = rem / p.W;
            rx[i] = rem - ry[i] * p.W;
        } else {
            rt[i] = -(1 << 20);   // never valid
            ry[i] = rx[i] = 0;
        }
    }
    const char* wbase = (const char*)p.w + ((int64_t)mblk * nk) * (2 * A_PLANE) + tid * 16;

    u32x4 ra[2], rb[8];
    auto load_regs = [&](int ks, int tap, int ch) {
        const char* wp = wbase + (int64_t)ks * (2 * A_PLANE);
        ra[0] = gload16(wp);
        ra[1] = gload16(wp + A_PLANE);
        const int s = (ch >= n0) ? 1 : 0;
        const int c0 = (ch - (s ? n0 : 0)) * BK + cB * 8;
        const bf16_t* sh = (const bf16_t*)p.seg[s].hi;
        const bf16_t* sl = (const bf16_t*)p.seg[s].lo;
        const int ld = p.seg[s].ld;
        // tap -> (dt, dy, dx)
        const int kx = tap % p.kw;
        const int r2 = tap / p.kw;
        const int ky = r2 % p.kh;
        const int kz = r2 / p.kh;
        const int dx = kx - (p.kw >> 1), dy = ky - (p.kh >> 1), dt = kz - (p.kt >> 1);
        const int64_t shift = (int64_t)dt * HW + dy * p.W + dx;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const bool ok = (unsigned)(rx[i] + dx) < (unsigned)p.W && (unsigned)(ry[i] + dy) < (unsigned)p.H &&
                            (unsigned)(rt[i] + dt) < (unsigned)p.T;
            if (ok) {
                const int64_t off = (rp[i] + shift) * ld + c0;
                rb[2 * i] = gload16(sh + off);
                rb[2 * i + 1] = gload16(sl + off);
            } else {
                rb[2 * i] = (u32x4){0, 0, 0, 0};
                rb[2 * i + 1] = (u32x4){0, 0, 0, 0};
            }
        }
    };
    auto store_lds = [&](int stage) {
        char* s = smem + stage * STAGE;
        *(u32x4*)(s + tid * 16) = ra[0];
        *(u32x4*)(s + A_PLANE + tid * 16) = ra[1];
        char* bh = s + 2 * A_PLANE;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int off = swz((tid >> 2) + 64 * i, cB);
            *(u32x4*)(bh + off) = rb[2 * i];
            *(u32x4*)(bh + B_PLANE + off) = rb[2 * i + 1];
        }
    };

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (f32x16){0};

    const int r = lane & 31, h = lane >> 5;
    int tap = 0, ch = 0;          // of the k-step being LOADED
    load_regs(0, 0, 0);
    store_lds(0);
    __syncthreads();
    for (int ks = 0; ks < nk; ++ks) {
        const bool more = ks + 1 < nk;
        if (more) {
            if (++ch == cpt) {
                ch = 0;
                ++tap;
            }
            load_regs(ks + 1, tap, ch);
        }
        const char* s = smem + (ks & 1) * STAGE;
        const char* bh = s + 2 * A_PLANE;
#pragma unroll
        for (int k16 = 0; k16 < 2; ++k16) {
            bf16x8 ah[2], al[2], bhv[2], blv[2];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const int off = swz(mb * 32 + r, 2 * k16 + h);
                ah[mb] = *(const bf16x8*)(s + off);
                al[mb] = *(const bf16x8*)(s + A_PLANE + off);
            }
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const int off = swz(wave * 64 + nb * 32 + r, 2 * k16 + h);
                bhv[nb] = *(const bf16x8*)(bh + off);
                blv[nb] = *(const bf16x8*)(bh + B_PLANE + off);
            }
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mb], bhv[nb], acc[mb][nb], 0, 0, 0);
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mb], blv[nb], acc[mb][nb], 0, 0, 0);
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mb], bhv[nb], acc[mb][nb], 0, 0, 0);
                }
        }
        if (more) store_lds((ks + 1) & 1);
        __syncthreads();
    }

    // ---- epilogue: D col = lane&31 (pixel), row = (reg&3) + 8*(reg>>2) + 4*(lane>>5) (cout) ------------
    const int cblock = mblk * BM;
    const int half = (cblock >= p.m_split) ? 1 : 0;
    const ppms_epilogue& e = p.epi[half];
    const int cbase = cblock - (half ? p.m_split : 0);
    // one copy of the epilogue code, 16 trips; the accumulator group of trip `it` is picked with static indices
    for (int it = 0; it < 16; ++it) {
        const int nb = it >> 3, mb = (it >> 2) & 1, g = it & 3;
        float a4[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
        for (int s_nb = 0; s_nb < 2; ++s_nb)
#pragma unroll
            for (int s_mb = 0; s_mb < 2; ++s_mb)
#pragma unroll
                for (int s_g = 0; s_g < 4; ++s_g)
                    if (it == s_nb * 8 + s_mb * 4 + s_g) {
#pragma unroll
                        for (int j = 0; j < 4; ++j) a4[j] = acc[s_mb][s_nb][4 * s_g + j];
                    }
        const int64_t pix = (int64_t)ntile * BN + wave * 64 + nb * 32 + r;
        if (pix < P) {
            const int c4 = mb * 32 + 8 * g + 4 * h;
            const f32x4 b4 = *(const f32x4*)(p.bias + cblock + c4);
            float v[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = a4[j] + b4[j];
            epilogue_group(e, v, pix, cbase + c4, HW);
        }
    }
}

}  // namespace

extern "C" int ppms_conv_gemm(const ppms_conv* d, const ppms_conv* dev_desc, void* stream) {
    PPMS_REQUIRE(d != nullptr && dev_desc != nullptr, "conv_gemm: null descriptor (host copy and device copy are both required)");
    PPMS_REQUIRE(d->nseg == 1 || d->nseg == 2, "conv_gemm: nseg=%d", d->nseg);
    PPMS_REQUIRE(d->T > 0 && d->H > 0 && d->W > 0, "conv_gemm: bad volume %dx%dx%d", d->T, d->H, d->W);
    PPMS_REQUIRE((d->kt & 1) && (d->kh & 1) && (d->kw & 1), "conv_gemm: kernel extents must be odd");
    PPMS_REQUIRE(d->M > 0 && d->M % BM == 0, "conv_gemm: M=%d not a multiple of %d", d->M, BM);
    PPMS_REQUIRE(d->m_split % BM == 0, "conv_gemm: m_split=%d not a multiple of %d", d->m_split, BM);
    PPMS_REQUIRE(d->w != nullptr && d->bias != nullptr, "conv_gemm: weights/bias missing");
    int cpt = 0;
    for (int s = 0; s < d->nseg; ++s) {
        PPMS_REQUIRE(d->seg[s].hi && d->seg[s].lo && d->seg[s].c > 0 && d->seg[s].c % BK == 0 && d->seg[s].ld % 8 == 0,
                     "conv_gemm: segment %d needs hi/lo planes, c %% 32 == 0 and ld %% 8 == 0 (c=%d ld=%d)", s, d->seg[s].c, d->seg[s].ld);
        PPMS_REQUIRE(((uintptr_t)d->seg[s].hi & 15) == 0 && ((uintptr_t)d->seg[s].lo & 15) == 0, "conv_gemm: segment %d not 16-B aligned", s);
        cpt += d->seg[s].c / BK;
    }
    for (int hlf = 0; hlf < 2; ++hlf) {
        const ppms_epilogue& e = d->epi[hlf];
        if (hlf == 1 && d->m_split >= d->M) break;
        PPMS_REQUIRE(e.n_valid > 0, "conv_gemm: epilogue %d has n_valid=%d", hlf, e.n_valid);
        PPMS_REQUIRE(e.pre_f32 == nullptr || (e.n_valid % 4 == 0 && e.pre_f32_ld % 4 == 0), "conv_gemm: pre_f32 needs n_valid and pre_f32_ld to be multiples of 4");
        if (e.out_sp.hi) PPMS_REQUIRE(e.out_sp.lo && e.out_sp.ld % 4 == 0 && ((uintptr_t)e.out_sp.hi & 7) == 0 && ((uintptr_t)e.out_sp.lo & 7) == 0,
                                      "conv_gemm: epilogue %d SP output misaligned", hlf);
        if (e.out_f32) PPMS_REQUIRE(e.out_f32_ld % 4 == 0 || e.kind == PPMS_EPI_ADDF32, "conv_gemm: epilogue %d f32 ld", hlf);
        if (e.kind == PPMS_EPI_RESID || e.kind == PPMS_EPI_RH || e.kind == PPMS_EPI_GRU)
            PPMS_REQUIRE(e.aux_sp.hi && e.aux_sp.lo && e.aux_sp.ld % 4 == 0, "conv_gemm: epilogue %d needs aux_sp", hlf);
        if (e.kind == PPMS_EPI_GRU) PPMS_REQUIRE(e.aux_f32 != nullptr, "conv_gemm: GRU epilogue needs z");
        if (e.kind == PPMS_EPI_ADDF32) PPMS_REQUIRE(e.out_f32 != nullptr, "conv_gemm: ADDF32 epilogue needs out_f32");
    }
    const int64_t P = (int64_t)d->T * d->H * d->W;
    const int ntiles = ceil_div(P, BN);
    const int mblocks = d->M / BM;
    const int taps = d->kt * d->kh * d->kw;
    const int nk = taps * cpt;
    const int n0 = d->seg[0].c / BK;
    const int grid = ((ntiles + 7) / 8) * 8 * mblocks;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv_gemm_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE);
        attr_set = true;
    }
    hipLaunchKernelGGL(conv_gemm_kernel, dim3(grid), dim3(256), 2 * STAGE, (hipStream_t)stream, dev_desc, ntiles, mblocks, nk, cpt, n0);
    return ppms_check_launch("conv_gemm");
}
