// Implicit-GEMM convolution, second generation: data-reuse tiling for gfx950.
//
// Same math and descriptor as conv_gemm.hip (bf16x3 split MFMA, fp32 accumulate, fused epilogues) but the operand
// traffic is cut to what the 32 MiB of L2 can deliver:
//   * one workgroup owns ALL output channels of its pixel tile (BM = 64*WM couts, WM in 1..4), so the activation
//     tile is fetched once instead of once per 64-cout block;
//   * the pixel tile is an R x C patch (R*C = 128) and the activations of one (dt, dy, 32-channel chunk) are staged
//     in LDS as a halo'd window of R x (C + kw - 1) pixel rows; the kw taps along x then sweep that window from LDS
//     (15 taps of the GRU's (1,1,15) conv re-read 1 window instead of 15 tiles);
//   * per k-step only the weight tile (BM x 32, contiguous, pre-swizzled) streams in: it is the same for every
//     workgroup, so it comes out of L2.
// Bytes per k-step drop from 40 KiB per 1.05 MFLOP (v1) to <= (8*WM + 16/kw..16) KiB per 0.26*WM MFLOP.
//
// Waves: 2*WM waves, wave (wm, wn) computes couts [64 wm, 64 wm + 64) x pixels [64 wn, 64 wn + 64) = 2x2 MFMA
// 32x32x16 tiles.  LDS: weight stages 2 x WM x 8 KiB, window stages 2 x 2 planes x rows x 64 B (XOR-swizzled rows).
// Packed weights (ppmstereo_amd/packing.py pack_conv2): [k-step][M/64][hi,lo][64][32] with
// k-step = ((kz*kh + ky) * nchunk + chunk) * kw + kx.
#include "common.h"
#include "conv_epilogue.h"
#include <type_traits>

namespace {

constexpr int BK = 32;
#ifndef CONV2_ABL_A
#define CONV2_ABL_A 0        // ablation builds (-DCONV2_ABL_A=1): every k-step loads the FIRST step's weights (cache hits): wrong results, timing only
#endif
constexpr int A_BLK = 64 * BK * 2 * 2;       // one 64-cout block, hi + lo planes: 8 KiB

struct Geo2 {
    int C, R, logC;          // patch: R rows x C cols, R*C = 128
    int tiles_x, tiles_y;    // per frame
    int WR;                  // window row length in pixels = C + kw - 1
    int Wr;                  // window rows = R * WR
    int nchunk, n0;          // 32-channel chunks per tap (all segments), chunks of segment 0
    int nk;                  // k-steps = kt*kh*nchunk*kw
    int mgroups;             // M / (64*WM)
    int bstages;             // 2: window double-buffered; 1: single window + one extra barrier per window switch
    int ysweep;              // 1: (1, kh, 1) conv swept along y: column-major patch / window (fast axis y, logF = log2 R), taps step
                             //    the window row; weights packed with kh / kw swapped.  0: taps along x (fast axis x, logF = log2 C)
    int ksw, logF;           // taps swept inside one window (kw; kh when ysweep; kh*kw for a 2-D window); log2 of the fast-axis length
    int swn, jump;           // the LDS row advances by 1 per tap and by `jump` more after every `swn` taps (2-D window: kw, WR - kw)
    int hs2;                 // 2-D window: halo rows above the patch (kh / 2); 0 otherwise
    int nslice;              // grid-level K slices (gridDim.y): slice s takes the row-steps s*KG + kg, + KG*nslice, ...
    float* part;             // nslice > 1: fp32 partial sums [slice][pixel][M] (no bias), finished by conv_slice_reduce_kernel
    int64_t P;               // pixels = T*H*W
    unsigned wr_magic;       // ceil(2^20 / WR): wrow / WR == (wrow * wr_magic) >> 20 for every window row (checked on the host); 0: divide
    // x / d == __umulhi(x, floor(2^32 / d) + 1) for x, d < 2^16 (host-checked ranges): the set-up of a small-map launch was mostly runtime
    // integer divisions (tile index, row-step -> tap / chunk), ~40 instructions each
    unsigned m_mgroups, m_tiles_x, m_tiles_y, m_nchunk, m_kho, m_kstride;
    int kho;                 // kernel rows NOT swept inside a window (p.kh, or 1 for the y-swept / 2-D forms)
#ifdef PPMS_CONV2_TIMING
    long long* dbg;          // debug build only: [workgroup (x + gridDim.x * y)][8] wall-clock stamps (100 MHz) of wave 0
#endif
};

#ifdef PPMS_CONV2_TIMING
static long long* g_conv2_dbg = nullptr;
#define CONV2_STAMP(K)                                                                                                    \
    if (g.dbg != nullptr && __builtin_amdgcn_readfirstlane(threadIdx.x) == 0) /* all of wave 0: a uniform branch */       \
        g.dbg[((int64_t)blockIdx.x + (int64_t)gridDim.x * blockIdx.y) * 8 + (K)] = wall_clock64();
#else
#define CONV2_STAMP(K)
#endif

// x / d with m = floor(2^32 / d) + 1 (exact while x * d < 2^32; the host checks the ranges), m == 0 standing for d == 1
__device__ __forceinline__ int fdiv(int x, unsigned m) { return m ? (int)__umulhi((unsigned)x, m) : x; }

__device__ __forceinline__ int swz2(int row, int chunk) { return row * 64 + ((chunk ^ ((row >> 2) & 3)) << 4); }

// KG > 1: intra-workgroup split-K for small maps (too few tiles to fill 256 CUs): KG wave groups, each with its own
// LDS stages, take the (dt, dy, chunk) row-steps kg, kg+KG, ... of the SAME output tile, so KG k-step pipelines run
// concurrently in one workgroup; partial accumulators are summed through LDS in fixed order (deterministic) and wave
// group 0 runs the epilogue.
template <int WM, int KG>
__global__ __launch_bounds__(128 * WM * KG) void conv2_kernel(const ppms_conv pv, const Geo2 g) {
    constexpr int NT = 128 * WM;              // threads of one K-group
    constexpr int MAXSLOT = (WM == 4) ? 2 : (WM == 3) ? 3 : (WM == 2) ? 4 : 8;      // window 16-B chunks per thread and plane
    extern __shared__ __attribute__((aligned(16))) char smem[];
    CONV2_STAMP(0)
    // the descriptor travels BY VALUE in the kernel arguments (copied from the host descriptor at launch): one dependent memory round trip
    // less at the head of every launch than through the device copy, and kernarg loads are known not to alias the kernel's stores
    const ppms_conv& p = pv;
    const int kg = (KG > 1) ? (int)threadIdx.x / NT : 0;
    const int tid = (KG > 1) ? (int)threadIdx.x % NT : (int)threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int r = lane & 31, h = lane >> 5;
    const int tile0 = fdiv((int)blockIdx.x, g.m_mgroups);
    const int mgrp = (int)blockIdx.x - tile0 * g.mgroups;
    const int tile1 = fdiv(tile0, g.m_tiles_x);
    const int tx = tile0 - tile1 * g.tiles_x;
    const int tf = fdiv(tile1, g.m_tiles_y);               // frame
    const int ty = tile1 - tf * g.tiles_y;
    const int x0 = tx * g.C, y0 = ty * g.R;
    const int H = p.H, W = p.W, T = p.T;
    const int HW = H * W;
    const int hy = p.kh >> 1, ht = p.kt >> 1;
    const int hsw = g.swn >> 1;               // halo of the window along the fast (swept) axis
    const int F = 1 << g.logF;                // patch extent along the fast (swept) axis

    const int bplane = g.Wr * 64;
    char* sA = smem + kg * (2 * WM * A_BLK + g.bstages * 2 * bplane);      // this K-group's stages: 2 x WM x 8 KiB weights,
    char* sB = sA + 2 * WM * A_BLK;                                        // then bstages x 2 planes x Wr x 64 B window

    // ---- window slots of this thread (fixed for the whole kernel) ------------------------------------------
    int sl_off[MAXSLOT];          // pixel offset (t*H + y)*W + x of the slot at (dt, dy) = 0, or -1: x outside the image
    int sl_y[MAXSLOT];
    int sl_lds[MAXSLOT];          // swizzled LDS byte offset inside a plane, -1: slot unused
    const int nslot_total = g.Wr * 4;
#pragma unroll
    for (int i = 0; i < MAXSLOT; ++i) {
        const int j = tid + i * NT;
        sl_lds[i] = -1;
        sl_off[i] = -1;
        sl_y[i] = 0;
        if (j < nslot_total) {
            const int wrow = j >> 2, c = j & 3;
            // slow / fast (swept, halo'd) window coordinate: a multiply-shift instead of a runtime division per slot (8 slots: ~1 us of set-up)
            const int ws = g.wr_magic ? (int)(((unsigned)wrow * g.wr_magic) >> 20) : wrow / g.WR, wf = wrow - ws * g.WR;
            const int x = g.ysweep ? x0 + ws : x0 + wf - hsw, y = g.ysweep ? y0 + wf - hsw : y0 + ws - g.hs2;
            sl_lds[i] = swz2(wrow, c);
            sl_y[i] = y;
            if ((unsigned)x < (unsigned)W) sl_off[i] = (tf * H + y) * W + x;
        }
    }
    const int cB = tid & 3;       // 16-B chunk inside the 64-B channel group (j & 3 == tid & 3 because NT % 4 == 0)

    const char* wbase = (const char*)p.w + (int64_t)mgrp * WM * A_BLK + tid * 16;
    const int64_t wstep = (int64_t)(p.M / 64) * A_BLK;

    u32x4 ra[4], rbh[MAXSLOT], rbl[MAXSLOT];
    auto load_a = [&](int ks) {
        const char* wp = wbase + (int64_t)(CONV2_ABL_A ? 0 : ks) * wstep;
#pragma unroll
        for (int i = 0; i < 4; ++i) ra[i] = gload16(wp + i * NT * 16);
    };
    auto store_a = [&](int stage) {
        char* s = sA + stage * WM * A_BLK + tid * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) *(u32x4*)(s + i * NT * 16) = ra[i];
    };
    auto load_b = [&](int trow, int chunk) {
        const int kho = g.kho;                                     // kernel rows NOT swept inside a window
        const int kz = fdiv(trow, g.m_kho), ky = trow - kz * kho;
        const int dy = kho == 1 ? 0 : ky - hy, dt = kz - ht;
        const int s = (chunk >= g.n0) ? 1 : 0;
        const int c0 = (chunk - (s ? g.n0 : 0)) * BK + cB * 8;
        const bf16_t* sh = (const bf16_t*)p.seg[s].hi;
        const bf16_t* sl = (const bf16_t*)p.seg[s].lo;
        const int ld = p.seg[s].ld;
        const bool tok = (unsigned)(tf + dt + p.t_halo) < (unsigned)(T + 2 * p.t_halo);
        const int shift = (dt * H + dy) * W;
#pragma unroll
        for (int i = 0; i < MAXSLOT; ++i) {
            const bool ok = tok && sl_off[i] >= 0 && (unsigned)(sl_y[i] + dy) < (unsigned)H;
            if (ok) {
                const int64_t off = (int64_t)(sl_off[i] + shift) * ld + c0;
                rbh[i] = gload16(sh + off);
                rbl[i] = gload16(sl + off);
            } else {
                rbh[i] = (u32x4){0, 0, 0, 0};
                rbl[i] = (u32x4){0, 0, 0, 0};
            }
        }
    };
    auto store_b = [&](int stage) {
        char* s = sB + stage * 2 * bplane;
#pragma unroll
        for (int i = 0; i < MAXSLOT; ++i)
            if (sl_lds[i] >= 0) {
                *(u32x4*)(s + sl_lds[i]) = rbh[i];
                *(u32x4*)(s + bplane + sl_lds[i]) = rbl[i];
            }
    };

    // ---- this lane's two B-operand pixels (patch-linear id -> window row at kx = 0) ------------------------------
    int brow[2];
#pragma unroll
    for (int nb = 0; nb < 2; ++nb) {
        const int pid = wn * 64 + nb * 32 + r;
        brow[nb] = (pid >> g.logF) * g.WR + (pid & (F - 1));
    }

    f32x16 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = (f32x16){0};

    // k-step state of the step being LOADED: (trow, chunk, kx)
    // temporal taps that fall outside [0, T) for this tile's frame contribute zeros: skip their k-steps altogether
    // (T = 5: 24 % of the (5,1,1) GRU pass, 13 % of the 3x3x3 flow head).  Valid kz form one contiguous range.
    const int kz0 = (ht - tf - p.t_halo) > 0 ? (ht - tf - p.t_halo) : 0;
    const int kz1 = (ht + T + p.t_halo - 1 - tf) < (p.kt - 1) ? (ht + T + p.t_halo - 1 - tf) : (p.kt - 1);
    const int rs_per_kz = ((g.ysweep || g.hs2) ? 1 : p.kh) * g.nchunk;
    const int rs_end = (kz1 + 1) * rs_per_kz;
    const int kstride = KG * g.nslice;        // row-step = trow * nchunk + chunk; K-group kg of slice s takes every kstride-th one
    int rs = kz0 * rs_per_kz + (int)blockIdx.y * KG + kg;
    const int nsteps = fdiv(rs_end - kz0 * rs_per_kz, g.m_kstride) * g.ksw;      // identical for every group (rs_per_kz % kstride == 0)
    {
        const int trow = fdiv(rs, g.m_nchunk);
        load_a(rs * g.ksw);
        load_b(trow, rs - trow * g.nchunk);
    }
    CONV2_STAMP(1)
    store_a(0);
    store_b(0);
    __syncthreads();
    CONV2_STAMP(2)
    int bsel = 0, kx = 0, swx = 0, trow = 0;     // kx: tap index inside the window; trow: its LDS row offset
    for (int j = 0; j < nsteps; ++j) {
        const bool more = j + 1 < nsteps;
        const bool need_b = more && (kx + 1 == g.ksw);
        if (more) {
            const int nrs = need_b ? rs + kstride : rs;
            load_a(nrs * g.ksw + (need_b ? 0 : kx + 1));
            if (need_b) {
                const int trow = fdiv(nrs, g.m_nchunk);
                load_b(trow, nrs - trow * g.nchunk);
            }
        }
        const char* a_s = sA + (j & 1) * WM * A_BLK + wm * A_BLK;
        const char* b_s = sB + bsel * 2 * bplane;
#pragma unroll
        for (int k16 = 0; k16 < 2; ++k16) {
            bf16x8 ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int mb = 0; mb < 2; ++mb) {
                const int off = swz2(mb * 32 + r, 2 * k16 + h);
                ah[mb] = *(const bf16x8*)(a_s + off);
                al[mb] = *(const bf16x8*)(a_s + 4096 + off);
            }
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const int off = swz2(brow[nb] + trow, 2 * k16 + h);
                bh[nb] = *(const bf16x8*)(b_s + off);
                bl[nb] = *(const bf16x8*)(b_s + bplane + off);
            }
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb) {
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[mb], bh[nb], acc[mb][nb], 0, 0, 0);
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mb], bl[nb], acc[mb][nb], 0, 0, 0);
                    acc[mb][nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[mb], bh[nb], acc[mb][nb], 0, 0, 0);
                }
        }
        if (more) {
            store_a((j + 1) & 1);
            if (need_b) {
                if (g.bstages == 1) {
                    __syncthreads();          // every wave is done sweeping the single window before it is replaced
                    store_b(0);
                } else {
                    store_b(bsel ^ 1);
                }
            }
        }
        __syncthreads();
        if (need_b) {
            bsel = (g.bstages == 1) ? 0 : (bsel ^ 1);
            kx = swx = trow = 0;
            rs += kstride;
        } else {
            ++kx;
            ++trow;
            if (++swx == g.swn) {
                swx = 0;
                trow += g.jump;
            }
        }
    }

    CONV2_STAMP(3)
    // ---- intra-workgroup split-K: fixed-order sum of the K-groups' partial tiles through LDS ----------------------
    if (KG > 1) {
        // (the loop's last barrier guarantees nobody still reads the staging area that is reused here)
        float* red = (float*)smem;
        const int slot = ((kg - 1) * 2 * WM + wave) * 64 * 64;              // 64 regs x 64 lanes per wave
        if (kg > 0) {
#pragma unroll
            for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const f32x4 v4 = {acc[mb][nb][4 * q], acc[mb][nb][4 * q + 1], acc[mb][nb][4 * q + 2], acc[mb][nb][4 * q + 3]};
                        *(f32x4*)(red + slot + (((mb * 2 + nb) * 4 + q) * 64 + lane) * 4) = v4;
                    }
        }
        __syncthreads();
        if (kg == 0)
            for (int k = 1; k < KG; ++k) {
                const int src = ((k - 1) * 2 * WM + wave) * 64 * 64;
#pragma unroll
                for (int mb = 0; mb < 2; ++mb)
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb)
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            const f32x4 v4 = *(const f32x4*)(red + src + (((mb * 2 + nb) * 4 + q) * 64 + lane) * 4);
#pragma unroll
                            for (int e = 0; e < 4; ++e) acc[mb][nb][4 * q + e] += v4[e];
                        }
            }
        __syncthreads();                     // the exchange area is reused as the epilogue's staging patches
        if (kg > 0) return;
    }

    CONV2_STAMP(4)
    // ---- epilogue: accumulators -> wave-private LDS patch [32 px][64 couts] -> 8 couts of one pixel per lane -------
    // (the loop's / the reduction's last barrier guarantees nobody still reads the operand stages reused here; from
    // here on every wave touches only its own patch, and LDS operations of one wave execute in order)
    const int cblock = (mgrp * WM + wm) * 64;
    const int half = (cblock >= p.m_split) ? 1 : 0;
    const ppms_epilogue e = p.epi[half];          // BY VALUE (SGPRs): through a reference every field is re-read from memory behind every
                                                  // store of the row loop (the stores might alias the descriptor), one scalar-load round trip each
    const int cbase = cblock - (half ? p.m_split : 0);
    float* stg = (float*)(smem + wave * STG_WAVE);
    const int q = lane & 7;
    float b8[8];
    {
        const f32x4 b0 = gld<f32x4>(p.bias + cblock + q * 8), b1 = gld<f32x4>(p.bias + cblock + q * 8 + 4);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            b8[j] = b0[j];
            b8[4 + j] = b1[j];
        }
        // the bias must have LANDED before the row loop: vmcnt counts loads and stores in one order, so a wait for this load placed
        // inside the loop (where its first use is) is a wait for every store of the previous 8-row step too -- one HBM write round
        // trip (~1 us) per step, which is what the epilogues cost before this line
#pragma unroll
        for (int j = 0; j < 8; ++j) asm volatile("" : "+v"(b8[j]));
    }
    // the row loop exists twice: once for plain STORE epilogues, whose body holds no load (so nothing in it ever waits for the previous
    // step's stores: conv_epilogue.h), once for everything else
    auto rows = [&](auto cls_tag, auto grp_tag) {
        constexpr int CLS = decltype(cls_tag)::value, G = decltype(grp_tag)::value;
    #pragma unroll 1
        for (int nb = 0; nb < 2; ++nb) {
            // pixel-major V^T (the attention's value operand, bf16 [frame][cout][H*W]).  Fast form: from the staged patch below, a lane takes
            // ONE cout and 8 consecutive pixels per step: 16-byte stores (the accumulator layout gave 2-byte stores, 64 per lane: the epilogue
            // of the 128 -> 128 to_v GEMM cost 25-30 us at every scale, more than its K loop).  Needs rows of 8 aligned pixels.
            const bool vt_fast = !g.ysweep && F >= 8 && (W & 7) == 0 && (HW & 7) == 0;
            if (CLS == EPI_CLS_ANY && e.out_vt != nullptr && !vt_fast) {                       // (any geometry) straight from the accumulator layout
                const int pid = wn * 64 + nb * 32 + r;
                const int pf_ = pid & (F - 1), ps_ = pid >> g.logF;
                const int px = x0 + (g.ysweep ? ps_ : pf_), py = y0 + (g.ysweep ? pf_ : ps_);
    #pragma unroll
                for (int mb = 0; mb < 2; ++mb)
    #pragma unroll
                    for (int gq = 0; gq < 4; ++gq) {
                        const int c4 = mb * 32 + 8 * gq + 4 * h;
                        const f32x4 bb = gld<f32x4>(p.bias + cblock + c4);
                        float v4[4];
    #pragma unroll
                        for (int j = 0; j < 4; ++j) v4[j] = (nb ? acc[mb][1][4 * gq + j] : acc[mb][0][4 * gq + j]) + bb[j];
                        if (px < W && py < H) epilogue_vt4(e, v4, tf, py * W + px, cbase + c4, HW);
                    }
            }
    #pragma unroll
            for (int mb = 0; mb < 2; ++mb)
    #pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    f32x4 a4;
    #pragma unroll
                    for (int j = 0; j < 4; ++j) a4[j] = nb ? acc[mb][1][4 * gq + j] : acc[mb][0][4 * gq + j];
                    stage_write32(stg, r, h, mb, gq, a4);
                }
            __builtin_amdgcn_wave_barrier();
            if (CLS == EPI_CLS_ANY && e.out_vt != nullptr && vt_fast) {
                const int cl = cbase + lane;                    // this lane's cout (of the epilogue half)
                const float bc = gld<float>(p.bias + cblock + lane);
                bf16_t* vrow = (bf16_t*)e.out_vt + ((int64_t)tf * e.n_valid + cl) * HW;
    #pragma unroll
                for (int k8 = 0; k8 < 4; ++k8) {
                    const int pid = wn * 64 + nb * 32 + 8 * k8;
                    const int px = x0 + (pid & (F - 1)), py = y0 + (pid >> g.logF);
                    float y[8];
    #pragma unroll
                    for (int j = 0; j < 8; ++j) y[j] = stg[(8 * k8 + j) * STG_LD + lane] + bc;
                    apply_act_n<8>(y, e.act, e.scale);
                    bf16x8 o8;
    #pragma unroll
                    for (int j = 0; j < 8; ++j) o8[j] = vt_enc(y[j], e.vt_f16);
                    if (px < W && py < H && cl < e.n_valid) gst<bf16x8>(vrow + (int64_t)py * W + px, o8);
                }
            }
    #pragma unroll 1
            for (int it0 = 0; it0 < 4; it0 += G) {          // groups of G 8-row steps: operands first, then the stores (conv_epilogue.h)
                row8_aux aux[G];
                int64_t pixg[G];
                bool okg[G];
    #pragma unroll
                for (int gi = 0; gi < G; ++gi) {
                    const int pid = wn * 64 + nb * 32 + (it0 + gi) * 8 + (lane >> 3);
                    const int pf_ = pid & (F - 1), ps_ = pid >> g.logF;
                    const int px = x0 + (g.ysweep ? ps_ : pf_), py = y0 + (g.ysweep ? pf_ : ps_);
                    okg[gi] = px < W && py < H;
                    pixg[gi] = (int64_t)(tf * H + py) * W + px;
                    if (okg[gi] && g.nslice == 1) row8_fetch<CLS>(e, pixg[gi], cbase + q * 8, aux[gi]);
                }
    #pragma unroll
                for (int gi = 0; gi < G; ++gi) {
                    float v[8];
                    stage_read8(stg, (it0 + gi) * 8 + (lane >> 3), q, v);
                    if (okg[gi]) {
                        if (g.nslice > 1) {                              // K-sliced launch: raw partial sums, finished by the reduce kernel
                            float* pp = g.part + ((int64_t)blockIdx.y * g.P + pixg[gi]) * p.M + cblock + q * 8;
                            const f32x4 o0 = {v[0], v[1], v[2], v[3]}, o1 = {v[4], v[5], v[6], v[7]};
                            gst<f32x4>(pp, o0);
                            gst<f32x4>(pp + 4, o1);
                        } else {
    #pragma unroll
                            for (int j = 0; j < 8; ++j) v[j] += b8[j];
                            row8_finish<CLS>(e, v, pixg[gi], cbase + q * 8, HW, aux[gi]);
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    };
    using I0 = std::integral_constant<int, EPI_CLS_PLAIN>;
    using I1 = std::integral_constant<int, EPI_CLS_PRE>;
    using I2 = std::integral_constant<int, EPI_CLS_AUX>;
    using I3 = std::integral_constant<int, EPI_CLS_GRU>;
    using I4 = std::integral_constant<int, EPI_CLS_ANY>;
    using I5 = std::integral_constant<int, EPI_CLS_AUXPRE>;
    using G2 = std::integral_constant<int, 2>;
    using G1 = std::integral_constant<int, 1>;
    using G4 = std::integral_constant<int, 4>;
    const int cls = g.nslice > 1 ? (int)EPI_CLS_PLAIN : epilogue_class(e);       // (a K-sliced launch stores raw partials: no operand to load)
    if (cls == EPI_CLS_PLAIN) rows(I0{}, G1{});
    else if (cls == EPI_CLS_PRE) rows(I1{}, G4{});
    else if (cls == EPI_CLS_AUX) rows(I2{}, G4{});
    else if (cls == EPI_CLS_GRU) rows(I3{}, G1{});
    else if (cls == EPI_CLS_AUXPRE) rows(I5{}, G2{});
    else rows(I4{}, G1{});
    CONV2_STAMP(5)
}

// Second half of a K-sliced convolution: sums the slices' partial tiles in slice order (deterministic), adds the bias and
// runs the conv's own fused epilogue.  One thread = one pixel x 8 couts (coalesced 32-byte reads per slice).
__global__ __launch_bounds__(256) void conv_slice_reduce_kernel(const ppms_conv pv, const float* __restrict__ part, int nslice,
                                                                int64_t P) {
    const ppms_conv& p = pv;
    const int groups = p.M >> 3;
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= P * groups) return;
    const int c8 = (int)(idx % groups) * 8;
    const int64_t pix = idx / groups;
    float v[8];
    {
        const f32x4 b0 = gld<f32x4>(p.bias + c8), b1 = gld<f32x4>(p.bias + c8 + 4);
        f32x4 s0 = {0.0f, 0.0f, 0.0f, 0.0f}, s1 = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int s = 0; s < nslice; ++s) {
            const float* pp = part + ((int64_t)s * P + pix) * p.M + c8;
            s0 += *(const f32x4*)pp;
            s1 += *(const f32x4*)(pp + 4);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] = s0[j] + b0[j];
            v[4 + j] = s1[j] + b1[j];
        }
    }
    const int half = (c8 >= p.m_split) ? 1 : 0;
    epilogue_row8(p.epi[half], v, pix, c8 - (half ? p.m_split : 0), p.H * p.W);
}

template <int WM, int KG>
int launch2(const ppms_conv* d, const ppms_conv* dev_desc, const Geo2& g, int ntiles, hipStream_t stream) {
    size_t lds = (size_t)KG * ((size_t)2 * WM * A_BLK + (size_t)g.bstages * 2 * g.Wr * 64);
    const size_t red = (size_t)(KG - 1) * 2 * WM * 64 * 64 * 4;          // partial-accumulator exchange reuses the staging area
    if (red > lds) lds = red;
    const size_t stg = (size_t)2 * WM * STG_WAVE;                        // so do the epilogue's transposition patches
    if (stg > lds) lds = stg;
    static ppms_device_once once;                                        // one per template instantiation
    once.run([] { (void)hipFuncSetAttribute((const void*)conv2_kernel<WM, KG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    if (lds > 160 * 1024) {
        ppms_set_error("conv_gemm2: LDS budget exceeded (%zu B)", lds);
        return PPMS_EINVAL;
    }
#ifdef PPMS_CONV2_TIMING
    Geo2 gd = g;
    gd.dbg = g_conv2_dbg;
    hipLaunchKernelGGL((conv2_kernel<WM, KG>), dim3(ntiles * g.mgroups, g.nslice), dim3(128 * WM * KG), lds, stream, *d, gd);
    return ppms_check_launch("conv_gemm2");
#endif
    hipLaunchKernelGGL((conv2_kernel<WM, KG>), dim3(ntiles * g.mgroups, g.nslice), dim3(128 * WM * KG), lds, stream, *d, g);
    return ppms_check_launch("conv_gemm2");
}

}  // namespace

static int conv2_launch(const ppms_conv* d, const ppms_conv* dev_desc, int wm_hint, int nslice, float* part, void* stream, int ysweep = 0);
#ifdef PPMS_CONV2_TIMING
extern "C" void ppms_debug_conv2_timing(long long* p) { g_conv2_dbg = p; }      // debug builds only (tools/conv2_phase_probe.py)
#endif

// wm_hint: 0 = choose (all couts per workgroup when the grid still fills the chip, otherwise 64-cout blocks)
extern "C" int ppms_conv_gemm2(const ppms_conv* d, const ppms_conv* dev_desc, int wm_hint, void* stream) {
    return conv2_launch(d, dev_desc, wm_hint, 1, nullptr, stream);
}

// Small maps (fewer workgroups than the chip has CUs, long K loops): how many grid-level K slices pay off.  1 = none.
extern "C" int ppms_conv_gemm2_slices(const ppms_conv* d) {
    if (d == nullptr || d->nseg < 1 || d->nseg > 2 || d->M <= 0 || d->M % 64 != 0) return 1;
    if (d->epi[0].out_vt != nullptr || (d->m_split < d->M && d->epi[1].out_vt != nullptr)) return 1;   // V^T is written from the accumulators
    int nchunk = 0;
    for (int s = 0; s < d->nseg; ++s) nchunk += d->seg[s].c / BK;
    const int64_t P = (int64_t)d->T * d->H * d->W;
    const int64_t nwg = (P + 127) / 128 * (d->M / 64);              // 64-cout x 128-pixel workgroups without slicing
    if (nwg >= 512) return 1;
    const int rs_per_kz = d->kh * nchunk;
    const int64_t steps = (int64_t)d->kt * rs_per_kz * d->kw;        // k-steps of the whole K loop
    // Short K loops: the unsliced launch splits K INSIDE the workgroup (2 or 4 K-groups, conv2_launch) and needs no reduce launch; it wins
    // when that leaves <= 10 k-steps per K-group, the workgroups fit one round on the chip and still make >= 512 waves (tools/slice_tune.py,
    // 1/16 and 1/8 scales of config 2: mask head 25.8 -> 21.6 us, the GRU's (1,1,5) tails 26 -> 18 us, the 384-wide 1x1 GEMMs 21.7 -> 16.5 us)
    {
        const int64_t ntiles = (P + 127) / 128;
        const int kg = (ntiles < 160 || d->M == 64) ? ((nwg <= 128 && nchunk % 4 == 0) ? 4 : (nwg <= 512 && nchunk % 2 == 0) ? 2 : 1) : 1;
        if (kg > 1 && steps / kg <= 10 && nwg <= 256 && nwg * kg * 2 >= 512) return 1;
    }
    int best = 1;
    for (int s = 2; s <= 8; ++s)
        if (rs_per_kz % s == 0 && nwg * s <= 1024 && steps / s >= 6) best = s;
    return best;
}

// Same planner for the y-swept and the 2-D window forms (row-steps per temporal tap = chunks only)
extern "C" int ppms_conv_gemm2_ysweep_slices(const ppms_conv* d) {
    if (d == nullptr || d->kh <= 1) return 0;
    if (d->kw > 1) {                   // 2-D window: some 128-pixel patch with its halo must fit the staging slots
        bool fits = false;
        for (int C = 16; C <= 128; C *= 2) fits = fits || (C + d->kw - 1) * (128 / C + d->kh - 1) <= 256;
        if (!fits) return 0;
    }
    ppms_conv t = *d;
    t.kw = d->kh * d->kw;              // as an x-swept (1, 1, taps) conv: same number of workgroups, row-steps and k-steps
    t.kh = 1;
    return ppms_conv_gemm2_slices(&t);
}

extern "C" int64_t ppms_conv_gemm2_slice_workspace_bytes(const ppms_conv* d, int nslice) {
    if (d == nullptr || nslice <= 1) return 0;
    return (int64_t)nslice * d->T * d->H * d->W * d->M * 4;
}

// second half of every K-sliced launch (also conv_gemm5.hip's): sums the slices' partial tiles, adds the bias, runs the fused epilogue
int ppms_launch_slice_reduce(const ppms_conv* d, const ppms_conv* dev_desc, const float* workspace, int nslice, void* stream) {
    const int64_t P = (int64_t)d->T * d->H * d->W;
    const int64_t total = P * (d->M / 8);
    hipLaunchKernelGGL(conv_slice_reduce_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, *d, workspace, nslice, P);
    return ppms_check_launch("conv_slice_reduce");
}

// K-sliced form for small maps: nslice workgroups share each output tile (each takes every nslice-th row-step of the K loop
// and writes fp32 partial sums to `workspace`), then a reduce kernel sums them in slice order and runs the fused epilogue.
extern "C" int ppms_conv_gemm2_sliced(const ppms_conv* d, const ppms_conv* dev_desc, int nslice, void* workspace, void* stream) {
    PPMS_REQUIRE(nslice >= 1 && nslice <= 16, "conv_gemm2_sliced: nslice=%d", nslice);
    if (nslice == 1) return conv2_launch(d, dev_desc, 0, 1, nullptr, stream);
    PPMS_REQUIRE(workspace != nullptr && ((uintptr_t)workspace & 15) == 0, "conv_gemm2_sliced: workspace missing or not 16-B aligned");
    const int rc = conv2_launch(d, dev_desc, 1, nslice, (float*)workspace, stream);
    if (rc != 0) return rc;
    const int64_t P = (int64_t)d->T * d->H * d->W;
    const int64_t total = P * (d->M / 8);
    hipLaunchKernelGGL(conv_slice_reduce_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, *d, (const float*)workspace, nslice, P);
    return ppms_check_launch("conv_gemm2_sliced");
}

// Convs with kh > 1 whose taps all step ONE halo'd window per (dt, chunk) (the plain entry loads a window per kernel row):
// (kt, kh, 1) kernels are swept along y, weights packed with kh / kw swapped; kernels with kw > 1 too use a 2-D window, weights
// packed with (ky, kx) flattened into x -- the sweep-ordered packs conv_gemm5 / conv_gemm6 use too.  nslice as in ppms_conv_gemm2_sliced.
extern "C" int ppms_conv_gemm2_ysweep(const ppms_conv* d, const ppms_conv* dev_desc, int nslice, void* workspace, void* stream) {
    PPMS_REQUIRE(d != nullptr && d->kh > 1, "conv_gemm2_ysweep: needs a kernel with kh > 1");
    const int mode = d->kw > 1 ? 2 : 1;              // kw > 1: 2-D window over all kh x kw taps
    PPMS_REQUIRE(nslice >= 1 && nslice <= 16, "conv_gemm2_ysweep: nslice=%d", nslice);
    if (nslice == 1) return conv2_launch(d, dev_desc, 0, 1, nullptr, stream, mode);
    PPMS_REQUIRE(workspace != nullptr && ((uintptr_t)workspace & 15) == 0, "conv_gemm2_ysweep: workspace missing or not 16-B aligned");
    const int rc = conv2_launch(d, dev_desc, 1, nslice, (float*)workspace, stream, mode);
    if (rc != 0) return rc;
    const int64_t P = (int64_t)d->T * d->H * d->W;
    const int64_t total = P * (d->M / 8);
    hipLaunchKernelGGL(conv_slice_reduce_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, *d, (const float*)workspace, nslice, P);
    return ppms_check_launch("conv_gemm2_ysweep");
}

static int conv2_launch(const ppms_conv* d, const ppms_conv* dev_desc, int wm_hint, int nslice, float* part, void* stream, int ysweep) {
    PPMS_REQUIRE(d != nullptr && dev_desc != nullptr, "conv_gemm2: null descriptor (host copy and device copy are both required)");
    PPMS_REQUIRE(d->nseg == 1 || d->nseg == 2, "conv_gemm2: nseg=%d", d->nseg);
    PPMS_REQUIRE(d->groups <= 1, "conv_gemm2: a grouped convolution (groups=%d) is served by ppms_conv_gemm6 only", d->groups);
    PPMS_REQUIRE(d->T > 0 && d->H > 0 && d->W > 0, "conv_gemm2: bad volume %dx%dx%d", d->T, d->H, d->W);
    PPMS_REQUIRE(d->t_halo >= 0 && d->t_halo <= 8, "conv_gemm2: t_halo=%d", d->t_halo);
    PPMS_REQUIRE((d->kt & 1) && (d->kh & 1) && (d->kw & 1) && d->kw <= 15, "conv_gemm2: kernel extents must be odd, kw <= 15");
    PPMS_REQUIRE(d->M > 0 && d->M % 64 == 0 && d->m_split % 64 == 0, "conv_gemm2: M=%d / m_split=%d not multiples of 64", d->M, d->m_split);
    PPMS_REQUIRE(d->w != nullptr && d->bias != nullptr, "conv_gemm2: weights/bias missing");
    // pixel INDICES are 32-bit inside the kernel (slot offsets, (dt, dy) shifts of up to kt / 2 frames); every byte offset is formed in 64 bits
    PPMS_REQUIRE((int64_t)(d->T + 2 * d->t_halo + d->kt) * d->H * d->W < (1ll << 30), "conv_gemm2: volume too large for 32-bit pixel indices");
    int nchunk = 0;
    for (int s = 0; s < d->nseg; ++s) {
        PPMS_REQUIRE(d->seg[s].hi && d->seg[s].lo && d->seg[s].c > 0 && d->seg[s].c % BK == 0 && d->seg[s].ld % 8 == 0,
                     "conv_gemm2: segment %d needs hi/lo planes, c %% 32 == 0 and ld %% 8 == 0 (c=%d ld=%d)", s, d->seg[s].c, d->seg[s].ld);
        PPMS_REQUIRE(((uintptr_t)d->seg[s].hi & 15) == 0 && ((uintptr_t)d->seg[s].lo & 15) == 0, "conv_gemm2: segment %d not 16-B aligned", s);
        nchunk += d->seg[s].c / BK;
    }
    for (int hlf = 0; hlf < 2; ++hlf) {
        const ppms_epilogue& e = d->epi[hlf];
        if (hlf == 1 && d->m_split >= d->M) break;
        PPMS_REQUIRE(e.n_valid > 0, "conv_gemm2: epilogue %d has n_valid=%d", hlf, e.n_valid);
        PPMS_REQUIRE(e.pre_f32 == nullptr || (e.n_valid % 4 == 0 && e.pre_f32_ld % 4 == 0), "conv_gemm2: pre_f32 needs n_valid and pre_f32_ld to be multiples of 4");
        {
            const char* why = epilogue_row8_check(e);
            PPMS_REQUIRE(why == nullptr, "conv_gemm2: epilogue %d: %s", hlf, why ? why : "");
        }
        if (e.out_sp.hi) PPMS_REQUIRE(e.out_sp.lo && e.out_sp.ld % 4 == 0 && ((uintptr_t)e.out_sp.hi & 7) == 0 && ((uintptr_t)e.out_sp.lo & 7) == 0,
                                      "conv_gemm2: epilogue %d SP output misaligned", hlf);
        if (e.out_f32) PPMS_REQUIRE(e.out_f32_ld % 4 == 0 || e.kind == PPMS_EPI_ADDF32, "conv_gemm2: epilogue %d f32 ld", hlf);
        if (e.kind == PPMS_EPI_RESID || e.kind == PPMS_EPI_RH || e.kind == PPMS_EPI_GRU)
            PPMS_REQUIRE(e.aux_sp.hi && e.aux_sp.lo && e.aux_sp.ld % 4 == 0, "conv_gemm2: epilogue %d needs aux_sp", hlf);
        if (e.kind == PPMS_EPI_GRU) PPMS_REQUIRE(e.aux_f32 != nullptr, "conv_gemm2: GRU epilogue needs z");
        if (e.kind == PPMS_EPI_ADDF32) PPMS_REQUIRE(e.out_f32 != nullptr, "conv_gemm2: ADDF32 epilogue needs out_f32");
    }
    // patch width: the power of two in [16,128] wasting the fewest pixels of the 128-pixel tile (ties: the widest for an x
    // sweep, the narrowest -- tallest patch, smallest halo share -- for a y sweep)
    // ysweep: 0 = taps along x (one window per (dt, dy)); 1 = (kt, kh, 1) kernel swept along y; 2 = 2-D window: all kh x kw taps of
    // a (dt, chunk) step one window with a halo in x and y (weights packed with (ky, kx) flattened into x)
    const bool win2d = ysweep == 2;
    ysweep = ysweep == 1;
    PPMS_REQUIRE(!ysweep || (d->kw == 1 && d->kh > 1), "conv_gemm2: y sweep is for (kt, kh, 1) kernels");
    PPMS_REQUIRE(!win2d || (d->kw > 1 && d->kh > 1), "conv_gemm2: the 2-D window is for kernels with kh > 1 and kw > 1");
    Geo2 g;
    int bestC = 16;
    double bestw = 1e30;
    for (int C = 16; C <= 128; C *= 2) {
        const int R = 128 / C;
        if (ysweep && C * (R + d->kh - 1) > 256) continue;          // window rows must fit the staging slots of one cout block
        if (win2d && (C + d->kw - 1) * (R + d->kh - 1) > 256) continue;
        const double waste = (double)((d->W + C - 1) / C * C) * ((d->H + R - 1) / R * R) / ((double)d->W * d->H);
        if (ysweep ? waste < bestw - 1e-9 : waste <= bestw + 1e-9) {
            bestw = waste;
            bestC = C;
        }
    }
    PPMS_REQUIRE(bestw < 1e29, "conv_gemm2: no patch shape fits the window");
    g.C = bestC;
    g.R = 128 / bestC;
    g.logC = 0;
    while ((1 << g.logC) < g.C) ++g.logC;
    g.ysweep = ysweep;
    g.ksw = ysweep ? d->kh : win2d ? d->kh * d->kw : d->kw;
    g.logF = ysweep ? 7 - g.logC : g.logC;
    g.tiles_x = (d->W + g.C - 1) / g.C;
    g.tiles_y = (d->H + g.R - 1) / g.R;
    g.swn = ysweep ? d->kh : d->kw;
    g.WR = (ysweep ? g.R : g.C) + g.swn - 1;
    g.hs2 = win2d ? d->kh >> 1 : 0;
    g.jump = win2d ? g.WR - d->kw : 0;
    g.Wr = (ysweep ? g.C : win2d ? g.R + d->kh - 1 : g.R) * g.WR;
    g.nchunk = nchunk;
    g.n0 = d->seg[0].c / BK;
    g.nk = d->kt * d->kh * nchunk * d->kw;
    g.nslice = nslice;
    g.wr_magic = (unsigned)(((1u << 20) + g.WR - 1) / g.WR);
    for (int wrow = 0; wrow < g.Wr; ++wrow)             // exactness over the rows that exist (Wr <= 2048, WR <= 142: holds; kept as a guard)
        if ((int)(((unsigned)wrow * g.wr_magic) >> 20) != wrow / g.WR) {
            g.wr_magic = 0;
            break;
        }
    g.part = part;
    g.P = (int64_t)d->T * d->H * d->W;
    if (nslice > 1) {
        PPMS_REQUIRE((((ysweep || win2d) ? 1 : d->kh) * nchunk) % nslice == 0, "conv_gemm2: %d row-steps per temporal tap do not split into %d slices",
                     ((ysweep || win2d) ? 1 : d->kh) * nchunk, nslice);
        PPMS_REQUIRE(d->epi[0].out_vt == nullptr && (d->m_split >= d->M || d->epi[1].out_vt == nullptr), "conv_gemm2: sliced launch cannot write out_vt");
    }
    const int ntiles = g.tiles_x * g.tiles_y * d->T;
    const int mblocks = d->M / 64;
    int wm = wm_hint;
    if (wm <= 0) {
        // <= 3 cout blocks: one group; 4 blocks (M = 256): two 128-cout groups (measured 8 % faster than one 8-wave group)
        wm = mblocks <= 3 ? mblocks : 2;
        if (mblocks == 4 && d->kw == 1 && d->kh == 1) wm = 4;   // no window reuse across taps (1x1, temporal): share each window among all couts
        if (mblocks % wm) wm = 1;
        if (ntiles < 160 && mblocks > 1) wm = 1;          // small maps: spread cout blocks over more workgroups
    }
    PPMS_REQUIRE(wm >= 1 && wm <= 4 && mblocks % wm == 0, "conv_gemm2: wm=%d does not divide M/64=%d", wm, mblocks);
    g.mgroups = mblocks / wm;
    // kw > 1: the window changes every kw-th k-step -> keep ONE copy (extra barrier per switch) so that three 4-wave
    // workgroups fit a CU's LDS; kw == 1: it changes every k-step -> double-buffer it
    g.bstages = (g.ksw == 1) ? 2 : 1;
    const int maxslot = wm == 4 ? 2 : wm == 3 ? 3 : wm == 2 ? 4 : 8;
    PPMS_REQUIRE(g.Wr * 4 <= 128 * wm * maxslot, "conv_gemm2: window of %d rows does not fit the staging slots", g.Wr);
    PPMS_REQUIRE(2 * wm * A_BLK + g.bstages * 2 * g.Wr * 64 <= 160 * 1024, "conv_gemm2: LDS budget exceeded");
    hipStream_t st = (hipStream_t)stream;
    // small maps: split K inside the workgroup so that more waves work on the few tiles there are
    const int nwg = ntiles * g.mgroups;
    int kgs = 1;
    if (wm == 1 && wm_hint <= 0 && nslice == 1) {
        if (nwg <= 128 && nchunk % 4 == 0) kgs = 4;
        else if (nwg <= 512 && nchunk % 2 == 0) kgs = 2;   // ~68 KiB of LDS each: two such workgroups share a CU
    }
    if (kgs > 1) g.bstages = 1;                // K-groups keep ONE window copy each (LDS budget), at one more barrier per window
    {   // multiply-high reciprocals of the kernel's runtime divisors (fdiv)
        auto magic = [](int dd) { return dd == 1 ? 0u : (unsigned)((1ull << 32) / (unsigned)dd) + 1u; };
        g.kho = (ysweep || win2d) ? 1 : d->kh;
        const int64_t rows_all = (int64_t)d->kt * d->kh * nchunk;
        const int kstride = kgs * nslice;
        PPMS_REQUIRE((int64_t)nwg * g.mgroups < (1ll << 32) && (int64_t)ntiles * g.tiles_x < (1ll << 32) && rows_all * nchunk < (1ll << 32) &&
                         rows_all * kstride < (1ll << 32),
                     "conv_gemm2: geometry too large for the 32-bit reciprocal divisions");
        g.m_mgroups = magic(g.mgroups), g.m_tiles_x = magic(g.tiles_x), g.m_tiles_y = magic(g.tiles_y), g.m_nchunk = magic(nchunk), g.m_kho = magic(g.kho),
        g.m_kstride = magic(kstride);
    }
    if (kgs == 4) return launch2<1, 4>(d, dev_desc, g, ntiles, st);
    if (kgs == 2) return launch2<1, 2>(d, dev_desc, g, ntiles, st);
    switch (wm) {
        case 1: return launch2<1, 1>(d, dev_desc, g, ntiles, st);
        case 2: return launch2<2, 1>(d, dev_desc, g, ntiles, st);
        case 3: return launch2<3, 1>(d, dev_desc, g, ntiles, st);
        default: return launch2<4, 1>(d, dev_desc, g, ntiles, st);
    }
}
