// 1-D all-pairs correlation pyramid: build (fp32 MFMA, exact fp32 FMA chain) and multi-level lookup.
// Replaces CorrBlock1D of /root/reference/models/core/corr.py:55-104 (einsum :102, avg_pool2d :71,
// grid_sample :21).  HBM-bound kernels: the volume is written once and gathered once per iteration.
#include "common.h"
#include "corr_lookup.h"

// ------------------------------------------------------------------------------------------------
// build: one workgroup per (epipolar line, 32-wide x1 tile); each wave owns 32-wide x2 tiles.
// v_mfma_f32_32x32x2_f32 takes A[i=l&31][k=l>>5], B[k=l>>5][j=l&31]: with NCHW features both operands are
// coalesced 128-B row segments of one channel, no transpose and no LDS staging needed.
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void corr_build_kernel(const float* __restrict__ f1, const float* __restrict__ f2,
                                                         float* __restrict__ p0, float* __restrict__ p1, float* __restrict__ p2,
                                                         float* __restrict__ p3, float* __restrict__ p4, int C, int H, int W) {
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    // XCD-aware order: workgroups go to the 8 XCDs round-robin by dispatch index and every XCD has its own L2.  The x1 tiles of one
    // epipolar line all stream the same f2 line (C x W floats), so they are dealt to ONE XCD: of 8 * ntx consecutive workgroups, number
    // 8 t + k is tile t of line k of the group (plain order: each XCD fetches every f2 line itself, ntx times the traffic)
    int row, tx;
    {
        const int ntx = gridDim.x, nrow = gridDim.y;
        const int lin = blockIdx.x + ntx * blockIdx.y;
        const int grp = lin / (8 * ntx), in = lin - grp * 8 * ntx;
        const int full = nrow / 8;                       // groups of 8 lines; the last nrow % 8 lines keep the plain order
        if (grp < full) {
            row = grp * 8 + (in & 7);
            tx = in >> 3;
        } else {
            const int rest = lin - full * 8 * ntx;
            row = full * 8 + rest / ntx;
            tx = rest - (rest / ntx) * ntx;
        }
    }
    const int b = row / H, y = row - b * H;
    const int i0 = tx * 32;
    const int li = lane & 31, lk = lane >> 5;
    const int64_t chan_stride = (int64_t)H * W;
    const float* a_base = f1 + ((int64_t)b * C * H + y) * W;      // + c*chan_stride + x
    const float* b_base = f2 + ((int64_t)b * C * H + y) * W;
    const int xa = i0 + li;
    const bool va = xa < W;
    const float* ap = a_base + (va ? xa : 0) + (int64_t)lk * chan_stride;
    const float inv = sqrtf((float)C);
    const int W1 = W >> 1, W2 = W1 >> 1, W3 = W2 >> 1, W4 = W3 >> 1;
    const int ntile = (W + 31) >> 5;
    for (int jt = wave; jt < ntile; jt += 4) {
        const int j0 = jt * 32;
        const int xb = j0 + li;
        const bool vb = xb < W;
        const float* bp = b_base + (vb ? xb : 0) + (int64_t)lk * chan_stride;
        f32x16 acc = {0};
        for (int c = 0; c < C; c += 16) {
            float av[8], bv[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                const int cc = c + 2 * u;
                const bool vc = (cc + lk) < C;
                av[u] = (va && vc) ? ap[(int64_t)cc * chan_stride] : 0.0f;
                bv[u] = (vb && vc) ? bp[(int64_t)cc * chan_stride] : 0.0f;
            }
#pragma unroll
            for (int u = 0; u < 8; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av[u], bv[u], acc, 0, 0, 0);
        }
        // D: col = lane&31 (x2), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (x1)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = i0 + (r & 3) + 8 * (r >> 2) + 4 * lk;
            const float v0 = acc[r] / inv;
            const float v1 = 0.5f * (v0 + __shfl_xor(v0, 1));
            const float v2 = 0.5f * (v1 + __shfl_xor(v1, 2));
            const float v3 = 0.5f * (v2 + __shfl_xor(v2, 4));
            const float v4 = 0.5f * (v3 + __shfl_xor(v3, 8));
            if (i < W) {
                const int64_t prow = (int64_t)row * W + i;
                if (xb < W) p0[prow * W + xb] = v0;
                if ((li & 1) == 0 && (xb >> 1) < W1) p1[prow * W1 + (xb >> 1)] = v1;
                if ((li & 3) == 0 && (xb >> 2) < W2) p2[prow * W2 + (xb >> 2)] = v2;
                if ((li & 7) == 0 && (xb >> 3) < W3) p3[prow * W3 + (xb >> 3)] = v3;
                if ((li & 15) == 0 && (xb >> 4) < W4) p4[prow * W4 + (xb >> 4)] = v4;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// build, line-resident form (round 4; W % 4 == 0, C % 16 == 0): one workgroup per (epipolar line, block of R1 <= 128 x1 positions); it
// walks the x2 positions of the line in blocks of R2 = R1 (one block at config 2's widths: every feature element is then fetched exactly
// once).  Features travel 16 bytes per lane straight into LDS (LDS-DMA, a ring of four 16-channel stages requested three stages ahead,
// ONE barrier per stage; with several x2 blocks the left block is re-staged per block out of L2); each wave keeps one or two 32 x 32
// tiles of the block pair in its accumulators over the whole channel loop (same k order and the same exact-fp32 MFMA as the kernel
// above: identical bits).  (Measured and dropped: 64-wide x2 blocks under 128-wide x1 blocks, so that a block's stores drain under the
// next block's MFMAs -- twice the barriers and a re-staged left block cost more than the overlap returns, 75 vs 62 us.)  The product is formed TRANSPOSED (A = right features, B = left features), so a lane ends up with 4
// consecutive x2 of one x1: level 0 leaves as 16-byte stores, levels 1 and 2 are sums inside the lane, level 3 needs the lane's partner
// (lane ^ 32), level 4 again the lane alone.  In the old form every operand value was a 4-byte load repeated by the 4 waves of a
// workgroup and by the W / 32 workgroups of a line, and every pyramid value a 4-byte store.
// vmcnt counts loads AND stores in one order, so the wait for a stage that was requested BEFORE an epilogue must allow for that
// epilogue's stores behind it: every store of the epilogue is an inline-asm instruction issued by every lane (lanes without a valid
// position write to a sink line), which makes their number a compile-time constant (S = 18 per tile); for three stages after an epilogue
// the wait is vmcnt(2 PPT + S), otherwise vmcnt(2 PPT).
// Algorithmic bytes: 2 * 256 * P * 4 in + 1.9375 * P * W * 4 out; at config 2's 1/4 scale 105 MB + 51 MB; the fp32 matrix
// pipe needs 2 * 256 * P * W FLOP / 157 TFLOP/s = 21 us for the same launch, HBM at ~6.3 TB/s achievable 25 us: a balanced kernel.
// ------------------------------------------------------------------------------------------------
#ifndef CORR_ABL
#define CORR_ABL 0                // ablation builds (timing only, wrong results): 1 no pyramid stores, 2 no MFMAs, 4 no feature DMA
#endif
constexpr int CORR_CK = 16;       // channels per LDS stage
constexpr int CORR_NST = 4;       // stages in the ring (three requested ahead)
constexpr int CORR_S = 18;        // store instructions of one tile's epilogue: 4 x (level 0, 1, 2) + 4 (level 3) + 2 (level 4)

__device__ __attribute__((aligned(256))) unsigned int g_corr_zero_page[64];       // source of the padding pieces (x >= W, stages past the end)
// where lanes without a valid position store: 1 KiB (16 B per lane) per workgroup slot, 1024 slots, so that the ragged blocks of different
// workgroups do not hammer one cache line (a single shared sink cost 10 % at W = 320)
__device__ __attribute__((aligned(256))) unsigned int g_corr_sink[1024 * 64 * 4];

// (the s_nop: a store of more than 64 bits must not have its data registers rewritten in the next wait states -- a hazard the compiler pads
//  for its own stores and cannot see inside inline asm; without it lanes 12-15 of the first store of an epilogue lost their first dword)
__device__ __forceinline__ void corr_st16(float* p, f32x4 v) { asm volatile("global_store_dwordx4 %0, %1, off\n\ts_nop 1" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void corr_st8(float* p, f32x2 v) { asm volatile("global_store_dwordx2 %0, %1, off" ::"v"(p), "v"(v) : "memory"); }
__device__ __forceinline__ void corr_st4(float* p, float v) { asm volatile("global_store_dword %0, %1, off" ::"v"(p), "v"(v) : "memory"); }

template <int R1LOG, int R2LOG>
struct CorrGeo {
    static constexpr int R1 = 1 << R1LOG, R2 = 1 << R2LOG;
    static constexpr int S1 = R1 / 32, S2 = R2 / 32;           // 32-wide tiles per block side
    static constexpr int TPW = S1 * S2 == 16 ? 2 : 1;           // tiles per multiplying wave
    static constexpr int NWC = S1 * S2 / TPW;                   // multiplying waves
    static constexpr int NW = NWC == 1 ? 2 : NWC;               // (one tile: a second wave only moves data)
    static constexpr int NT = NW * 64;
    static constexpr int Q1 = CORR_CK * R1 / 4, Q2 = CORR_CK * R2 / 4;      // 16-byte pieces of the left / right part of a stage
    static constexpr int PPT = (Q1 + Q2 + NT - 1) / NT;         // pieces per thread and stage (the last ones may be padding)
    static constexpr int STAGE = PPT * NT * 16;                 // bytes
    static constexpr size_t LDS = (size_t)CORR_NST * STAGE;
};

template <int R1LOG, int R2LOG>
__global__ __launch_bounds__((CorrGeo<R1LOG, R2LOG>::NT)) void corr_build_line_kernel(
    const float* __restrict__ f1, const float* __restrict__ f2, float* __restrict__ p0, float* __restrict__ p1, float* __restrict__ p2,
    float* __restrict__ p3, float* __restrict__ p4, int C, int H, int W, int nblk2) {
    using G = CorrGeo<R1LOG, R2LOG>;
    constexpr int R1 = G::R1, R2 = G::R2, NT = G::NT, PPT = G::PPT, STAGE = G::STAGE, TPW = G::TPW;
    static_assert(PPT == 2, "the counted vmcnt waits below are written for two pieces per thread and stage");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lk = lane >> 5;
    // XCD-aware order as above: the x1 blocks of one line (they stream the same right-feature line) go to one XCD
    int row, bx;
    {
        const int ntx = gridDim.x, nrow = gridDim.y;
        const int lin = blockIdx.x + ntx * blockIdx.y;
        const int grp = lin / (8 * ntx), in = lin - grp * 8 * ntx;
        const int full = nrow / 8;
        if (grp < full) {
            row = grp * 8 + (in & 7);
            bx = in >> 3;
        } else {
            const int rest = lin - full * 8 * ntx;
            row = full * 8 + rest / ntx;
            bx = rest - (rest / ntx) * ntx;
        }
    }
    const int b = row / H, y = row - b * H;
    const int64_t chan_stride = (int64_t)H * W;
    const float* line1 = f1 + ((int64_t)b * C * H + y) * W;      // + c * chan_stride + x
    const float* line2 = f2 + ((int64_t)b * C * H + y) * W;
    const int x1b = bx * R1;
    const int nch = C / CORR_CK, total = nblk2 * nch;

    // ---- the DMA pieces of this thread: LDS piece q = tid + s * NT; q < Q1: left block (channel q / (R1/4), 4 pixels), then the right
    // block, then padding.  Odd channel rows of a block wider than 32 pixels are stored with x ^ 32: the two channels a wave reads
    // together (lanes 0-31: channel c, lanes 32-63: channel c + 1) then sit in different bank halves
    int pc[PPT], px[PPT], pop[PPT];
#pragma unroll
    for (int s = 0; s < PPT; ++s) {
        const int q = tid + s * NT;
        if (q < G::Q1) {
            pop[s] = 0, pc[s] = q / (R1 / 4);
            px[s] = ((q % (R1 / 4)) * 4) ^ ((R1 >= 64 && (pc[s] & 1)) ? 32 : 0);
        } else if (q < G::Q1 + G::Q2) {
            const int q2 = q - G::Q1;
            pop[s] = 1, pc[s] = q2 / (R2 / 4);
            px[s] = ((q2 % (R2 / 4)) * 4) ^ ((R2 >= 64 && (pc[s] & 1)) ? 32 : 0);
        } else {
            pop[s] = 2, pc[s] = 0, px[s] = 0;                 // padding piece: always the zero page
        }
    }
    const char* zpage = (const char*)g_corr_zero_page;
    asm volatile("" : "+s"(zpage));
    auto issue = [&](int it) {
        const bool live = it < total && !(CORR_ABL & 4);
        const int bj = it < total ? it / nch : 0;
        const int c0 = (it - bj * nch) * CORR_CK;
        char* dst = smem + (it & (CORR_NST - 1)) * STAGE + wave * 1024;
#pragma unroll
        for (int s = 0; s < PPT; ++s) {
            const int x = (pop[s] == 1 ? bj * R2 : x1b) + px[s];
            const float* src = (pop[s] == 1 ? line2 : line1) + (int64_t)(c0 + pc[s]) * chan_stride + x;
            const void* ps = (live && pop[s] < 2 && x < W) ? (const void*)src : (const void*)zpage;
            __builtin_amdgcn_global_load_lds((const PPMS_GLOBAL void*)(uintptr_t)ps, (__attribute__((address_space(3))) void*)(dst + s * NT * 16), 16, 0, 0);
        }
    };

    // ---- the tiles of this wave: t = wave + NWC * s -> (x1 tile, x2 tile) = (t / S2, t % S2)
    int t1[TPW], t2[TPW], xo1[TPW], xo2[TPW];
    f32x16 acc[TPW];
#pragma unroll
    for (int s = 0; s < TPW; ++s) {
        const int t = wave + G::NWC * s;
        t1[s] = t / G::S2, t2[s] = t % G::S2;
        xo1[s] = lk * R1 + ((t1[s] * 32 + li) ^ ((R1 >= 64 && lk) ? 32 : 0));                   // (this lane reads channel c + lk)
        xo2[s] = CORR_CK * R1 + lk * R2 + ((t2[s] * 32 + li) ^ ((R2 >= 64 && lk) ? 32 : 0));
        acc[s] = (f32x16){0};
    }
    const bool computes = wave < G::NWC;
    const float inv = sqrtf((float)C);
    const int W1 = W >> 1, W2 = W1 >> 1, W3 = W2 >> 1, W4 = W3 >> 1;
    float* const sink = (float*)g_corr_sink + (((blockIdx.x + gridDim.x * blockIdx.y) & 1023) * 64 + lane) * 4;

    issue(0);
    issue(1);
    issue(2);
    int since = 4;                                            // stages begun since the last epilogue (>= 4: its stores are no longer in the way)
    for (int it = 0; it < total; ++it) {
        // this thread's pieces of stage `it` have landed: two younger stages may be in flight, and behind them the stores of an epilogue
        if (since <= 2 && computes) {
            if (TPW == 2)
                asm volatile("s_waitcnt vmcnt(40)" ::: "memory");
            else
                asm volatile("s_waitcnt vmcnt(22)" ::: "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        }
        static_assert(2 * PPT + CORR_S == 22 && 2 * PPT + 2 * CORR_S == 40 && 2 * PPT == 4, "vmcnt immediates");
        __builtin_amdgcn_s_barrier();                         // ... and everybody else's; stage it - 1 has been read by all waves
        issue(it + 3);                                        // into the ring slot of stage it - 1
        ++since;
        const int bj = it / nch;
        if (computes && !(CORR_ABL & 2)) {
            const float* st = (const float*)(smem + (it & (CORR_NST - 1)) * STAGE);
#pragma unroll
            for (int u = 0; u < CORR_CK / 2; ++u)
#pragma unroll
                for (int s = 0; s < TPW; ++s)
                    acc[s] = __builtin_amdgcn_mfma_f32_32x32x2f32(st[xo2[s] + 2 * u * R2], st[xo1[s] + 2 * u * R1], acc[s], 0, 0, 0);
        }
        if (it - bj * nch == nch - 1 && computes) {
            // D^T: lane (li, lk), register 4 q + j  <->  corr[x1 = tile1 * 32 + li][x2 = tile2 * 32 + 8 q + 4 lk + j]
#pragma unroll
            for (int s = 0; s < TPW; ++s) {
                const int x1 = x1b + t1[s] * 32 + li;
                const int64_t prow = (int64_t)row * W + x1;
                const bool rowok = x1 < W && !(CORR_ABL & 1);
                const int x2t = bj * R2 + t2[s] * 32;
                float d3[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const int x2 = x2t + 8 * q + 4 * lk;
                    f32x4 v0;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v0[j] = acc[s][4 * q + j] / inv;
                    const f32x2 v1 = {0.5f * (v0[0] + v0[1]), 0.5f * (v0[2] + v0[3])};
                    const float v2 = 0.5f * (v1[0] + v1[1]);
                    d3[q] = 0.5f * (v2 + __shfl_xor(v2, 32));            // (both lanes of the pair form the same sum)
                    const bool ok = rowok && x2 < W;
                    corr_st16(ok ? p0 + prow * W + x2 : sink, v0);
                    corr_st8(ok ? p1 + prow * W1 + (x2 >> 1) : sink, v1);
                    corr_st4(ok ? p2 + prow * W2 + (x2 >> 2) : sink, v2);
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)          // (both lanes of a pair hold the value and store it to the same place: no sink traffic)
                    corr_st4((rowok && (x2t >> 3) + q < W3) ? p3 + prow * W3 + (x2t >> 3) + q : sink, d3[q]);
#pragma unroll
                for (int m = 0; m < 2; ++m)
                    corr_st4((rowok && (x2t >> 4) + m < W4) ? p4 + prow * W4 + (x2t >> 4) + m : sink, 0.5f * (d3[2 * m] + d3[2 * m + 1]));
                acc[s] = (f32x16){0};
            }
            since = 0;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the padding stages requested past the end: nothing may land after the workgroup left
}

template <int R1LOG, int R2LOG>
static void corr_build_line_launch(const float* f1, const float* f2, float* const pyr[5], int B, int C, int H, int W, hipStream_t st) {
    using G = CorrGeo<R1LOG, R2LOG>;
    static ppms_device_once once;
    once.run([] { (void)hipFuncSetAttribute((const void*)corr_build_line_kernel<R1LOG, R2LOG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)G::LDS); });
    hipLaunchKernelGGL((corr_build_line_kernel<R1LOG, R2LOG>), dim3((W + G::R1 - 1) / G::R1, B * H), dim3(G::NT), G::LDS, st, f1, f2, pyr[0], pyr[1],
                       pyr[2], pyr[3], pyr[4], C, H, W, (W + G::R2 - 1) / G::R2);
}

extern "C" int ppms_corr_build(const float* fmap1, const float* fmap2, float* const pyr[5], int B, int C, int H, int W, void* stream) {
    PPMS_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "corr_build: bad shape B=%d C=%d H=%d W=%d", B, C, H, W);
    // the reference's 4 avg_pool2d([1,2]) calls need a width of at least 16 (corr.py:70-72; SURVEY hazard 2)
    PPMS_REQUIRE((W >> 4) >= 1, "corr_build: W=%d too small for a 4-level pyramid (needs W >= 16 at this scale)", W);
    PPMS_REQUIRE(fmap1 && fmap2 && pyr && pyr[0] && pyr[1] && pyr[2] && pyr[3] && pyr[4], "corr_build: null operand");
    const bool aligned = (((uintptr_t)fmap1 | (uintptr_t)fmap2 | (uintptr_t)pyr[0]) & 15) == 0 && ((uintptr_t)pyr[1] & 7) == 0;
    if (W % 4 == 0 && C % CORR_CK == 0 && aligned) {          // line-resident form: 16-byte row pieces, whole 16-channel stages
        if (W <= 32)
            corr_build_line_launch<5, 5>(fmap1, fmap2, pyr, B, C, H, W, (hipStream_t)stream);
        else if (W <= 64)
            corr_build_line_launch<6, 6>(fmap1, fmap2, pyr, B, C, H, W, (hipStream_t)stream);
        else
            corr_build_line_launch<7, 7>(fmap1, fmap2, pyr, B, C, H, W, (hipStream_t)stream);
        return ppms_check_launch("corr_build");
    }
    dim3 grid((W + 31) / 32, B * H);                         // any width / channel count / alignment
    hipLaunchKernelGGL(corr_build_kernel, grid, dim3(256), 0, (hipStream_t)stream, fmap1, fmap2, pyr[0], pyr[1], pyr[2], pyr[3],
                       pyr[4], C, H, W);
    return ppms_check_launch("corr_build");
}

// ------------------------------------------------------------------------------------------------
// lookup: 4 levels x 9 taps of linear interpolation at x + flow_x, zero padding, align_corners=True.
// ------------------------------------------------------------------------------------------------
// fast path: 4 pixels x 64 channel slots per block, channel-last SP output (36 real + 28 zero channels).
// One wave per pixel, lane = output channel, two loads per tap (neighbouring lanes of a level hit the same cache lines).  Round 3 tried
// north_star's literal prescription -- a 12-float window per (pixel, level) staged by 48 lanes, taps picked with wave shuffles, 4 pixels
// per wave for memory-level parallelism: 20.2 us against 16.4 us at the 1/4 scale of config 2 (8.9 / 7.4, 6.4 / 4.4 at 1/8, 1/16): the
// kernel is bound by its store stream (128 B per plane and pixel) and by wave count, not by the tap loads, so this form stays.
__global__ __launch_bounds__(256) void corr_lookup_sp_kernel(const float* __restrict__ l0, const float* __restrict__ l1,
                                                             const float* __restrict__ l2, const float* __restrict__ l3,
                                                             const float* __restrict__ flow, int flow_nhwc, bf16_t* __restrict__ ohi,
                                                             bf16_t* __restrict__ olo, int out_ld, bf16_t* __restrict__ fhi,
                                                             bf16_t* __restrict__ flo, int f_ld, int H, int W, int64_t P) {
    const int64_t p = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int ch = threadIdx.x & 63;
    if (p >= P) return;
    const int x = (int)(p % W);
    const int64_t hw = (int64_t)H * W;
    const int64_t frame = p / hw;
    const int64_t rem = p - frame * hw;
    float fx, fy;
    if (flow_nhwc) {
        fx = flow[p * 2];
        fy = flow[p * 2 + 1];
    } else {
        fx = flow[frame * 2 * hw + rem];
        fy = flow[(frame * 2 + 1) * hw + rem];
    }
    float v = 0.0f;
    if (ch < 36) {
        const int lvl = ch / 9, kk = ch - lvl * 9;
        const int Wl = W >> lvl;
        const float* L = (lvl == 0 ? l0 : lvl == 1 ? l1 : lvl == 2 ? l2 : l3) + p * Wl;
        v = lookup_tap(L, Wl, (float)x + fx, lvl, kk);
    }
    bf16_t h, l;
    split_bf16(v, h, l);
    ohi[p * out_ld + ch] = h;
    olo[p * out_ld + ch] = l;
    if (fhi != nullptr && (ch == 36 || ch == 37)) {
        split_bf16(ch == 36 ? fx : fy, h, l);
        fhi[p * f_ld + (ch - 36)] = h;
        flo[p * f_ld + (ch - 36)] = l;
    }
}

// API-compat path: fp32 (B,36,H,W) like CorrBlock1D.__call__
__global__ __launch_bounds__(256) void corr_lookup_nchw_kernel(const float* __restrict__ l0, const float* __restrict__ l1,
                                                               const float* __restrict__ l2, const float* __restrict__ l3,
                                                               const float* __restrict__ flow, int flow_nhwc, float* __restrict__ out, int H,
                                                               int W, int64_t P) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;      // (frame, ch, pixel-in-frame)
    const int64_t hw = (int64_t)H * W;
    if (idx >= P * 36) return;
    const int64_t frame = idx / (36 * hw);
    const int64_t r = idx - frame * 36 * hw;
    const int ch = (int)(r / hw);
    const int64_t rem = r - (int64_t)ch * hw;
    const int64_t p = frame * hw + rem;
    const int x = (int)(rem % W);
    const float fx = flow_nhwc ? flow[p * 2] : flow[frame * 2 * hw + rem];
    const int lvl = ch / 9, kk = ch - lvl * 9;
    const int Wl = W >> lvl;
    const float* L = (lvl == 0 ? l0 : lvl == 1 ? l1 : lvl == 2 ? l2 : l3) + p * Wl;
    out[idx] = lookup_tap(L, Wl, (float)x + fx, lvl, kk);
}

extern "C" int ppms_corr_lookup(const float* const pyr[4], const float* flow, int flow_nhwc, float* out_nchw, void* out_hi, void* out_lo,
                                int out_ld, void* flow_sp_hi, void* flow_sp_lo, int flow_sp_ld, int B, int H, int W, void* stream) {
    PPMS_REQUIRE(B > 0 && H > 0 && (W >> 3) >= 2, "corr_lookup: bad shape B=%d H=%d W=%d (level 3 needs >= 2 columns)", B, H, W);
    PPMS_REQUIRE(out_nchw != nullptr || out_hi != nullptr, "corr_lookup: no output given");
    const int64_t P = (int64_t)B * H * W;
    if (out_hi != nullptr) {
        PPMS_REQUIRE(out_lo != nullptr && out_ld >= 64, "corr_lookup: SP output needs lo plane and ld >= 64");
        hipLaunchKernelGGL(corr_lookup_sp_kernel, dim3(ceil_div(P, 4)), dim3(256), 0, (hipStream_t)stream, pyr[0], pyr[1], pyr[2], pyr[3],
                           flow, flow_nhwc, (bf16_t*)out_hi, (bf16_t*)out_lo, out_ld, (bf16_t*)flow_sp_hi, (bf16_t*)flow_sp_lo,
                           flow_sp_ld, H, W, P);
    }
    if (out_nchw != nullptr) {
        hipLaunchKernelGGL(corr_lookup_nchw_kernel, dim3(ceil_div(P * 36, 256)), dim3(256), 0, (hipStream_t)stream, pyr[0], pyr[1],
                           pyr[2], pyr[3], flow, flow_nhwc, out_nchw, H, W, P);
    }
    return ppms_check_launch("corr_lookup");
}
