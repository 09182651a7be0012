// Register-streamed convolution for SMALL maps (the 1/16 and 1/8 scales of the update blocks: 3 200 / 12 800 pixels at config 2), every
// convolution of SequenceUpdateBlock3D with taps (ppmtereo_update.py:254-312 GRU passes, :445-482 motion encoder, :670-678 flow head,
// :889-893 uncertainty head, :910-914 mask head) and the K = 768 Linear layers of update_block16's space attention (:619-631).
//
// Why another kernel.  On these maps a convolution is 25-400 tiles of conv_gemm2: its k-steps (one LDS round trip + barrier + exposed global
// prefetch each) form a latency chain, so the chip was filled by cutting K over several workgroups per tile -- fp32 partial slabs to HBM and a
// second launch (conv_slice_reduce_kernel) that sums them: 14 + 13 extra launches per iteration at the 1/16 and 1/8 scales, 5-13 us each,
// 5.6 % of all GPU time, with the convolution launch itself at 0.10-0.22 of the split bound.  Here the gemm1.hip structure ("one memory
// round trip deep") is extended to taps and long K:
//   * workgroup = 4 waves = 4 K-groups of ONE (32 PB)-pixel x (32 CB)-cout tile; pixels are consecutive in the flattened (t, y, x) order;
//   * a k16-step is (tap, 16-channel chunk); within every tap wave w takes the chunks w, w + 4, ...: the four waves of a workgroup read the
//     four 32-byte quarters of the same 128-byte activation lines at about the same time;
//   * both operands go STRAIGHT to registers in MFMA-fragment order through a ring of D steps (D x (2 PB + 2 CB) 16-byte requests per lane in
//     flight): activations are channel-last, so lane (r, h) of a step reads X[pixel r + tap offset][16 chunk + 8 h ..] of each plane (lanes
//     whose tap falls outside the volume read a zero page -- no branch around a load); weights are packed per (32-cout block, tap, chunk,
//     plane) as the lane image (ppmstereo_amd/packing.py pack_stream);
//   * temporal taps that leave the readable frames for EVERY pixel of the tile are skipped (T = 5: 24 % of the (5,1,1) pass);
//   * 3 PB CB MFMAs per step (bf16x3 split, lo products first), no LDS and no barrier in the K loop; then the four partial tiles are summed
//     through LDS in wave order (deterministic) and every thread finishes 8 couts of a pixel with the shared row epilogue.
// No slices, no workspace, no reduce launch; the chip is filled by tiles x cout blocks (3 200 pixels x 256 couts = 400 workgroups).
#include "common.h"
#include "conv_epilogue.h"
#include <type_traits>

namespace {

__device__ __attribute__((aligned(256))) unsigned int g_zero_page7[64];     // zero-initialised: what out-of-volume taps read

struct Geo7 {
    int64_t P;               // pixels = T*H*W
    int nk16;                // 16-channel chunks per tap (all segments)
    int npw;                 // chunks per wave and tap = nk16 / KG
    int n0;                  // chunks of segment 0
    int nsteps;              // k16-steps of the whole convolution = taps * nk16 (stride of a 32-cout block in the pack, in 2 KiB units)
    int nmb;                 // workgroups along M = M / (32 CB)
};

template <int CB, int PB, int KG>
struct G7 {
    static constexpr int LD = 32 * CB + 4;                  // floats per staged pixel row (+4: conflict-free b128 phases)
    static constexpr int RED = KG * 32 * PB * LD * 4;       // bytes: KG partial tiles
};

// KG = waves (K-groups) per workgroup: 4, or 8 where the tiles alone leave CUs with a single 4-wave workgroup -- what a CU takes in from L2
// is set by the requests it has in flight (waves x D x (2 PB + 2 CB)), not by the MFMA rate
template <int CB, int PB, int D, int KG>
__global__ __launch_bounds__(64 * KG) void conv_stream_kernel(const ppms_conv pv, const Geo7 g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const ppms_conv& p = pv;
    const int tid = threadIdx.x, lane = tid & 63;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);          // K-group of this wave (uniform: keeps the chunk arithmetic scalar)
    const int r = lane & 31, h = lane >> 5;
    const int tile = (int)blockIdx.x / g.nmb, mgrp = (int)blockIdx.x - tile * g.nmb;
    const int64_t px0 = (int64_t)tile * (32 * PB);
    const int cout0 = mgrp * 32 * CB;                                  // first cout of this workgroup (of the whole M)
    const int H = p.H, W = p.W, T = p.T, HW = H * W;
    const int kh = p.kh, kw = p.kw;
    const int ht = p.kt >> 1, hy = kh >> 1, hx = kw >> 1;

    // ---- this lane's PB pixels (rows past the end are computed on the last pixel and dropped) ----------------------------------------------
    int pix[PB], pt[PB], py[PB], pxx[PB];
#pragma unroll
    for (int pb = 0; pb < PB; ++pb) {
        const int64_t q = px0 + pb * 32 + r;
        pix[pb] = (int)(q < g.P ? q : g.P - 1);
        pt[pb] = pix[pb] / HW;
        const int rem = pix[pb] - pt[pb] * HW;
        py[pb] = rem / W;
        pxx[pb] = rem - py[pb] * W;
    }
    // temporal taps readable by at least one pixel of the tile: one contiguous range of kz (the tile's frames are [tmin, tmax])
    const int64_t plast = px0 + 32 * PB - 1 < g.P ? px0 + 32 * PB - 1 : g.P - 1;
    const int tmin = (int)(px0 / HW), tmax = (int)(plast / HW);
    const int kz0 = (ht - tmax - p.t_halo) > 0 ? (ht - tmax - p.t_halo) : 0;
    const int kz1 = (ht + T + p.t_halo - 1 - tmin) < (p.kt - 1) ? (ht + T + p.t_halo - 1 - tmin) : (p.kt - 1);
    const int NSW = (kz1 - kz0 + 1) * kh * kw * g.npw;               // k16-steps of this wave

    // ---- producer cursor: the step whose operands are requested next ---------------------------------------------------------------------
    int kz = kz0, ky = 0, kx = 0, jj = 0, ip = 0;
    int tap = kz0 * kh * kw;
    int boff[PB];
    bool ok[PB];
    auto set_tap = [&]() {
        const int dt = kz - ht, dy = ky - hy, dx = kx - hx;
        const int shift = (dt * H + dy) * W + dx;
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
            ok[pb] = (unsigned)(pt[pb] + dt + p.t_halo) < (unsigned)(T + 2 * p.t_halo) && (unsigned)(py[pb] + dy) < (unsigned)H &&
                     (unsigned)(pxx[pb] + dx) < (unsigned)W;
            boff[pb] = pix[pb] + shift;
        }
    };
    set_tap();
    const bf16_t* s0h = (const bf16_t*)p.seg[0].hi;
    const bf16_t* s0l = (const bf16_t*)p.seg[0].lo;
    const bf16_t* s1h = (const bf16_t*)p.seg[p.nseg > 1 ? 1 : 0].hi;
    const bf16_t* s1l = (const bf16_t*)p.seg[p.nseg > 1 ? 1 : 0].lo;
    const int ld0 = p.seg[0].ld, ld1 = p.seg[p.nseg > 1 ? 1 : 0].ld;
    const bf16_t* zpage = (const bf16_t*)g_zero_page7;
    const char* wlane = (const char*)p.w + (size_t)lane * 16 + (size_t)mgrp * CB * g.nsteps * 2048;
    const size_t wblk = (size_t)g.nsteps * 2048;                      // bytes between consecutive 32-cout blocks

    bf16x8 bh[D][PB], bl[D][PB], ah[D][CB], al[D][CB];
    auto issue = [&](auto dtag) {
        constexpr int d = decltype(dtag)::value;
        const bool live = ip < NSW;                                   // (uniform) past the last step: every operand from the zero page
        const int chunk = KG * jj + w;
        const bool s1 = chunk >= g.n0;
        const int cc = (chunk - (s1 ? g.n0 : 0)) * 16 + 8 * h;
        const bf16_t* sh = s1 ? s1h : s0h;
        const bf16_t* sl = s1 ? s1l : s0l;
        const int ld = s1 ? ld1 : ld0;
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) {
            const int64_t off = (int64_t)boff[pb] * ld + cc;
            bh[d][pb] = gld<bf16x8>(ok[pb] && live ? sh + off : zpage);
            bl[d][pb] = gld<bf16x8>(ok[pb] && live ? sl + off : zpage);
        }
        const char* wp = live ? wlane + (size_t)(tap * g.nk16 + chunk) * 2048 : (const char*)zpage;
        const size_t wb = live ? wblk : 0;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            ah[d][cb] = gld<bf16x8>(wp + cb * wb);
            al[d][cb] = gld<bf16x8>(wp + cb * wb + (live ? 1024 : 0));
        }
        ++ip;
        if (++jj == g.npw) {                                          // next tap (uniform)
            jj = 0;
            ++tap;
            if (++kx == kw) {
                kx = 0;
                if (++ky == kh) {
                    ky = 0;
                    ++kz;
                }
            }
            set_tap();
        }
    };

    f32x16 acc[CB][PB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb)
#pragma unroll
        for (int pb = 0; pb < PB; ++pb) acc[cb][pb] = (f32x16){0};
    auto consume = [&](auto dtag) {
        constexpr int d = decltype(dtag)::value;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int pb = 0; pb < PB; ++pb) {
                acc[cb][pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[d][cb], bh[d][pb], acc[cb][pb], 0, 0, 0);
                acc[cb][pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[d][cb], bl[d][pb], acc[cb][pb], 0, 0, 0);
                acc[cb][pb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[d][cb], bh[d][pb], acc[cb][pb], 0, 0, 0);
            }
    };
    // the ring: D steps requested ahead, slot d of the ring = steps d, d + D, ... (compile-time register indices).  The loop has NO branch
    // around a load or an MFMA: the compiler's s_waitcnt insertion then counts exactly (vmcnt(N) with the younger D - 1 steps left in
    // flight); with conditional steps it merged the paths into one vmcnt(0) per D steps.  The step count is therefore rounded up to a
    // multiple of D: steps past the last one multiply zero-page operands (at most D - 1 of them), and the last D requests hit the zero page.
#define CS_FOR_SLOTS(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11)
#define CS_FILL(dv) \
    if (dv < D) issue(std::integral_constant<int, (dv < D ? dv : 0)>{});
    CS_FOR_SLOTS(CS_FILL)
#undef CS_FILL
    for (int i0 = 0; i0 < NSW; i0 += D) {
#define CS_STEP(dv)                                                 \
    if (dv < D) {                                                   \
        consume(std::integral_constant<int, (dv < D ? dv : 0)>{}); \
        issue(std::integral_constant<int, (dv < D ? dv : 0)>{});   \
    }
        CS_FOR_SLOTS(CS_STEP)
#undef CS_STEP
    }
#undef CS_FOR_SLOTS

    // ---- the four K-groups' partial tiles -> LDS [wave][pixel][cout]; summed in wave order by the finishing threads -------------------------
    constexpr int LD = G7<CB, PB, KG>::LD;
    float* red = (float*)smem;
    {
        float* mine = red + (size_t)w * 32 * PB * LD;
#pragma unroll
        for (int pb = 0; pb < PB; ++pb)
#pragma unroll
            for (int cb = 0; cb < CB; ++cb)
#pragma unroll
                for (int gq = 0; gq < 4; ++gq) {
                    const f32x4 v4 = {acc[cb][pb][4 * gq], acc[cb][pb][4 * gq + 1], acc[cb][pb][4 * gq + 2], acc[cb][pb][4 * gq + 3]};
                    *(f32x4*)(mine + (pb * 32 + r) * LD + cb * 32 + 8 * gq + 4 * h) = v4;
                }
    }
    __syncthreads();
    const int half = (cout0 >= p.m_split) ? 1 : 0;            // (a workgroup's couts lie in ONE epilogue half: checked on the host)
    const ppms_epilogue e = p.epi[half];
    const int cbase = cout0 - (half ? p.m_split : 0);

    // tasks = (pixel row, 8-cout group) of the tile: 32 PB x 4 CB of them over the workgroup's threads
    constexpr int NTASK = 32 * PB * 4 * CB, NTHR = 64 * KG;
    constexpr int PER = (NTASK + NTHR - 1) / NTHR;
    auto rows = [&](auto cls_tag) {
        constexpr int CLS = decltype(cls_tag)::value;
        row8_aux aux[PER];
        int64_t pixg[PER];
        bool okg[PER];
        int qg[PER], pxg[PER];
#pragma unroll
        for (int i = 0; i < PER; ++i) {                                // operands of every row of this thread first, then the stores
            const int task = i * NTHR + tid;
            pxg[i] = task / (4 * CB);
            qg[i] = task - pxg[i] * (4 * CB);
            pixg[i] = px0 + pxg[i];
            okg[i] = task < NTASK && pixg[i] < g.P;
            if (okg[i]) row8_fetch<CLS>(e, pixg[i], cbase + qg[i] * 8, aux[i]);
        }
#pragma unroll
        for (int i = 0; i < PER; ++i) {
            if (!okg[i]) continue;
            const float* src = red + pxg[i] * LD + qg[i] * 8;
            f32x4 s0 = *(const f32x4*)src, s1 = *(const f32x4*)(src + 4);
#pragma unroll
            for (int k = 1; k < KG; ++k) {
                s0 += *(const f32x4*)(src + (size_t)k * 32 * PB * LD);
                s1 += *(const f32x4*)(src + (size_t)k * 32 * PB * LD + 4);
            }
            const f32x4 b0 = gld<f32x4>(p.bias + cout0 + qg[i] * 8), b1 = gld<f32x4>(p.bias + cout0 + qg[i] * 8 + 4);
            float v[8];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                v[j] = s0[j] + b0[j];
                v[4 + j] = s1[j] + b1[j];
            }
            row8_finish<CLS>(e, v, pixg[i], cbase + qg[i] * 8, HW, aux[i]);
        }
    };
    using I0 = std::integral_constant<int, EPI_CLS_PLAIN>;
    using I1 = std::integral_constant<int, EPI_CLS_PRE>;
    using I2 = std::integral_constant<int, EPI_CLS_AUX>;
    using I3 = std::integral_constant<int, EPI_CLS_GRU>;
    using I4 = std::integral_constant<int, EPI_CLS_ANY>;
    using I5 = std::integral_constant<int, EPI_CLS_AUXPRE>;
    const int cls = epilogue_class(e);
    if (cls == EPI_CLS_PLAIN) rows(I0{});
    else if (cls == EPI_CLS_PRE) rows(I1{});
    else if (cls == EPI_CLS_AUX) rows(I2{});
    else if (cls == EPI_CLS_GRU) rows(I3{});
    else if (cls == EPI_CLS_AUXPRE) rows(I5{});
    else rows(I4{});
}

struct Plan7 {
    int pb, depth, kg;
    Geo7 g;
};

bool plan7(const ppms_conv* d, Plan7& pl, int hint) {
    if (d == nullptr || d->nseg < 1 || d->nseg > 2 || d->w == nullptr || d->bias == nullptr || d->groups > 1) return false;      // (grouped: conv_gemm6 only)
    if (d->kt < 1 || d->kh < 1 || d->kw < 1 || !(d->kt & 1) || !(d->kh & 1) || !(d->kw & 1)) return false;
    if (d->T <= 0 || d->H <= 0 || d->W <= 0 || d->t_halo < 0) return false;
    if (d->M <= 0 || d->M % 64 != 0) return false;
    int K = 0;
    for (int s = 0; s < d->nseg; ++s) {
        if (d->seg[s].hi == nullptr || d->seg[s].lo == nullptr || d->seg[s].c <= 0 || d->seg[s].c % 16 != 0 || d->seg[s].ld % 8 != 0) return false;
        if (((uintptr_t)d->seg[s].hi & 15) || ((uintptr_t)d->seg[s].lo & 15)) return false;
        K += d->seg[s].c;
    }
    if (K % 64 != 0) return false;                       // every tap's chunks are dealt to the four waves in equal shares
    const int64_t P = (int64_t)d->T * d->H * d->W;
    if (P >= (1ll << 31) / 8) return false;              // (pixel offsets incl. temporal halos stay in 32 bits)
    const int64_t taps = (int64_t)d->kt * d->kh * d->kw;
    if (taps * (K / 16) >= (1 << 20)) return false;
    const bool two = d->m_split < d->M;
    if (two && d->m_split % 64 != 0) return false;
    for (int hlf = 0; hlf < 2; ++hlf) {
        const ppms_epilogue& e = d->epi[hlf];
        if (hlf == 1 && !two) break;
        if (e.n_valid <= 0) return false;
        if (epilogue_row8_check(e) != nullptr) return false;
        if (e.kind == PPMS_EPI_ADDF32 || e.out_vt != nullptr) return false;
    }
    Geo7& g = pl.g;
    g.P = P;
    g.nk16 = K / 16;
    g.n0 = d->seg[0].c / 16;
    g.nsteps = (int)taps * g.nk16;
    g.nmb = d->M / 64;
    if (hint > 2) {                                      // probes: (KG << 8) | (PB << 4) | D
        pl.kg = hint >> 8, pl.pb = (hint >> 4) & 15, pl.depth = hint & 15;
        if ((pl.kg != 4 && pl.kg != 8) || (pl.pb != 1 && pl.pb != 2) || g.nk16 % pl.kg) return false;
    } else {
        // tile: 32 pixels x 64 couts per workgroup fills the chip on the smallest maps (3 200 pixels x 128 couts = 200 workgroups); 64-pixel tiles
        // halve the weight bytes per MFMA once the couts alone give enough workgroups (measured, tools/conv_stream_probe.py: 3 200 pixels -- M >= 192:
        // 64-pixel tiles 27-37 us against 32-42; M = 128: 32-pixel tiles 18-22 against 25-31; 12 800 pixels: 64-pixel tiles throughout)
        pl.pb = (hint == 1 || hint == 2) ? hint : ((P > 4096 || d->M >= 192) ? 2 : 1);
        pl.depth = pl.pb == 2 ? 3 : 6;           // (sweep: deeper rings and 8-wave workgroups change nothing or lose -- the CU's L1 path is the bound)
        pl.kg = 4;
    }
    g.npw = g.nk16 / pl.kg;
    return true;
}

template <int CB, int PB, int D, int KG>
int launch7(const ppms_conv* d, const Plan7& pl, hipStream_t st) {
    const size_t lds = (size_t)G7<CB, PB, KG>::RED;
    static ppms_device_once once;
    once.run([] { (void)hipFuncSetAttribute((const void*)conv_stream_kernel<CB, PB, D, KG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    const int64_t tiles = (pl.g.P + 32 * PB - 1) / (32 * PB);
    hipLaunchKernelGGL((conv_stream_kernel<CB, PB, D, KG>), dim3((unsigned)(tiles * pl.g.nmb)), dim3(64 * KG), lds, st, *d, pl.g);
    return ppms_check_launch("conv_stream");
}

}  // namespace

// 0: not served.  1: served AND measured faster than the LDS-staged kernel with its K slices + reduce launch (tools/conv_stream_probe.py,
// profiles/r04_conv_stream_probe.txt).  2: served, but the LDS-staged kernels win.  Why there is a crossover: every operand byte of this
// kernel passes the CU's vector L1 (1 KiB per MFMA with 32-pixel tiles, 0.67 with 64; an activation request touches 32 cache lines for
// 1 KiB), which moves ~64 B per clock for the whole CU: the K loop runs at ~30 % of the MFMA rate whatever the ring depth or the wave count
// (sweep in the probe log), so its time grows with pixels x K x M, while the staged kernels re-use one LDS window over the spatial taps.
// Measured: maps of <= 4 096 pixels (the 1/16 scale at 320x512) unless K x M is large (the 15-tap z/r conv of update_block16: 59 us staged,
// 69 streamed); larger maps only where the staged kernel has no window to re-use or too few tiles: convolutions without spatial taps (the
// temporal (kt, 1, 1) GRU pass: 61 against 83 us and 31 against 49 at 12 800 pixels; the K = 768 Linear layers of the space attention: 131
// against 156 and 74 against 101 at 18 400) up to 32 768 pixels, and 64-cout convs (16 against 24 us) up to 16 384.
extern "C" int ppms_conv_stream_applicable(const ppms_conv* d) {
    Plan7 pl;
    if (!plan7(d, pl, 0)) return 0;
    const int64_t K = (int64_t)pl.g.nsteps * 16;
    const bool spatial = d->kh > 1 || d->kw > 1;
    if (pl.g.P <= 4096) return (double)pl.g.P * (double)K * d->M <= 4.0e9 ? 1 : 2;       // (3 200 pixels: K x M <= 1.25 M)
    if (!spatial && pl.g.P <= 32768 && (d->kt > 1 || K >= 768)) return 1;
    if (d->M <= 64 && pl.g.P <= 16384) return 1;
    return 2;
}

// hint: 0 = the library chooses; 1 / 2 = 32- / 64-pixel tiles
extern "C" int ppms_conv_stream(const ppms_conv* d, const ppms_conv* dev_desc, int hint, void* stream) {
    (void)dev_desc;
    Plan7 pl;
#ifdef PPMS_STREAM_PROBE
    PPMS_REQUIRE(hint >= 0, "conv_stream: hint must be 0 (choose), 1 or 2 (32-pixel blocks per tile), or a probe encoding (KG << 8) | (PB << 4) | D");
#else
    PPMS_REQUIRE(hint >= 0 && hint <= 2, "conv_stream: hint must be 0 (choose), 1 or 2 (32-pixel blocks per tile)");
#endif
    PPMS_REQUIRE(plan7(d, pl, hint), "conv_stream: not a convolution this kernel serves (odd taps, input channels a multiple of 64 in 16-channel-aligned "
                                     "segments, M %% 64 == 0, pack_stream weights, aligned SP operands, no out_vt / ADDF32 epilogue; ppms_conv_stream_applicable tells)");
    hipStream_t st = (hipStream_t)stream;
#define S7_CASE(PBV, DV, KGV) \
    if (pl.pb == PBV && pl.depth == DV && pl.kg == KGV) return launch7<2, PBV, DV, KGV>(d, pl, st);
    S7_CASE(1, 6, 4) S7_CASE(2, 3, 4)
#ifdef PPMS_STREAM_PROBE        // (tools/conv_stream_probe.py builds: ring depth / K-group sweeps)
    S7_CASE(1, 4, 4) S7_CASE(1, 8, 4) S7_CASE(1, 10, 4) S7_CASE(2, 6, 4) S7_CASE(1, 4, 8) S7_CASE(1, 6, 8) S7_CASE(1, 8, 8) S7_CASE(2, 4, 8) S7_CASE(2, 6, 8)
    S7_CASE(2, 4, 4) S7_CASE(2, 3, 8)
#endif
#undef S7_CASE
    ppms_set_error("conv_stream: no instantiation for PB=%d D=%d KG=%d", pl.pb, pl.depth, pl.kg);
    return PPMS_EINVAL;
}
