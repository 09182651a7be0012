// Thin GEMM for the 1x1 convolutions / Linear layers of the path (to_v, the flow head's second layer, mask head tail, q/k projection,
// update_block16's time / space attention layers: ppmtereo_update.py:129, 646, 674, 914, 593-631): out[p][m] = sum_k W[m][k] X[p][k].
//
// Why not the implicit-GEMM kernel: with K = 128 ... 768 it runs 4 ... 24 k-steps, each one LDS round trip + barrier + an exposed global
// prefetch (~1 us per step for a workgroup of two waves), under a 2 us cold start and in front of a 2.6 us epilogue -- 19-27 us for
// GEMMs whose MFMA work is < 1 us, plus a slice-reduce launch at the small scales.  These launches are LATENCY, not throughput.
// Here a launch is ONE memory round trip deep:
//   * workgroup = 4 waves = 4 K-groups of one 32-pixel x (32 CB)-cout tile: wave w owns the k16-steps [w NS, (w + 1) NS), K = 64 NS;
//   * both operands go STRAIGHT to registers in MFMA-fragment order, all requests of a wave issued before the first is consumed:
//     activations are channel-last, so lane (r, h) of step s reads the 16 bytes X[pixel r][16 s + 8 h ..] of each plane; weights are
//     packed per (32-cout block, k16-step, plane) as the lane image (ppmstereo_amd/packing.py pack_gemm1);
//   * 3 NS CB MFMAs per wave (bf16x3 split), then the four partial tiles are summed through LDS in wave order (deterministic) and
//     every thread finishes 8 couts of one pixel with the shared row epilogue (conv_epilogue.h: bias, activation, residual, hoisted
//     share, SP / fp32 outputs); the attention's transposed bf16 V goes through an LDS patch and leaves as 16-byte row pieces.
// No operand LDS, no K loop barrier, no slices, no reduce launch.
#include "common.h"
#include "conv_epilogue.h"
#include <type_traits>

namespace {

template <int CB>
struct G1 {
    static constexpr int LD = 32 * CB + 4;                  // floats per staged pixel row (+4: conflict-free b128 phases)
    static constexpr int RED = 4 * 32 * LD * 4;             // bytes: four partial tiles
    static constexpr int VTP = 32 * CB * 32 * 2;            // bytes: transposed bf16 patch [cout][32 px]
};

template <int CB, int NS>
__global__ __launch_bounds__(256) void gemm1_kernel(const ppms_conv pv, const int64_t P, const int nk) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const ppms_conv& p = pv;
    const int tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const int r = lane & 31, h = lane >> 5;
    const int64_t px0 = (int64_t)blockIdx.x * 32;
    const int cout0 = (int)blockIdx.y * 32 * CB;             // first cout of this workgroup (of the whole M)
    const int HW = p.H * p.W;

    // ---- operands of this wave's K slice: every request goes out before anything is consumed ---------------------------------------
    bf16x8 bh[NS], bl[NS], ah[CB][NS], al[CB][NS];
    {
        const int64_t pix = (px0 + r < P) ? px0 + r : P - 1;             // (rows past the end are computed and dropped)
        const int c0 = p.seg[0].c;
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int k = 16 * (w * NS + s);
            const int sg = (k >= c0) ? 1 : 0;
            const int kk = k - (sg ? c0 : 0) + 8 * h;
            const bf16_t* xh = (const bf16_t*)p.seg[sg].hi + pix * p.seg[sg].ld + kk;
            const bf16_t* xl = (const bf16_t*)p.seg[sg].lo + pix * p.seg[sg].ld + kk;
            bh[s] = gld<bf16x8>(xh);
            bl[s] = gld<bf16x8>(xl);
        }
        const char* wb = (const char*)p.w + (size_t)lane * 16;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            const int blk = (int)blockIdx.y * CB + cb;
#pragma unroll
            for (int s = 0; s < NS; ++s) {
                const size_t off = (((size_t)blk * nk + (w * NS + s)) * 2) * 1024;
                ah[cb][s] = gld<bf16x8>(wb + off);
                al[cb][s] = gld<bf16x8>(wb + off + 1024);
            }
        }
    }
    f32x16 acc[CB];
#pragma unroll
    for (int cb = 0; cb < CB; ++cb) acc[cb] = (f32x16){0};
#pragma unroll
    for (int s = 0; s < NS; ++s)
#pragma unroll
        for (int cb = 0; cb < CB; ++cb) {
            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al[cb][s], bh[s], acc[cb], 0, 0, 0);
            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cb][s], bl[s], acc[cb], 0, 0, 0);
            acc[cb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah[cb][s], bh[s], acc[cb], 0, 0, 0);
        }

    // ---- the four K-groups' partial tiles -> LDS [wave][pixel][cout]; summed in wave order by the finishing threads ---------------
    constexpr int LD = G1<CB>::LD;
    float* red = (float*)smem;
    {
        float* mine = red + (size_t)w * 32 * LD + r * LD;
#pragma unroll
        for (int cb = 0; cb < CB; ++cb)
#pragma unroll
            for (int gq = 0; gq < 4; ++gq) {
                const f32x4 v4 = {acc[cb][4 * gq], acc[cb][4 * gq + 1], acc[cb][4 * gq + 2], acc[cb][4 * gq + 3]};
                *(f32x4*)(mine + cb * 32 + 8 * gq + 4 * h) = v4;
            }
    }
    __syncthreads();
    const int half = (cout0 >= p.m_split) ? 1 : 0;            // (a workgroup's couts lie in ONE epilogue half: checked on the host)
    const ppms_epilogue e = p.epi[half];
    const int cbase = cout0 - (half ? p.m_split : 0);
    bf16_t* vtp = (bf16_t*)(smem + G1<CB>::RED);              // [cout 32 CB][32 px] bf16, only with out_vt
    const bool vt = e.out_vt != nullptr;

    auto rows = [&](auto cls_tag) {
        constexpr int CLS = decltype(cls_tag)::value;
#pragma unroll
        for (int t0 = 0; t0 < 128 * CB; t0 += 256) {
            const int task = t0 + tid;                        // (pixel row, 8-cout group) of the tile
            if (128 * CB < 256 && task >= 128 * CB) break;
            const int px = task / (4 * CB), q = task - px * (4 * CB);
            const int64_t pix = px0 + px;
            float v[8];
            {
                const float* src = red + px * LD + q * 8;
                f32x4 s0 = *(const f32x4*)src, s1 = *(const f32x4*)(src + 4);
#pragma unroll
                for (int k = 1; k < 4; ++k) {
                    s0 += *(const f32x4*)(src + (size_t)k * 32 * LD);
                    s1 += *(const f32x4*)(src + (size_t)k * 32 * LD + 4);
                }
                const f32x4 b0 = gld<f32x4>(p.bias + cout0 + q * 8), b1 = gld<f32x4>(p.bias + cout0 + q * 8 + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    v[j] = s0[j] + b0[j];
                    v[4 + j] = s1[j] + b1[j];
                }
            }
            if (pix < P) {
                row8_aux aux;
                row8_fetch<CLS>(e, pix, cbase + q * 8, aux);
                row8_finish<CLS>(e, v, pix, cbase + q * 8, HW, aux);
            }
            if (CLS == EPI_CLS_ANY && vt) {                   // STORE epilogue (checked on the host): the same values, bf16, transposed
                float y[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) y[j] = v[j];
                apply_act_n<8>(y, e.act, e.scale);
#pragma unroll
                for (int j = 0; j < 8; ++j) vtp[(q * 8 + j) * 32 + px] = vt_enc(y[j], e.vt_f16);
            }
        }
    };
    using I0 = std::integral_constant<int, EPI_CLS_PLAIN>;
    using I4 = std::integral_constant<int, EPI_CLS_ANY>;
    if (epilogue_class(e) == EPI_CLS_PLAIN) rows(I0{});
    else rows(I4{});

    if (vt) {                                                 // (uniform) transposed V: [frame][n_valid][H*W] bf16
        __syncthreads();
        const bool rowwise = (HW & 31) == 0;                  // the tile is 32 consecutive pixels of ONE frame, 64-byte aligned
        if (rowwise) {
            const int64_t frame = px0 / HW, rem = px0 - frame * HW;
            for (int t = tid; t < 32 * CB * 4; t += 256) {
                const int c = t >> 2, ch = t & 3;
                const int cl = cbase + c;
                if (cl < e.n_valid && px0 + 8 * ch < P)       // (P is a multiple of H*W: whole rows of 8 exist)
                    gst<u32x4>((bf16_t*)e.out_vt + ((int64_t)frame * e.n_valid + cl) * HW + rem + 8 * ch, *(const u32x4*)(vtp + c * 32 + 8 * ch));
            }
        } else {
            for (int t = tid; t < 32 * CB * 32; t += 256) {
                const int c = t >> 5, px = t & 31;
                const int cl = cbase + c;
                const int64_t pix = px0 + px;
                if (cl < e.n_valid && pix < P) {
                    const int64_t frame = pix / HW, rem = pix - frame * HW;
                    gst<bf16_t>((bf16_t*)e.out_vt + ((int64_t)frame * e.n_valid + cl) * HW + rem, vtp[c * 32 + px]);
                }
            }
        }
    }
}

struct Plan1 {
    int cb, ns, nk;
};

// K = 64 NS with NS in {2, 3, 4, 6, 8}; CB (32-cout blocks per workgroup): 2 when that divides the cout blocks and keeps a workgroup inside
// one epilogue half, else 1 (measured at config 2's sizes, tools/gemm1_probe.py: 2 is best or tied everywhere, 4 gains nothing).
bool plan1(const ppms_conv* d, Plan1& pl, int cb_hint = 0) {
    if (d == nullptr || d->kt != 1 || d->kh != 1 || d->kw != 1 || d->nseg < 1 || d->nseg > 2 || d->groups > 1) return false;      // (grouped: conv_gemm6 only)
    if (d->M <= 0 || d->M % 32 != 0 || d->w == nullptr || d->bias == nullptr) return false;
    int K = 0;
    for (int s = 0; s < d->nseg; ++s) {
        if (d->seg[s].hi == nullptr || d->seg[s].lo == nullptr || d->seg[s].c <= 0 || d->seg[s].c % 16 != 0 || d->seg[s].ld % 8 != 0) return false;
        if (((uintptr_t)d->seg[s].hi & 15) || ((uintptr_t)d->seg[s].lo & 15)) return false;
        K += d->seg[s].c;
    }
    if (K % 64 != 0) return false;
    const int ns = K / 64;
    // (K = 768, i.e. 12 steps per wave, was built and measured: 48 operand requests per lane cost the occupancy that hides them -- the
    // 768 -> 768 Linear of update_block16 took 48 us against 45 us for the K-sliced implicit GEMM + reduce -- so such layers stay there)
    if (ns != 2 && ns != 3 && ns != 4 && ns != 6 && ns != 8) return false;
    const bool two = d->m_split < d->M;
    if (two && d->m_split % 32 != 0) return false;
    const int64_t P = (int64_t)d->T * d->H * d->W;
    if (P <= 0 || P >= (1ll << 31)) return false;
    const int64_t tiles = (P + 31) / 32;
    const int mblocks = d->M / 32;
    (void)tiles;
    int best = 0;
    for (int cb = (cb_hint > 0 ? cb_hint : 2); cb >= 1; cb >>= 1) {
        if (mblocks % cb || cb * ns > 16) continue;
        if (two && d->m_split % (32 * cb)) continue;
        best = cb;
        break;
    }
    if (best == 0) return false;
    for (int hlf = 0; hlf < 2; ++hlf) {
        const ppms_epilogue& e = d->epi[hlf];
        if (hlf == 1 && !two) break;
        if (e.n_valid <= 0) return false;
        if (epilogue_row8_check(e) != nullptr) return false;
        if (e.kind == PPMS_EPI_ADDF32) return false;
    }
    pl.cb = best;
    pl.ns = ns;
    pl.nk = K / 16;
    return true;
}

template <int CB, int NS>
int launch1(const ppms_conv* d, const Plan1& pl, hipStream_t st) {
    const int64_t P = (int64_t)d->T * d->H * d->W;
    const size_t lds = (size_t)G1<CB>::RED + G1<CB>::VTP;
    static ppms_device_once once;
    once.run([] { (void)hipFuncSetAttribute((const void*)gemm1_kernel<CB, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); });
    hipLaunchKernelGGL((gemm1_kernel<CB, NS>), dim3((unsigned)((P + 31) / 32), (unsigned)(d->M / (32 * CB))), dim3(256), lds, st, *d, P, pl.nk);
    return ppms_check_launch("gemm1");
}

}  // namespace

// 1: this kernel serves the convolution AND is the faster choice.  Measured (tools/gemm1_probe.py, profiles/r03_gemm1_probe.txt): on maps of
// <= 16 384 pixels (the 1/8 and 1/16 scales of every BASELINE configuration at 320x512, the 1/16 scale at 736x1280) it halves the launch
// (to_v 16 -> 8 us, 15 -> 7 us; 384 -> 384 Linear 17.5 -> 10.7 us; 384 -> 1152 38.8 -> 24.1 us incl. the slice reduce it makes unnecessary);
// on the 51 200-pixel map it ties with the implicit GEMM (to_v 25.5 / 25.6 us, convf1 19.7 / 19.2) except for narrow outputs (the flow head's
// 256 -> 54: 21 against 37 us) and loses on wide ones (256 -> 144: 46 against 42 us).  2: it serves it but the implicit GEMM is as fast.
extern "C" int ppms_gemm1_applicable(const ppms_conv* d) {
    Plan1 pl;
    if (!plan1(d, pl)) return 0;
    const int64_t P = (int64_t)d->T * d->H * d->W;
    return (P <= 16384 || d->M <= 64) ? 1 : 2;
}

extern "C" int ppms_gemm1(const ppms_conv* d, const ppms_conv* dev_desc, int cb_hint, void* stream) {
    (void)dev_desc;
    Plan1 pl;
    PPMS_REQUIRE(cb_hint == 0 || cb_hint == 1 || cb_hint == 2 || cb_hint == 4, "gemm1: cb_hint must be 0 (choose), 1, 2 or 4");
    PPMS_REQUIRE(plan1(d, pl, cb_hint), "gemm1: not a 1x1 convolution this kernel serves (K = 128 / 192 / 256 / 384 / 512 in 16-channel-aligned segments, M %% 32 == 0, "
                               "pack_gemm1 weights, aligned SP operands; ppms_gemm1_applicable tells)");
    hipStream_t st = (hipStream_t)stream;
#define G1_CASE(CBV, NSV) \
    if (pl.cb == CBV && pl.ns == NSV) return launch1<CBV, NSV>(d, pl, st);
    G1_CASE(4, 2) G1_CASE(4, 4) G1_CASE(2, 2) G1_CASE(2, 3) G1_CASE(2, 4) G1_CASE(2, 6) G1_CASE(2, 8) G1_CASE(1, 2) G1_CASE(1, 3) G1_CASE(1, 4) G1_CASE(1, 6)
    G1_CASE(1, 8)
#undef G1_CASE
    ppms_set_error("gemm1: no instantiation for CB=%d NS=%d", pl.cb, pl.ns);
    return PPMS_EINVAL;
}
