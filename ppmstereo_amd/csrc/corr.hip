// 1-D all-pairs correlation pyramid: build (fp32 MFMA, exact fp32 FMA chain) and multi-level lookup.
// Replaces CorrBlock1D of /root/reference/models/core/corr.py:55-104 (einsum :102, avg_pool2d :71,
// grid_sample :21).  HBM-bound kernels: the volume is written once and gathered once per iteration.
#include "common.h"

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

extern "C" int ppms_corr_build(const float* fmap1, const float* fmap2, float* const pyr[5], int B, int C, int H, int W, void* stream) {
    PPMS_REQUIRE(B > 0 && C > 0 && H > 0 && W > 0, "corr_build: bad shape B=%d C=%d H=%d W=%d", B, C, H, W);
    // the reference's 4 avg_pool2d([1,2]) calls need a width of at least 16 (corr.py:70-72; SURVEY hazard 2)
    PPMS_REQUIRE((W >> 4) >= 1, "corr_build: W=%d too small for a 4-level pyramid (needs W >= 16 at this scale)", W);
    dim3 grid((W + 31) / 32, B * H);
    hipLaunchKernelGGL(corr_build_kernel, grid, dim3(256), 0, (hipStream_t)stream, fmap1, fmap2, pyr[0], pyr[1], pyr[2], pyr[3],
                       pyr[4], C, H, W);
    return ppms_check_launch("corr_build");
}

// ------------------------------------------------------------------------------------------------
// lookup: 4 levels x 9 taps of linear interpolation at x + flow_x, zero padding, align_corners=True.
// ------------------------------------------------------------------------------------------------
__device__ __forceinline__ float lookup_tap(const float* __restrict__ L, int Wl, float xs, int lvl, int kk) {
    // same fp32 op sequence as the reference: normalise (corr.py:14,85) then grid_sample's un-normalise
    const float pos = (float)(kk - 4) + xs / (float)(1 << lvl);
    const float wm1 = (float)(Wl - 1);
    const float g = 2.0f * pos / wm1 - 1.0f;
    const float p = ((g + 1.0f) / 2.0f) * wm1;
    const float pf = floorf(p);
    const float a = p - pf;
    // far out-of-range positions (|p| beyond int range) contribute nothing
    if (!(pf >= -1.0f && pf <= (float)Wl)) return 0.0f;
    const int i0 = (int)pf, i1 = i0 + 1;
    const float v0 = (i0 >= 0 && i0 < Wl) ? L[i0] : 0.0f;
    const float v1 = (i1 >= 0 && i1 < Wl) ? L[i1] : 0.0f;
    return (1.0f - a) * v0 + a * v1;
}

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
