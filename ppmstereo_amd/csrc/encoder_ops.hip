// Pieces of the feature encoder `BasicEncoder(output_dim=256, norm_fn="instance")` (fnet, SURVEY.md section 8 row f3;
// /root/reference/models/core/extractor.py:302-423) that are not convolutions.  The convolutions run on the implicit-GEMM
// kernels of conv_gemm2/5.hip; the two stride-2 layers (conv1 7x7 s2, layer2.0 conv1 3x3 s2 + its 1x1 s2 skip) become stride-1
// convolutions on a 2x2 space-to-depth copy of their input (host side: ppmstereo_amd/encoder.py re-lays the weights).
#include "common.h"

namespace {

// ---- space to depth, factor 2: dst[(n, i, j)][phase * C + c] = src[(n, 2i + dy, 2j + dx)][c], phase = 2 dy + dx -------------
// image form: src is the NCHW fp32 image batch the reference hands to fnet (3 channels); channels >= 4 C of dst are zeroed
__global__ __launch_bounds__(256) void img_s2d_kernel(const float* __restrict__ img, ppms_sp dst, int C, int H, int W, int64_t npix) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;          // one thread = one output pixel x 8 channels
    const int groups = dst.c >> 3;
    if (idx >= npix * groups) return;
    const int g8 = (int)(idx % groups);
    const int64_t p = idx / groups;
    const int OW = W >> 1, OH = H >> 1;
    const int j = (int)(p % OW), i = (int)((p / OW) % OH);
    const int64_t n = p / ((int64_t)OW * OH);
    bf16x8 oh, ol;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ch = g8 * 8 + e;
        float v = 0.0f;
        if (ch < 4 * C) {
            const int ph = ch / C, c = ch - ph * C;
            v = img[((n * C + c) * H + 2 * i + (ph >> 1)) * W + 2 * j + (ph & 1)];
        }
        bf16_t hh, ll;
        split_bf16(v, hh, ll);
        oh[e] = hh;
        ol[e] = ll;
    }
    const int64_t od = p * dst.ld + g8 * 8;
    *(bf16x8*)((bf16_t*)dst.hi + od) = oh;
    *(bf16x8*)((bf16_t*)dst.lo + od) = ol;
}

// split-plane form: pure copies of 16-B channel groups (both planes)
__global__ __launch_bounds__(256) void sp_s2d_kernel(ppms_sp src, ppms_sp dst, int H, int W, int64_t npix) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;          // one thread = one output pixel x 8 channels
    const int C = src.c, groups = (4 * C) >> 3;
    if (idx >= npix * groups) return;
    const int g8 = (int)(idx % groups);
    const int64_t p = idx / groups;
    const int OW = W >> 1, OH = H >> 1;
    const int j = (int)(p % OW), i = (int)((p / OW) % OH);
    const int64_t n = p / ((int64_t)OW * OH);
    const int ch = g8 * 8, ph = ch / C, c = ch - ph * C;                  // C % 8 == 0: a group never straddles two phases
    const int64_t sp = ((n * H + 2 * i + (ph >> 1)) * W + 2 * j + (ph & 1)) * src.ld + c;
    const int64_t od = p * dst.ld + ch;
    *(bf16x8*)((bf16_t*)dst.hi + od) = *(const bf16x8*)((const bf16_t*)src.hi + sp);
    *(bf16x8*)((bf16_t*)dst.lo + od) = *(const bf16x8*)((const bf16_t*)src.lo + sp);
}

// ---- InstanceNorm2d(affine=False, eps): per (sample, channel) mean and biased variance over the H*W pixels -------------------
// x: channel-last fp32 [N*HW][ld] (a conv's fp32 output).  Two launches, deterministic:
//  1. one workgroup = one sample x 32 channels x one of S pixel slices: slice mean, then slice sum of centred squares (two passes
//     over an L2-resident slice: no cancellation) -> part[n][s][c] = (mean_s, M2_s);
//  2. one thread per (sample, channel) merges the S slices in slice order with Chan's update
//     (mean += d n_b / n, M2 += M2_b + d^2 n_a n_b / n) -> stats[n][c] = (mean, 1 / sqrt(M2 / HW + eps)).
static int in_slices_host(int N, int HW, int C) {                      // enough workgroups to fill the chip, >= 64 pixels each
    const int per = (int)ceil_div(C, 32) * N;
    int S = 1024 / (per > 0 ? per : 1);
    const int smax = HW / 64;
    if (S > smax) S = smax;
    return S < 1 ? 1 : S;
}

__global__ __launch_bounds__(256) void instnorm_part_kernel(const float* __restrict__ x, int ld, int HW, int C, int S, float* __restrict__ part) {
    __shared__ float red[8][32];
    __shared__ float mean_s[32];
    const int n = blockIdx.y, c0 = blockIdx.x * 32, s = blockIdx.z;
    const int lane = threadIdx.x & 31, row = threadIdx.x >> 5;            // 8 pixel rows x 32 channels per step
    const int c = c0 + lane;
    const int chunk = (HW + S - 1) / S;
    const int p0 = s * chunk, p1 = (p0 + chunk < HW) ? p0 + chunk : HW;
    const int cnt = p1 - p0;
    const float* xp = x + (int64_t)n * HW * ld;
    float acc = 0.0f;
    if (c < C)
        for (int p = p0 + row; p < p1; p += 8) acc += xp[(int64_t)p * ld + c];
    red[row][lane] = acc;
    __syncthreads();
    if (row == 0) {
        float t = 0.0f;
        for (int r = 0; r < 8; ++r) t += red[r][lane];
        mean_s[lane] = cnt > 0 ? t / (float)cnt : 0.0f;
    }
    __syncthreads();
    const float mean = mean_s[lane];
    acc = 0.0f;
    if (c < C)
        for (int p = p0 + row; p < p1; p += 8) {
            const float d = xp[(int64_t)p * ld + c] - mean;
            acc += d * d;
        }
    __syncthreads();
    red[row][lane] = acc;
    __syncthreads();
    if (row == 0 && c < C) {
        float t = 0.0f;
        for (int r = 0; r < 8; ++r) t += red[r][lane];
        float* o = part + (((int64_t)n * S + s) * C + c) * 2;
        o[0] = mean;
        o[1] = t;
    }
}

__global__ __launch_bounds__(256) void instnorm_merge_kernel(const float* __restrict__ part, int HW, int C, int S, float eps, int total, float* __restrict__ stats) {
    const int idx = blockIdx.x * 256 + threadIdx.x;                        // (n, c)
    if (idx >= total) return;
    const int n = idx / C, c = idx - n * C;
    const int chunk = (HW + S - 1) / S;
    float na = 0.0f, mean = 0.0f, m2 = 0.0f;
    for (int s = 0; s < S; ++s) {
        const int p0 = s * chunk, p1 = (p0 + chunk < HW) ? p0 + chunk : HW;
        const float nb = (float)(p1 - p0);
        if (nb <= 0.0f) continue;
        const float* o = part + (((int64_t)n * S + s) * C + c) * 2;
        const float d = o[0] - mean, nn = na + nb;
        mean += d * (nb / nn);
        m2 += o[1] + d * d * (na * nb / nn);
        na = nn;
    }
    stats[(int64_t)idx * 2] = mean;
    stats[(int64_t)idx * 2 + 1] = 1.0f / sqrtf(m2 / (float)HW + eps);
}

// y = (x - mean) * rstd  [+ res]  [relu]  -> split planes; channels >= C of `out` (padding of the next conv's input) are zeroed
__global__ __launch_bounds__(256) void instnorm_apply_kernel(const float* __restrict__ x, int ld, const float* __restrict__ stats, ppms_sp res, int relu,
                                                             ppms_sp out, int HW, int C, int64_t npix) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;          // one thread = one pixel x 8 channels
    const int groups = out.c >> 3;
    if (idx >= npix * groups) return;
    const int g8 = (int)(idx % groups);
    const int64_t p = idx / groups;
    const int64_t n = p / HW;
    const int c0 = g8 * 8;
    bf16x8 oh, ol, rh, rl;
    const bool has_res = res.hi != nullptr && c0 < C;
    if (has_res) {
        rh = *(const bf16x8*)((const bf16_t*)res.hi + p * res.ld + c0);
        rl = *(const bf16x8*)((const bf16_t*)res.lo + p * res.ld + c0);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = c0 + e;
        float v = 0.0f;
        if (c < C) {
            const float* st = stats + ((int64_t)n * C + c) * 2;
            v = (x[p * ld + c] - st[0]) * st[1];
            if (has_res) v += join_bf16(rh[e], rl[e]);
            if (relu) v = fmaxf(v, 0.0f);
        }
        bf16_t hh, ll;
        split_bf16(v, hh, ll);
        oh[e] = hh;
        ol[e] = ll;
    }
    const int64_t od = p * out.ld + c0;
    *(bf16x8*)((bf16_t*)out.hi + od) = oh;
    *(bf16x8*)((bf16_t*)out.lo + od) = ol;
}

}  // namespace

extern "C" int ppms_img_s2d(const float* img, ppms_sp dst, int N, int C, int H, int W, void* stream) {
    PPMS_REQUIRE(img && dst.hi && dst.lo && N > 0 && C > 0 && H > 0 && W > 0, "img_s2d: bad arguments");
    PPMS_REQUIRE(H % 2 == 0 && W % 2 == 0, "img_s2d: H = %d, W = %d must be even", H, W);
    PPMS_REQUIRE(dst.c >= 4 * C && dst.c % 8 == 0 && dst.ld % 8 == 0 && (((uintptr_t)dst.hi | (uintptr_t)dst.lo) & 15) == 0,
                 "img_s2d: destination view needs >= %d channels, multiples of 8, 16-B aligned", 4 * C);
    const int64_t npix = (int64_t)N * (H / 2) * (W / 2);
    hipLaunchKernelGGL(img_s2d_kernel, dim3(ceil_div(npix * (dst.c / 8), 256)), dim3(256), 0, (hipStream_t)stream, img, dst, C, H, W, npix);
    return ppms_check_launch("img_s2d");
}

extern "C" int ppms_sp_s2d(ppms_sp src, ppms_sp dst, int N, int H, int W, void* stream) {
    PPMS_REQUIRE(src.hi && src.lo && dst.hi && dst.lo && N > 0 && H > 0 && W > 0, "sp_s2d: bad arguments");
    PPMS_REQUIRE(H % 2 == 0 && W % 2 == 0, "sp_s2d: H = %d, W = %d must be even", H, W);
    PPMS_REQUIRE(src.c % 8 == 0 && dst.c == 4 * src.c && src.ld % 8 == 0 && dst.ld % 8 == 0, "sp_s2d: dst must have 4x the channels of src (multiples of 8)");
    PPMS_REQUIRE((((uintptr_t)src.hi | (uintptr_t)src.lo | (uintptr_t)dst.hi | (uintptr_t)dst.lo) & 15) == 0, "sp_s2d: views must be 16-B aligned");
    const int64_t npix = (int64_t)N * (H / 2) * (W / 2);
    hipLaunchKernelGGL(sp_s2d_kernel, dim3(ceil_div(npix * (dst.c / 8), 256)), dim3(256), 0, (hipStream_t)stream, src, dst, H, W, npix);
    return ppms_check_launch("sp_s2d");
}

extern "C" int64_t ppms_instnorm_workspace_bytes(int N, int HW, int C) {
    if (N <= 0 || HW <= 0 || C <= 0) return 0;
    return (int64_t)N * in_slices_host(N, HW, C) * C * 2 * 4;
}

extern "C" int ppms_instnorm_stats(const float* x, int ld, int N, int HW, int C, float eps, float* stats, void* workspace, void* stream) {
    PPMS_REQUIRE(x && stats && workspace && N > 0 && HW > 0 && C > 0 && ld >= C && eps > 0.0f, "instnorm_stats: bad arguments");
    const int S = in_slices_host(N, HW, C);
    hipLaunchKernelGGL(instnorm_part_kernel, dim3(ceil_div(C, 32), N, S), dim3(256), 0, (hipStream_t)stream, x, ld, HW, C, S, (float*)workspace);
    hipLaunchKernelGGL(instnorm_merge_kernel, dim3(ceil_div(N * C, 256)), dim3(256), 0, (hipStream_t)stream, (const float*)workspace, HW, C, S, eps, N * C,
                       stats);
    return ppms_check_launch("instnorm_stats");
}

extern "C" int ppms_instnorm_apply(const float* x, int ld, const float* stats, ppms_sp res, int relu, ppms_sp out, int N, int HW, int C, void* stream) {
    PPMS_REQUIRE(x && stats && out.hi && out.lo && N > 0 && HW > 0 && C > 0 && ld >= C, "instnorm_apply: bad arguments");
    PPMS_REQUIRE(out.c >= C && out.c % 8 == 0 && out.ld % 8 == 0 && (((uintptr_t)out.hi | (uintptr_t)out.lo) & 15) == 0,
                 "instnorm_apply: output view needs >= %d channels, multiples of 8, 16-B aligned", C);
    PPMS_REQUIRE(res.hi == nullptr || (res.lo && res.c >= ((C + 7) / 8) * 8 && res.ld % 8 == 0 && (((uintptr_t)res.hi | (uintptr_t)res.lo) & 15) == 0),
                 "instnorm_apply: residual view must cover the normalised channels (multiples of 8, 16-B aligned)");
    const int64_t npix = (int64_t)N * HW;
    hipLaunchKernelGGL(instnorm_apply_kernel, dim3(ceil_div(npix * (out.c / 8), 256)), dim3(256), 0, (hipStream_t)stream, x, ld, stats, res, relu, out, HW, C,
                       npix);
    return ppms_check_launch("instnorm_apply");
}
