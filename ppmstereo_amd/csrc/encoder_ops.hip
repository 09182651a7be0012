// Pieces of the feature encoder `BasicEncoder(output_dim=256, norm_fn="instance")` (fnet, SURVEY.md section 8 row f3;
// /root/reference/models/core/extractor.py:302-423) that are not convolutions.  The convolutions run on the implicit-GEMM
// kernels of conv_gemm2/5.hip; the two stride-2 layers (conv1 7x7 s2, layer2.0 conv1 3x3 s2 + its 1x1 s2 skip) become stride-1
// convolutions on a 2x2 space-to-depth copy of their input (host side: ppmstereo_amd/encoder.py re-lays the weights).
#include "common.h"

namespace {

// ---- space to depth, factor k: dst[(n, i, j)][phase * C + c] = src[(n, k i + dy, k j + dx)][c], phase = k dy + dx -------------
// image form: src is the NCHW fp32 image batch the reference hands to its encoders (3 channels); channels >= k*k*C of dst are zeroed
__global__ __launch_bounds__(256) void img_s2d_kernel(const float* __restrict__ img, ppms_sp dst, int C, int H, int W, int k, int64_t npix) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;          // one thread = one output pixel x 8 channels
    const int groups = dst.c >> 3;
    if (idx >= npix * groups) return;
    const int g8 = (int)(idx % groups);
    const int64_t p = idx / groups;
    const int OW = W / k, OH = H / k;
    const int j = (int)(p % OW), i = (int)((p / OW) % OH);
    const int64_t n = p / ((int64_t)OW * OH);
    bf16x8 oh, ol;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ch = g8 * 8 + e;
        float v = 0.0f;
        if (ch < k * k * C) {
            const int ph = ch / C, c = ch - ph * C;
            v = img[((n * C + c) * H + k * i + ph / k) * W + k * j + ph % k];
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

// thread = 4 channels (one 16-byte load) of every 32nd pixel of the slice: 8 threads cover the workgroup's 32 channels, 32 pixel rows per step
// (the 4-byte-per-lane form of round 2 ran at 2.4 TB/s: 4x the load instructions)
__global__ __launch_bounds__(256) void instnorm_part_kernel(const float* __restrict__ x, int ld, int HW, int C, int S, float* __restrict__ part) {
    __shared__ float red[32][33];
    __shared__ float mean_s[32];
    const int n = blockIdx.y, c0 = blockIdx.x * 32, s = blockIdx.z;
    const int q = threadIdx.x & 7, row = threadIdx.x >> 3;                // channels c0 + 4 q .. + 3, pixel rows row, row + 32, ...
    const int c = c0 + 4 * q;
    const int chunk = (HW + S - 1) / S;
    const int p0 = s * chunk, p1 = (p0 + chunk < HW) ? p0 + chunk : HW;
    const int cnt = p1 - p0;
    const float* xp = x + (int64_t)n * HW * ld + c;
    const bool vec = (ld & 3) == 0 && c + 4 <= C && (((uintptr_t)x) & 15) == 0;      // (uniform per thread; the tail channels of a C % 4 != 0 layer go one by one)
    auto load4 = [&](int p, float (&v)[4]) {
        if (vec) {
            const f32x4 t = *(const f32x4*)(xp + (int64_t)p * ld);
            v[0] = t[0], v[1] = t[1], v[2] = t[2], v[3] = t[3];
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (c + j < C) ? xp[(int64_t)p * ld + j] : 0.0f;
        }
    };
    auto reduce_rows = [&](const float (&acc)[4], float scale_by, float* dst) {   // sum over the 32 row groups in row order -> dst[32 channels]
#pragma unroll
        for (int j = 0; j < 4; ++j) red[row][4 * q + j] = acc[j];
        __syncthreads();
        if (threadIdx.x < 32) {
            float t = 0.0f;
            for (int r = 0; r < 32; ++r) t += red[r][threadIdx.x];
            dst[threadIdx.x] = t * scale_by;
        }
        __syncthreads();
    };
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int p = p0 + row; p < p1; p += 32) {
        float v[4];
        load4(p, v);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += v[j];
    }
    reduce_rows(acc, cnt > 0 ? 1.0f / (float)cnt : 0.0f, mean_s);
    float mean[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) mean[j] = mean_s[4 * q + j], acc[j] = 0.0f;
    for (int p = p0 + row; p < p1; p += 32) {                            // second pass: the slice is L2 resident
        float v[4];
        load4(p, v);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float d = v[j] - mean[j];
            acc[j] += d * d;
        }
    }
    __shared__ float m2_s[32];
    reduce_rows(acc, 1.0f, m2_s);
    if (threadIdx.x < 32 && c0 + (int)threadIdx.x < C) {
        float* o = part + (((int64_t)n * S + s) * C + c0 + threadIdx.x) * 2;
        o[0] = mean_s[threadIdx.x];
        o[1] = m2_s[threadIdx.x];
    }
}

__global__ __launch_bounds__(256) void instnorm_merge_kernel(const float* __restrict__ part, int HW, int C, int S, float eps, int total, float* __restrict__ stats) {
    const int idx = blockIdx.x * 256 + threadIdx.x;                        // (n, c)
    if (idx >= total) return;
    const int n = idx / C, c = idx - n * C;
    const int chunk = (HW + S - 1) / S;
    float na = 0.0f, mean = 0.0f, m2 = 0.0f;
    const f32x2* pp = (const f32x2*)(part + ((int64_t)n * S * C + c) * 2);            // slice s at pp[s * C]
    for (int s0 = 0; s0 < S; s0 += 8) {          // the 8 loads of a group are in flight together (one by one the S ~ 50 dependent-looking
        f32x2 o[8];                              // round trips were the kernel: 13.6 us); the update order stays slice order
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (s0 + j < S) ? pp[(int64_t)(s0 + j) * C] : (f32x2){0.0f, 0.0f};
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int s = s0 + j;
            const int p0 = s * chunk, p1 = (p0 + chunk < HW) ? p0 + chunk : HW;
            const float nb = (float)(p1 - p0);
            if (s >= S || nb <= 0.0f) continue;
            const float d = o[j][0] - mean, nn = na + nb;
            mean += d * (nb / nn);
            m2 += o[j][1] + d * d * (na * nb / nn);
            na = nn;
        }
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
    float xv[8], sm[8], sr[8];
    if (c0 + 8 <= C && (ld & 3) == 0 && (C & 1) == 0 && ((((uintptr_t)x) | ((uintptr_t)stats)) & 15) == 0) {       // 16-byte loads: the pixel's 8 values and their 8 (mean, rstd) pairs
        const f32x4 a = *(const f32x4*)(x + p * ld + c0), b = *(const f32x4*)(x + p * ld + c0 + 4);
        const f32x4* st4 = (const f32x4*)(stats + ((int64_t)n * C + c0) * 2);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            xv[e] = a[e], xv[4 + e] = b[e];
            const f32x4 t = st4[e];
            sm[2 * e] = t[0], sr[2 * e] = t[1], sm[2 * e + 1] = t[2], sr[2 * e + 1] = t[3];
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = c0 + e;
            const bool in = c < C;
            xv[e] = in ? x[p * ld + c] : 0.0f;
            sm[e] = in ? stats[((int64_t)n * C + c) * 2] : 0.0f;
            sr[e] = in ? stats[((int64_t)n * C + c) * 2 + 1] : 0.0f;
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = c0 + e;
        float v = 0.0f;
        if (c < C) {
            v = (xv[e] - sm[e]) * sr[e];
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

// ================================================================================================ cnet (ConvNeXt-V2 + FPN) pieces
// (/root/reference/models/core/convnext.py: SURVEY.md section 8 row f5)
// nn.Upsample(scale_factor=2), nearest (:226-238): dst[(n, y, x)] = src[(n, y / 2, x / 2)]; split planes, 16-B channel groups
__global__ __launch_bounds__(256) void sp_upsample2_kernel(ppms_sp src, ppms_sp dst, int H, int W, int64_t npix) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;          // one thread = one OUTPUT pixel x 8 channels
    const int groups = src.c >> 3;
    if (idx >= npix * groups) return;
    const int g8 = (int)(idx % groups);
    const int64_t p = idx / groups;
    const int OW = 2 * W, OH = 2 * H;
    const int x = (int)(p % OW), y = (int)((p / OW) % OH);
    const int64_t n = p / ((int64_t)OW * OH);
    const int64_t sp = ((n * H + (y >> 1)) * W + (x >> 1)) * src.ld + g8 * 8;
    const int64_t od = p * dst.ld + g8 * 8;
    *(bf16x8*)((bf16_t*)dst.hi + od) = *(const bf16x8*)((const bf16_t*)src.hi + sp);
    *(bf16x8*)((bf16_t*)dst.lo + od) = *(const bf16x8*)((const bf16_t*)src.lo + sp);
}

// depthwise 7 x 7 convolution + bias (Block.dwconv, :60): split planes in, fp32 channel-last out.  Same scheme as dwconv_gelu_kernel
// (small_ops.hip): one thread = a run of 4 pixels along x times 8 channels, per kernel row the 4 + 6 window positions are loaded once
// (16-byte loads of both planes) and the 7 taps sweep them from registers; a workgroup serves one block of <= 64 channels (blockIdx.y)
// whose weights sit in LDS transposed to [tap][channel].  Weights [C][49] as in the state_dict.
constexpr int DWP_PX = 4;
__global__ __launch_bounds__(256) void dwconv_plain_kernel(ppms_sp x, float* __restrict__ y, int ldy, const float* __restrict__ w, const float* __restrict__ b,
                                                           int H, int W, int64_t rows) {
    constexpr int K = 7, R = 3, NW = DWP_PX + K - 1;
    __shared__ __attribute__((aligned(16))) float wl[K * K * 64];
    const int cb = blockIdx.y * 64;                                        // first channel of this workgroup's block
    const int cn = (x.c - cb) < 64 ? (x.c - cb) : 64;                      // channels in it (multiple of 8)
    for (int i = threadIdx.x; i < K * K * cn; i += 256) {
        const int c = i % cn, tap = i / cn;
        wl[tap * 64 + c] = w[(cb + c) * K * K + tap];
    }
    __syncthreads();
    const int groups = cn >> 3;
    const int rpr = (W + DWP_PX - 1) / DWP_PX;                             // runs per image row
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * rpr * groups) return;
    const int cl = (int)(idx % groups) * 8, c0 = cb + cl;
    const int64_t run = idx / groups;
    const int px0 = (int)(run % rpr) * DWP_PX;
    const int64_t row = run / rpr;                                         // sample * H + y
    const int py = (int)(row % H);
    const int64_t pix0 = row * W + px0;
    const bf16_t* xh = (const bf16_t*)x.hi + c0;
    const bf16_t* xl = (const bf16_t*)x.lo + c0;
    float acc[DWP_PX][8];
    {
        const f32x4 b0 = *(const f32x4*)(b + c0), b1 = *(const f32x4*)(b + c0 + 4);
#pragma unroll
        for (int p = 0; p < DWP_PX; ++p)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[p][j] = j < 4 ? b0[j & 3] : b1[j & 3];
    }
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {
        const int yy = py + ky - R;
        if ((unsigned)yy >= (unsigned)H) continue;
        float win[NW][8];
#pragma unroll
        for (int i = 0; i < NW; ++i) {
            const int xx = px0 - R + i;
            bf16x8 h8 = {0, 0, 0, 0, 0, 0, 0, 0}, l8 = {0, 0, 0, 0, 0, 0, 0, 0};
            if ((unsigned)xx < (unsigned)W) {
                const int64_t q = pix0 + (int64_t)(ky - R) * W + (i - R);
                h8 = *(const bf16x8*)(xh + q * x.ld);
                l8 = *(const bf16x8*)(xl + q * x.ld);
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) win[i][j] = join_bf16(h8[j], l8[j]);
        }
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            const f32x4 w0 = *(const f32x4*)(wl + (ky * K + kx) * 64 + cl), w1 = *(const f32x4*)(wl + (ky * K + kx) * 64 + cl + 4);
#pragma unroll
            for (int p = 0; p < DWP_PX; ++p)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[p][j] += win[p + kx][j] * (j < 4 ? w0[j & 3] : w1[j & 3]);
        }
    }
#pragma unroll
    for (int p = 0; p < DWP_PX; ++p) {
        if (px0 + p >= W) break;
        float* o = y + (pix0 + p) * ldy + c0;
        *(f32x4*)o = (f32x4){acc[p][0], acc[p][1], acc[p][2], acc[p][3]};
        *(f32x4*)(o + 4) = (f32x4){acc[p][4], acc[p][5], acc[p][6], acc[p][7]};
    }
}

// LayerNorm over the channels of a pixel, any C (convnext.py:11-35, eps 1e-6: both data formats are this per-pixel op on
// channel-last data): x fp32 [pixel][ld] -> split planes.  One wave per pixel, lanes stride the channels.
__global__ __launch_bounds__(256) void layernorm_any_kernel(const float* __restrict__ x, int ld, const float* __restrict__ w, const float* __restrict__ b,
                                                            float eps, ppms_sp out, int64_t pixels, int C) {
    // a wave per pixel; a lane owns 8 consecutive channels per round of 512 (two 16-byte loads, one 16-byte store per plane); the pixel's
    // values stay in registers between the three passes (C <= 1024: the host checks); sums in lane order, then a butterfly: deterministic
    const int lane = threadIdx.x & 63;
    const int64_t pix = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pix >= pixels) return;
    const float* xp = x + pix * ld;
    const bool vec = (ld & 3) == 0 && (C & 7) == 0 && ((((uintptr_t)x) | ((uintptr_t)w) | ((uintptr_t)b)) & 15) == 0;
    float v[2][8];
    float s = 0.0f;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int c0 = r * 512 + lane * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j) v[r][j] = 0.0f;
        if (c0 < C) {
            if (vec) {
                const f32x4 a = *(const f32x4*)(xp + c0), a2 = *(const f32x4*)(xp + c0 + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) v[r][j] = a[j], v[r][4 + j] = a2[j];
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j)
                    if (c0 + j < C) v[r][j] = xp[c0 + j];
            }
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[r][j];
        }
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)C;
    float q = 0.0f;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int c0 = r * 512 + lane * 8;
#pragma unroll
        for (int j = 0; j < 8; ++j)
            if (c0 + j < C) q += (v[r][j] - mean) * (v[r][j] - mean);
    }
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = 1.0f / sqrtf(q / (float)C + eps);
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int c0 = r * 512 + lane * 8;
        if (c0 >= out.c) continue;
        bf16x8 oh, ol;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int c = c0 + j;
            const float y = c < C ? (v[r][j] - mean) * rstd * w[c] + b[c] : 0.0f;
            bf16_t hi, lo;
            split_bf16(y, hi, lo);
            oh[j] = hi;
            ol[j] = lo;
        }
        if (c0 + 8 <= out.c && (out.ld & 7) == 0 && ((((uintptr_t)out.hi) | ((uintptr_t)out.lo)) & 15) == 0) {
            *(bf16x8*)((bf16_t*)out.hi + pix * out.ld + c0) = oh;
            *(bf16x8*)((bf16_t*)out.lo + pix * out.ld + c0) = ol;
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (c0 + j < out.c) {
                    ((bf16_t*)out.hi)[pix * out.ld + c0 + j] = oh[j];
                    ((bf16_t*)out.lo)[pix * out.ld + c0 + j] = ol[j];
                }
        }
    }
}

// GRN (convnext.py:37-48) on channel-last fp32 h [N*HW][ld]: Gx[n][c] = ||h[n, :, c]||_2 over the pixels; Nx = Gx / (mean_c Gx + 1e-6);
// out = gamma * (h * Nx) + beta + h.  Squares summed per pixel slice (part kernel), slices merged in order + the channel mean
// (one workgroup per sample), then the apply kernel.  Deterministic.
__global__ __launch_bounds__(256) void grn_part_kernel(const float* __restrict__ x, int ld, int HW, int C, int S, float* __restrict__ part) {
    __shared__ float red[32][33];
    const int n = blockIdx.y, c0 = blockIdx.x * 32, s = blockIdx.z;
    const int q = threadIdx.x & 7, row = threadIdx.x >> 3;                // thread = 4 channels (one 16-byte load) of every 32nd pixel of the slice
    const int c = c0 + 4 * q;
    const int chunk = (HW + S - 1) / S;
    const int p0 = s * chunk, p1 = (p0 + chunk < HW) ? p0 + chunk : HW;
    const float* xp = x + (int64_t)n * HW * ld + c;
    const bool vec = (ld & 3) == 0 && c + 4 <= C && (((uintptr_t)x) & 15) == 0;
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
    for (int p = p0 + row; p < p1; p += 32) {
        float v[4];
        if (vec) {
            const f32x4 t = *(const f32x4*)(xp + (int64_t)p * ld);
            v[0] = t[0], v[1] = t[1], v[2] = t[2], v[3] = t[3];
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (c + j < C) ? xp[(int64_t)p * ld + j] : 0.0f;
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[j] += v[j] * v[j];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) red[row][4 * q + j] = acc[j];
    __syncthreads();
    if (threadIdx.x < 32 && c0 + (int)threadIdx.x < C) {
        float t = 0.0f;
        for (int r = 0; r < 32; ++r) t += red[r][threadIdx.x];
        part[((int64_t)n * S + s) * C + c0 + threadIdx.x] = t;
    }
}
__global__ __launch_bounds__(256) void grn_merge_kernel(const float* __restrict__ part, int C, int S, float* __restrict__ nx) {
    __shared__ float red[256];
    const int n = blockIdx.x;
    float local = 0.0f;
    for (int c = threadIdx.x; c < C; c += 256) {
        float t = 0.0f;
        const float* pp = part + (int64_t)n * S * C + c;
        for (int s0 = 0; s0 < S; s0 += 8) {      // 8 loads in flight, summed in slice order
            float v[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (s0 + j < S) ? pp[(int64_t)(s0 + j) * C] : 0.0f;
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (s0 + j < S) t += v[j];
        }
        const float g = sqrtf(t);
        nx[(int64_t)n * C + c] = g;
        local += g;
    }
    red[threadIdx.x] = local;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if (threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    const float inv = 1.0f / (red[0] / (float)C + 1e-6f);
    for (int c = threadIdx.x; c < C; c += 256) nx[(int64_t)n * C + c] *= inv;
}
__global__ __launch_bounds__(256) void grn_apply_kernel(const float* __restrict__ x, int ld, const float* __restrict__ nx, const float* __restrict__ gamma,
                                                        const float* __restrict__ beta, ppms_sp out, int HW, int C, int64_t npix) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;          // one thread = one pixel x 8 channels
    const int groups = C >> 3;
    if (idx >= npix * groups) return;
    const int c0 = (int)(idx % groups) * 8;
    const int64_t p = idx / groups;
    const int64_t n = p / HW;
    bf16x8 oh, ol;
    float xv[8], nv[8], gv[8], bv[8];
    if ((ld & 3) == 0 && ((((uintptr_t)x) | ((uintptr_t)nx) | ((uintptr_t)gamma) | ((uintptr_t)beta)) & 15) == 0) {      // (C % 8 == 0: the host checks)
#pragma unroll
        for (int h4 = 0; h4 < 2; ++h4) {
            const f32x4 a = *(const f32x4*)(x + p * ld + c0 + 4 * h4), b4 = *(const f32x4*)(nx + n * C + c0 + 4 * h4);
            const f32x4 g4 = *(const f32x4*)(gamma + c0 + 4 * h4), e4 = *(const f32x4*)(beta + c0 + 4 * h4);
#pragma unroll
            for (int j = 0; j < 4; ++j) xv[4 * h4 + j] = a[j], nv[4 * h4 + j] = b4[j], gv[4 * h4 + j] = g4[j], bv[4 * h4 + j] = e4[j];
        }
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) xv[e] = x[p * ld + c0 + e], nv[e] = nx[n * C + c0 + e], gv[e] = gamma[c0 + e], bv[e] = beta[c0 + e];
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float v = xv[e];
        const float y = gv[e] * (v * nv[e]) + bv[e] + v;
        bf16_t hh, ll;
        split_bf16(y, hh, ll);
        oh[e] = hh;
        ol[e] = ll;
    }
    const int64_t od = p * out.ld + c0;
    *(bf16x8*)((bf16_t*)out.hi + od) = oh;
    *(bf16x8*)((bf16_t*)out.lo + od) = ol;
}

}  // namespace

extern "C" int ppms_sp_upsample2(ppms_sp src, ppms_sp dst, int N, int H, int W, void* stream) {
    PPMS_REQUIRE(src.hi && src.lo && dst.hi && dst.lo && N > 0 && H > 0 && W > 0, "sp_upsample2: bad arguments");
    PPMS_REQUIRE(src.c % 8 == 0 && dst.c == src.c && src.ld % 8 == 0 && dst.ld % 8 == 0, "sp_upsample2: equal channel counts, multiples of 8");
    PPMS_REQUIRE((((uintptr_t)src.hi | (uintptr_t)src.lo | (uintptr_t)dst.hi | (uintptr_t)dst.lo) & 15) == 0, "sp_upsample2: views must be 16-B aligned");
    const int64_t npix = (int64_t)N * 4 * H * W;
    hipLaunchKernelGGL(sp_upsample2_kernel, dim3(ceil_div(npix * (src.c / 8), 256)), dim3(256), 0, (hipStream_t)stream, src, dst, H, W, npix);
    return ppms_check_launch("sp_upsample2");
}

extern "C" int ppms_dwconv(ppms_sp x, float* y, int ldy, const float* w, const float* b, int k, int N, int H, int W, void* stream) {
    PPMS_REQUIRE(x.hi && x.lo && y && w && b && N > 0 && H > 0 && W > 0 && k == 7, "dwconv: bad arguments (k = 7: ConvNeXt's depthwise kernel)");
    PPMS_REQUIRE(x.c % 8 == 0 && x.ld % 8 == 0 && ldy >= x.c && ldy % 4 == 0 && (((uintptr_t)x.hi | (uintptr_t)x.lo | (uintptr_t)y) & 15) == 0,
                 "dwconv: channel counts multiples of 8, 16-B aligned operands");
    const int64_t rows = (int64_t)N * H;
    const int cblocks = (int)ceil_div(x.c, 64);
    const int64_t per_block = rows * ((W + DWP_PX - 1) / DWP_PX) * 8;       // threads of a full 64-channel block
    hipLaunchKernelGGL(dwconv_plain_kernel, dim3(ceil_div(per_block, 256), cblocks), dim3(256), 0, (hipStream_t)stream, x, y, ldy, w, b, H, W, rows);
    return ppms_check_launch("dwconv");
}

extern "C" int ppms_layernorm_any(const float* x, int ld, const float* w, const float* b, float eps, ppms_sp out, int64_t pixels, int C, void* stream) {
    PPMS_REQUIRE(x && w && b && out.hi && out.lo && pixels > 0 && C > 0 && ld >= C && out.c >= C && eps > 0.0f, "layernorm_any: bad arguments");
    PPMS_REQUIRE(out.c <= 1024, "layernorm_any: at most 1024 channels (got %d)", out.c);
    hipLaunchKernelGGL(layernorm_any_kernel, dim3(ceil_div(pixels, 4)), dim3(256), 0, (hipStream_t)stream, x, ld, w, b, eps, out, pixels, C);
    return ppms_check_launch("layernorm_any");
}

extern "C" int64_t ppms_grn_workspace_bytes(int N, int HW, int C) {
    if (N <= 0 || HW <= 0 || C <= 0) return 0;
    return ((int64_t)N * in_slices_host(N, HW, C) * C + (int64_t)N * C) * 4;
}

extern "C" int ppms_grn(const float* x, int ld, const float* gamma, const float* beta, ppms_sp out, int N, int HW, int C, void* workspace, void* stream) {
    PPMS_REQUIRE(x && gamma && beta && out.hi && out.lo && workspace && N > 0 && HW > 0 && C > 0 && C % 8 == 0 && ld >= C, "grn: bad arguments");
    PPMS_REQUIRE(out.c >= C && out.ld % 8 == 0 && (((uintptr_t)out.hi | (uintptr_t)out.lo) & 15) == 0, "grn: output view must cover C channels, 16-B aligned");
    const int S = in_slices_host(N, HW, C);
    float* part = (float*)workspace;
    float* nx = part + (size_t)N * S * C;
    hipLaunchKernelGGL(grn_part_kernel, dim3(ceil_div(C, 32), N, S), dim3(256), 0, (hipStream_t)stream, x, ld, HW, C, S, part);
    hipLaunchKernelGGL(grn_merge_kernel, dim3(N), dim3(256), 0, (hipStream_t)stream, (const float*)part, C, S, nx);
    const int64_t npix = (int64_t)N * HW;
    hipLaunchKernelGGL(grn_apply_kernel, dim3(ceil_div(npix * (C / 8), 256)), dim3(256), 0, (hipStream_t)stream, x, ld, (const float*)nx, gamma, beta, out, HW,
                       C, npix);
    return ppms_check_launch("grn");
}

extern "C" int ppms_img_s2d(const float* img, ppms_sp dst, int N, int C, int H, int W, int k, void* stream) {
    PPMS_REQUIRE(img && dst.hi && dst.lo && N > 0 && C > 0 && H > 0 && W > 0 && (k == 2 || k == 4), "img_s2d: bad arguments (k = 2 or 4)");
    PPMS_REQUIRE(H % k == 0 && W % k == 0, "img_s2d: H = %d, W = %d must be multiples of %d", H, W, k);
    PPMS_REQUIRE(dst.c >= k * k * C && dst.c % 8 == 0 && dst.ld % 8 == 0 && (((uintptr_t)dst.hi | (uintptr_t)dst.lo) & 15) == 0,
                 "img_s2d: destination view needs >= %d channels, multiples of 8, 16-B aligned", k * k * C);
    const int64_t npix = (int64_t)N * (H / k) * (W / k);
    hipLaunchKernelGGL(img_s2d_kernel, dim3(ceil_div(npix * (dst.c / 8), 256)), dim3(256), 0, (hipStream_t)stream, img, dst, C, H, W, k, npix);
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
