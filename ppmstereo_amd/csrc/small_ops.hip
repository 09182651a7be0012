// HBM-bound helper kernels of the hot path: depthwise conv, flow im2col, uncertainty tail, layout converters,
// convex / bilinear upsampling, frame similarity, QAM pick, attention operand preparation.
#include "common.h"

// ------------------------------------------------------------------------------------------------ errors
static thread_local char g_err[512] = "";
void ppms_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}
int ppms_check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        ppms_set_error("%s: HIP launch failed: %s", what, hipGetErrorString(e));
        return PPMS_ELAUNCH;
    }
    return PPMS_OK;
}
extern "C" const char* ppms_last_error(void) { return g_err; }
extern "C" int ppms_version(void) { return PPMS_ABI_VERSION; }
extern "C" int ppms_struct_sizes(int* sp, int* epilogue, int* conv) {
    if (sp) *sp = (int)sizeof(ppms_sp);
    if (epilogue) *epilogue = (int)sizeof(ppms_epilogue);
    if (conv) *conv = (int)sizeof(ppms_conv);
    return PPMS_OK;
}
extern "C" int ppms_device_info(char* name, int name_cap, int* cu_count, int* clock_mhz) {
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) {
        ppms_set_error("no HIP device");
        return PPMS_ENODEV;
    }
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, dev) != hipSuccess) {
        ppms_set_error("hipGetDeviceProperties failed");
        return PPMS_ENODEV;
    }
    if (name && name_cap > 0) snprintf(name, name_cap, "%s", prop.gcnArchName);
    if (cu_count) *cu_count = prop.multiProcessorCount;
    if (clock_mhz) *clock_mhz = prop.clockRate / 1000;
    return PPMS_OK;
}

// ------------------------------------------------------------------------------------------------ depthwise conv
// y = gelu(x + dw_k(x) + b), PCBlock4_Deep_nopool_res.forward (ppmtereo_update.py:1026-1027)
// one thread = a run of DW_PX pixels along x times 8 channels: per kernel row it loads the DW_PX + K - 1 window
// positions once (16-byte loads of the hi and lo planes, channel-last rows are contiguous) and sweeps the K taps over
// them from registers; the weights sit in LDS transposed to [tap][channel] (two b128 reads per tap).
constexpr int DW_PX = 2;            // (4: 31.6 us at the 1/4 scale, 2: 28.3 -- twice the threads hide more of the load latency)
template <int K>
__global__ __launch_bounds__(256) void dwconv_gelu_kernel(ppms_sp x, ppms_sp y, const float* __restrict__ w, const float* __restrict__ b,
                                                          int H, int W, int64_t rows, int groups) {
    __shared__ __attribute__((aligned(16))) float wl[K * K * 64];
    const int C = groups * 8;
    for (int i = threadIdx.x; i < K * K * C; i += 256) {
        const int c = i % C, tap = i / C;
        wl[tap * C + c] = w[c * K * K + tap];
    }
    __syncthreads();
    const int rpr = (W + DW_PX - 1) / DW_PX;                       // runs per image row
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= rows * rpr * groups) return;
    const int c0 = (int)(idx % groups) * 8;
    const int64_t run = idx / groups;
    const int px0 = (int)(run % rpr) * DW_PX;
    const int64_t row = run / rpr;                                 // frame * H + y
    const int py = (int)(row % H);
    const int64_t pix0 = row * W + px0;
    const bf16_t* xh = (const bf16_t*)x.hi + c0;
    const bf16_t* xl = (const bf16_t*)x.lo + c0;
    float acc[DW_PX][8], x0[DW_PX][8];
    {
        const f32x4 b0 = *(const f32x4*)(b + c0), b1 = *(const f32x4*)(b + c0 + 4);
#pragma unroll
        for (int p = 0; p < DW_PX; ++p)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc[p][j] = j < 4 ? b0[j & 3] : b1[j & 3];
                x0[p][j] = 0.0f;
            }
    }
    constexpr int R = K / 2, NW = DW_PX + K - 1;
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
        if (ky == R) {
#pragma unroll
            for (int p = 0; p < DW_PX; ++p)
#pragma unroll
                for (int j = 0; j < 8; ++j) x0[p][j] = win[p + R][j];
        }
#pragma unroll
        for (int kx = 0; kx < K; ++kx) {
            const f32x4 w0 = *(const f32x4*)(wl + (ky * K + kx) * C + c0), w1 = *(const f32x4*)(wl + (ky * K + kx) * C + c0 + 4);
#pragma unroll
            for (int p = 0; p < DW_PX; ++p)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[p][j] += (j < 4 ? w0[j & 3] : w1[j & 3]) * win[p + kx][j];   // out-of-image taps add 0 * w
        }
    }
#pragma unroll
    for (int p = 0; p < DW_PX; ++p) {
        if (px0 + p >= W) break;
        bf16x8 oh, ol;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            bf16_t hi, lo;
            split_bf16(gelu_erf(x0[p][j] + acc[p][j]), hi, lo);
            oh[j] = hi;
            ol[j] = lo;
        }
        *(bf16x8*)((bf16_t*)y.hi + (pix0 + p) * y.ld + c0) = oh;
        *(bf16x8*)((bf16_t*)y.lo + (pix0 + p) * y.ld + c0) = ol;
    }
}

extern "C" int ppms_dwconv_gelu(ppms_sp x, ppms_sp y, const float* w, const float* b, int k, int BT, int H, int W, void* stream) {
    PPMS_REQUIRE(k == 1 || k == 7, "dwconv_gelu: k=%d (only 1 and 7)", k);
    PPMS_REQUIRE(x.hi && x.lo && y.hi && y.lo && x.c == y.c && x.c > 0 && x.c <= 64 && x.c % 8 == 0 && x.ld % 8 == 0 && y.ld % 8 == 0,
                 "dwconv_gelu: views must have c <= 64 and c, ld multiples of 8");
    const int64_t rows = (int64_t)BT * H;
    const int groups = x.c / 8;
    const dim3 grid(ceil_div(rows * ((W + DW_PX - 1) / DW_PX) * groups, 256));
    if (k == 1)
        hipLaunchKernelGGL(dwconv_gelu_kernel<1>, grid, dim3(256), 0, (hipStream_t)stream, x, y, w, b, H, W, rows, groups);
    else
        hipLaunchKernelGGL(dwconv_gelu_kernel<7>, grid, dim3(256), 0, (hipStream_t)stream, x, y, w, b, H, W, rows, groups);
    return ppms_check_launch("dwconv_gelu");
}

// ------------------------------------------------------------------------------------------------ flow im2col (convf1 7x7, Cin=2)
__global__ __launch_bounds__(256) void flow_patch7_kernel(const float* __restrict__ flow, ppms_sp patch, int H, int W, int64_t P) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= P * 128) return;
    const int k = (int)(idx & 127);
    const int64_t pix = idx >> 7;
    float v = 0.0f;
    if (k < 98) {
        const int tap = k >> 1, c = k & 1;
        const int ky = tap / 7, kx = tap - ky * 7;
        const int px = (int)(pix % W), py = (int)((pix / W) % H);
        const int xx = px + kx - 3, yy = py + ky - 3;
        if ((unsigned)xx < (unsigned)W && (unsigned)yy < (unsigned)H) v = flow[(pix + (int64_t)(ky - 3) * W + (kx - 3)) * 2 + c];
    }
    bf16_t hi, lo;
    split_bf16(v, hi, lo);
    ((bf16_t*)patch.hi)[pix * patch.ld + k] = hi;
    ((bf16_t*)patch.lo)[pix * patch.ld + k] = lo;
}

extern "C" int ppms_flow_patch7(const float* flow_nhwc, ppms_sp patch, int BT, int H, int W, void* stream) {
    PPMS_REQUIRE(flow_nhwc && patch.hi && patch.lo && patch.ld >= 128, "flow_patch7: bad arguments");
    const int64_t P = (int64_t)BT * H * W;
    hipLaunchKernelGGL(flow_patch7_kernel, dim3(ceil_div(P * 128, 256)), dim3(256), 0, (hipStream_t)stream, flow_nhwc, patch, H, W, P);
    return ppms_check_launch("flow_patch7");
}

// ------------------------------------------------------------------------------------------------ uncertainty tail
// unc = sigmoid(w . x + b) (128 -> 1), plus deterministic per-(frame, 256-pixel block) partial sums
__global__ __launch_bounds__(256) void unc_tail_kernel(ppms_sp x, const float* __restrict__ w, float bias, float* __restrict__ unc,
                                                       float* __restrict__ partial, int HW, int nblk) {
    __shared__ float red[16];
    const int frame = blockIdx.y, blk = blockIdx.x;
    const int sub = threadIdx.x & 15, grp = threadIdx.x >> 4;   // 16 lanes per pixel, 8 channels each
    float wv[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) wv[j] = w[sub * 8 + j];
    // all 16 pixels of a lane group are requested before the first is consumed (the loop used to wait out one memory round trip per pixel:
    // 16 us per launch at every scale); the sums keep their order (per pixel: lanes by shuffle; per block: pixel order), so the
    // confidences are the same bits as before
    bf16x8 h8[16], l8[16];
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int pin = blk * 256 + it * 16 + grp;
        const int64_t pix = (int64_t)frame * HW + (pin < HW ? pin : HW - 1);
        h8[it] = gld<bf16x8>((const bf16_t*)x.hi + pix * x.ld + sub * 8);
        l8[it] = gld<bf16x8>((const bf16_t*)x.lo + pix * x.ld + sub * 8);
    }
    float local = 0.0f;
#pragma unroll
    for (int it = 0; it < 16; ++it) {
        const int pin = blk * 256 + it * 16 + grp;
        float s = 0.0f;
#pragma unroll
        for (int j = 0; j < 8; ++j) s += wv[j] * join_bf16(h8[it][j], l8[it][j]);
        s += __shfl_xor(s, 1);
        s += __shfl_xor(s, 2);
        s += __shfl_xor(s, 4);
        s += __shfl_xor(s, 8);
        if (pin < HW && sub == 0) {
            const float u = sigmoid_f(s + bias);
            unc[(int64_t)frame * HW + pin] = u;
            local += u;
        }
    }
    // fixed-order block sum: 16 group leaders -> LDS -> thread 0
    if (sub == 0) red[grp] = local;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.0f;
        for (int i = 0; i < 16; ++i) t += red[i];
        partial[frame * nblk + blk] = t;
    }
}

extern "C" int ppms_unc_tail(ppms_sp x, const float* w, float bias, float* unc, float* partial, int BT, int HW, void* stream) {
    PPMS_REQUIRE(x.hi && x.lo && x.c == 128 && x.ld % 8 == 0, "unc_tail: expects a 128-channel SP view");
    const int nblk = ceil_div(HW, 256);
    hipLaunchKernelGGL(unc_tail_kernel, dim3(nblk, BT), dim3(256), 0, (hipStream_t)stream, x, w, bias, unc, partial, HW, nblk);
    return ppms_check_launch("unc_tail");
}

// ------------------------------------------------------------------------------------------------ layout converters
// tiled transposes between [frame][C][HW] (NCHW) and [frame*HW][ld] (channel-last); 32x32 tiles through LDS
enum { CV_F32 = 0, CV_SP = 1 };
template <int DST_KIND>
__global__ __launch_bounds__(256) void nchw_to_cl_kernel(const float* __restrict__ src, float* __restrict__ dst_f32, int dst_ld,
                                                         ppms_sp dst_sp, int C, int HW) {
    __shared__ float tile[32][33];
    const int frame = blockIdx.z;
    const int c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;   // 8 rows per pass
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, p = p0 + tx;
        tile[i][tx] = (c < C && p < HW) ? src[((int64_t)frame * C + c) * HW + p] : 0.0f;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int p = p0 + i, c = c0 + tx;
        if (p < HW && c < C) {
            const float v = tile[tx][i];
            const int64_t pix = (int64_t)frame * HW + p;
            if (DST_KIND == CV_F32) {
                dst_f32[pix * dst_ld + c] = v;
            } else {
                bf16_t hi, lo;
                split_bf16(v, hi, lo);
                ((bf16_t*)dst_sp.hi)[pix * dst_sp.ld + c] = hi;
                ((bf16_t*)dst_sp.lo)[pix * dst_sp.ld + c] = lo;
            }
        }
    }
}
template <int SRC_KIND>
__global__ __launch_bounds__(256) void cl_to_nchw_kernel(const float* __restrict__ src_f32, int src_ld, ppms_sp src_sp,
                                                         float* __restrict__ dst, int C, int HW) {
    __shared__ float tile[32][33];
    const int frame = blockIdx.z;
    const int c0 = blockIdx.y * 32, p0 = blockIdx.x * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int i = ty; i < 32; i += 8) {
        const int p = p0 + i, c = c0 + tx;
        float v = 0.0f;
        if (p < HW && c < C) {
            const int64_t pix = (int64_t)frame * HW + p;
            if (SRC_KIND == CV_F32)
                v = src_f32[pix * src_ld + c];
            else
                v = join_bf16(((const bf16_t*)src_sp.hi)[pix * src_sp.ld + c], ((const bf16_t*)src_sp.lo)[pix * src_sp.ld + c]);
        }
        tile[i][tx] = v;
    }
    __syncthreads();
    for (int i = ty; i < 32; i += 8) {
        const int c = c0 + i, p = p0 + tx;
        if (c < C && p < HW) dst[((int64_t)frame * C + c) * HW + p] = tile[tx][i];
    }
}

extern "C" int ppms_nchw_to_sp(const float* src, ppms_sp dst, int BT, int C, int HW, void* stream) {
    PPMS_REQUIRE(src && dst.hi && dst.lo && dst.ld >= C, "nchw_to_sp: bad arguments");
    hipLaunchKernelGGL(nchw_to_cl_kernel<CV_SP>, dim3(ceil_div(HW, 32), ceil_div(C, 32), BT), dim3(256), 0, (hipStream_t)stream, src,
                       (float*)nullptr, 0, dst, C, HW);
    return ppms_check_launch("nchw_to_sp");
}
extern "C" int ppms_nchw_to_nhwc(const float* src, float* dst, int dst_ld, int BT, int C, int HW, void* stream) {
    PPMS_REQUIRE(src && dst && dst_ld >= C, "nchw_to_nhwc: bad arguments");
    ppms_sp none = {nullptr, nullptr, 0, 0};
    hipLaunchKernelGGL(nchw_to_cl_kernel<CV_F32>, dim3(ceil_div(HW, 32), ceil_div(C, 32), BT), dim3(256), 0, (hipStream_t)stream, src, dst,
                       dst_ld, none, C, HW);
    return ppms_check_launch("nchw_to_nhwc");
}
extern "C" int ppms_sp_to_nchw(ppms_sp src, float* dst, int BT, int C, int HW, void* stream) {
    PPMS_REQUIRE(dst && src.hi && src.lo && src.ld >= C, "sp_to_nchw: bad arguments");
    hipLaunchKernelGGL(cl_to_nchw_kernel<CV_SP>, dim3(ceil_div(HW, 32), ceil_div(C, 32), BT), dim3(256), 0, (hipStream_t)stream,
                       (const float*)nullptr, 0, src, dst, C, HW);
    return ppms_check_launch("sp_to_nchw");
}
extern "C" int ppms_nhwc_to_nchw(const float* src, int src_ld, float* dst, int BT, int C, int HW, void* stream) {
    PPMS_REQUIRE(src && dst && src_ld >= C, "nhwc_to_nchw: bad arguments");
    ppms_sp none = {nullptr, nullptr, 0, 0};
    hipLaunchKernelGGL(cl_to_nchw_kernel<CV_F32>, dim3(ceil_div(HW, 32), ceil_div(C, 32), BT), dim3(256), 0, (hipStream_t)stream, src,
                       src_ld, none, dst, C, HW);
    return ppms_check_launch("nhwc_to_nchw");
}

__global__ __launch_bounds__(256) void f32_to_sp_kernel(const float* __restrict__ src, int src_ld, ppms_sp dst, int64_t n) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const int c = (int)(idx % dst.c);
    const int64_t pix = idx / dst.c;
    bf16_t hi, lo;
    split_bf16(src[pix * src_ld + c], hi, lo);
    ((bf16_t*)dst.hi)[pix * dst.ld + c] = hi;
    ((bf16_t*)dst.lo)[pix * dst.ld + c] = lo;
}
__global__ __launch_bounds__(256) void sp_to_f32_kernel(ppms_sp src, float* __restrict__ dst, int dst_ld, int64_t n) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const int c = (int)(idx % src.c);
    const int64_t pix = idx / src.c;
    dst[pix * dst_ld + c] = join_bf16(((const bf16_t*)src.hi)[pix * src.ld + c], ((const bf16_t*)src.lo)[pix * src.ld + c]);
}
extern "C" int ppms_f32_to_sp(const float* src, int src_ld, ppms_sp dst, int64_t pixels, void* stream) {
    PPMS_REQUIRE(src && dst.hi && dst.lo && dst.c > 0, "f32_to_sp: bad arguments");
    const int64_t n = pixels * dst.c;
    hipLaunchKernelGGL(f32_to_sp_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, src, src_ld, dst, n);
    return ppms_check_launch("f32_to_sp");
}
extern "C" int ppms_sp_to_f32(ppms_sp src, float* dst, int dst_ld, int64_t pixels, void* stream) {
    PPMS_REQUIRE(dst && src.hi && src.lo && src.c > 0, "sp_to_f32: bad arguments");
    const int64_t n = pixels * src.c;
    hipLaunchKernelGGL(sp_to_f32_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, dst_ld, n);
    return ppms_check_launch("sp_to_f32");
}

// ------------------------------------------------------------------------------------------------ tap gather-sum
// Tail of a conv with very few output channels computed as a 1x1 GEMM to (taps x cout) channels:
//   out[p][c] = bias[c] + sum_tap y[p + d(tap)][tap*cout + c],  d(tap) = (kz - kt/2, ky - kh/2, kx - kw/2), zero outside.
// Used for FlowHead3D.conv2 (256 -> 2, 3x3x3, ppmtereo_update.py:674): 8 GEMM k-steps instead of 216.
__global__ __launch_bounds__(256) void tap_gather_sum_kernel(const float* __restrict__ y, int y_ld, const float* __restrict__ bias,
                                                             float* __restrict__ out, int out_ld, float* __restrict__ accum, int accum_ld, int cout,
                                                             int kt, int kh, int kw, int T, int H, int W, int t_halo, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % cout);
    const int64_t pix = idx / cout;
    const int x = (int)(pix % W), yy = (int)((pix / W) % H), t = (int)(pix / ((int64_t)W * H));
    float acc = bias ? bias[c] : 0.0f;
    int tap = 0;
    for (int kz = 0; kz < kt; ++kz)
        for (int ky = 0; ky < kh; ++ky)
            for (int kx = 0; kx < kw; ++kx, ++tap) {
                const int tt = t + kz - kt / 2, y2 = yy + ky - kh / 2, x2 = x + kx - kw / 2;
                if ((unsigned)(tt + t_halo) < (unsigned)(T + 2 * t_halo) && (unsigned)y2 < (unsigned)H && (unsigned)x2 < (unsigned)W)
                    acc += y[(((int64_t)tt * H + y2) * W + x2) * y_ld + tap * cout + c];
            }
    out[pix * out_ld + c] = acc;
    if (accum != nullptr) accum[pix * accum_ld + c] += acc;          // flow += delta_flow (ppmstereo.py:571) without a second launch
}
// 3 x 3 x 3 taps (the flow head's tail, 20 launches per clip), fully unrolled: all 27 loads of a thread are in flight together -- the generic
// kernel's runtime loops wait out a memory round trip per tap (12-14 us per launch whatever the map size).  Same order of additions.
__global__ __launch_bounds__(256) void tap_gather_sum333_kernel(const float* __restrict__ y, int y_ld, const float* __restrict__ bias,
                                                                float* __restrict__ out, int out_ld, float* __restrict__ accum, int accum_ld, int cout,
                                                                int T, int H, int W, int t_halo, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= total) return;
    const int c = (int)(idx % cout);
    const int64_t pix = idx / cout;
    const int x = (int)(pix % W), yy = (int)((pix / W) % H), t = (int)(pix / ((int64_t)W * H));
    float v[27];
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
        const int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
        const int tt = t + kz - 1, y2 = yy + ky - 1, x2 = x + kx - 1;
        const bool ok = (unsigned)(tt + t_halo) < (unsigned)(T + 2 * t_halo) && (unsigned)y2 < (unsigned)H && (unsigned)x2 < (unsigned)W;
        v[tap] = ok ? y[(((int64_t)tt * H + y2) * W + x2) * y_ld + tap * cout + c] : 0.0f;
    }
    float acc = bias ? bias[c] : 0.0f;
#pragma unroll
    for (int tap = 0; tap < 27; ++tap) {
        const int kz = tap / 9, ky = (tap / 3) % 3, kx = tap % 3;
        const int tt = t + kz - 1, y2 = yy + ky - 1, x2 = x + kx - 1;
        const bool ok = (unsigned)(tt + t_halo) < (unsigned)(T + 2 * t_halo) && (unsigned)y2 < (unsigned)H && (unsigned)x2 < (unsigned)W;
        if (ok) acc += v[tap];                                     // (skipped, not "+ 0": the generic kernel's sum, bit for bit)
    }
    out[pix * out_ld + c] = acc;
    if (accum != nullptr) accum[pix * accum_ld + c] += acc;
}
// accum (optional, [pixels][accum_ld], accum_ld >= cout): accum[p][c] += out[p][c] in the same launch
extern "C" int ppms_tap_gather_sum(const float* y, int y_ld, const float* bias, float* out, int out_ld, float* accum, int accum_ld, int cout, int kt,
                                   int kh, int kw, int T, int H, int W, int t_halo, void* stream) {
    PPMS_REQUIRE(y && out && cout > 0 && kt * kh * kw * cout <= y_ld && out_ld >= cout && t_halo >= 0, "tap_gather_sum: bad arguments");
    PPMS_REQUIRE(accum == nullptr || accum_ld >= cout, "tap_gather_sum: accum_ld=%d < cout=%d", accum_ld, cout);
    const int64_t total = (int64_t)T * H * W * cout;
    if (kt == 3 && kh == 3 && kw == 3)
        hipLaunchKernelGGL(tap_gather_sum333_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, y, y_ld, bias, out, out_ld, accum,
                           accum_ld, cout, T, H, W, t_halo, total);
    else
        hipLaunchKernelGGL(tap_gather_sum_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, y, y_ld, bias, out, out_ld, accum, accum_ld,
                           cout, kt, kh, kw, T, H, W, t_halo, total);
    return ppms_check_launch("tap_gather_sum");
}

// ------------------------------------------------------------------------------------------------ flow += delta_flow
__global__ __launch_bounds__(256) void flow_add_kernel(float* __restrict__ flow, const float* __restrict__ dflow, int dflow_ld, int64_t P) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= P * 2) return;
    flow[idx] += dflow[(idx >> 1) * dflow_ld + (idx & 1)];
}
extern "C" int ppms_flow_add(float* flow_nhwc, const float* dflow, int dflow_ld, int64_t pixels, void* stream) {
    PPMS_REQUIRE(flow_nhwc && dflow && dflow_ld >= 2, "flow_add: bad arguments");
    hipLaunchKernelGGL(flow_add_kernel, dim3(ceil_div(pixels * 2, 256)), dim3(256), 0, (hipStream_t)stream, flow_nhwc, dflow, dflow_ld, pixels);
    return ppms_check_launch("flow_add");
}

// ------------------------------------------------------------------------------------------------ convex upsample
// out[n,c,4y+i,4x+j] = sum_k softmax_k(mask[n,16k+4i+j,y,x]) * 4*flow[n,c,y+k/3-1,x+k%3-1]   (ppmstereo.py:185-197)
__global__ __launch_bounds__(256) void convex_upsample_kernel(const float* __restrict__ flow, const float* __restrict__ mask, int mask_ld,
                                                              float* __restrict__ out, int H, int W, int64_t P) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= P * 16) return;
    const int sub = (int)(idx & 15);
    const int64_t pix = idx >> 4;
    const int x = (int)(pix % W), y = (int)((pix / W) % H);
    const int64_t frame = pix / ((int64_t)H * W);
    const float* m = mask + pix * mask_ld + sub;
    float mv[9], mx = -INFINITY;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        mv[k] = m[16 * k];
        mx = fmaxf(mx, mv[k]);
    }
    float den = 0.0f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        mv[k] = expf(mv[k] - mx);
        den += mv[k];
    }
    float o0 = 0.0f, o1 = 0.0f;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int yy = y + k / 3 - 1, xx = x + k % 3 - 1;
        if ((unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
            const int64_t q = pix + (int64_t)(k / 3 - 1) * W + (k % 3 - 1);
            const float wgt = mv[k] / den;
            o0 += wgt * (4.0f * flow[q * 2]);
            o1 += wgt * (4.0f * flow[q * 2 + 1]);
        }
    }
    const int i = sub >> 2, j = sub & 3;
    const int64_t OW = 4 * (int64_t)W, OHW = 16 * (int64_t)H * W;
    const int64_t o = (frame * 2) * OHW + (int64_t)(4 * y + i) * OW + 4 * x + j;
    out[o] = o0;
    out[o + OHW] = o1;
}
extern "C" int ppms_convex_upsample(const float* flow_nhwc, const float* mask, int mask_ld, float* out, int BT, int H, int W, void* stream) {
    PPMS_REQUIRE(flow_nhwc && mask && out && mask_ld >= 144, "convex_upsample: bad arguments");
    const int64_t P = (int64_t)BT * H * W;
    hipLaunchKernelGGL(convex_upsample_kernel, dim3(ceil_div(P * 16, 256)), dim3(256), 0, (hipStream_t)stream, flow_nhwc, mask, mask_ld,
                       out, H, W, P);
    return ppms_check_launch("convex_upsample");
}

// PPMStereo.convex_upsample_3d, ppmstereo.py:199-228 (use_convex_3d=True): 27 neighbours (kt, ky, kx) of the (t, y, x) volume,
// out[t][c][4y+i][4x+j] = sum_k softmax_k(mask[t][16k + 4i + j][y][x]) * 4 flow[t+kt-1][c][y+ky-1][x+kx-1], zero padded
// (unfoldNd.UnfoldNd([3,3,3], padding=1): k = (kt*3 + ky)*3 + kx).  One thread = one (pixel, sub-position i*4+j).
__global__ __launch_bounds__(256) void convex_upsample3d_kernel(const float* __restrict__ flow, const float* __restrict__ mask, int mask_ld,
                                                                float* __restrict__ out, int T, int H, int W, int t_halo, int64_t P) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= P * 16) return;
    const int sub = (int)(idx & 15);
    const int64_t pix = idx >> 4;
    const int x = (int)(pix % W), y = (int)((pix / W) % H);
    const int64_t frame = pix / ((int64_t)W * H);
    const float* mp = mask + pix * mask_ld + sub;
    float mv[27];
    float mx = mp[0];
#pragma unroll
    for (int k = 0; k < 27; ++k) {
        mv[k] = mp[16 * k];
        mx = fmaxf(mx, mv[k]);
    }
    float den = 0.0f;
#pragma unroll
    for (int k = 0; k < 27; ++k) {
        mv[k] = expf(mv[k] - mx);
        den += mv[k];
    }
    float o0 = 0.0f, o1 = 0.0f;
#pragma unroll
    for (int k = 0; k < 27; ++k) {
        const int dt = k / 9 - 1, dy = (k / 3) % 3 - 1, dx = k % 3 - 1;
        const int tt = (int)frame + dt, yy = y + dy, xx = x + dx;
        if ((unsigned)(tt + t_halo) < (unsigned)(T + 2 * t_halo) && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W) {
            const int64_t q = pix + ((int64_t)dt * H + dy) * W + dx;
            const float wgt = mv[k] / den;
            o0 += wgt * (4.0f * flow[q * 2]);
            o1 += wgt * (4.0f * flow[q * 2 + 1]);
        }
    }
    const int i = sub >> 2, j = sub & 3;
    const int64_t OW = 4 * (int64_t)W, OHW = 16 * (int64_t)H * W;
    const int64_t o = (frame * 2) * OHW + (int64_t)(4 * y + i) * OW + 4 * x + j;
    out[o] = o0;
    out[o + OHW] = o1;
}
extern "C" int ppms_convex_upsample_3d(const float* flow_nhwc, const float* mask, int mask_ld, float* out, int T, int H, int W, int t_halo, void* stream) {
    PPMS_REQUIRE(flow_nhwc && mask && out && mask_ld >= 432 && T > 0 && H > 0 && W > 0 && t_halo >= 0, "convex_upsample_3d: bad arguments");
    const int64_t P = (int64_t)T * H * W;
    hipLaunchKernelGGL(convex_upsample3d_kernel, dim3(ceil_div(P * 16, 256)), dim3(256), 0, (hipStream_t)stream, flow_nhwc, mask, mask_ld,
                       out, T, H, W, t_halo, P);
    return ppms_check_launch("convex_upsample_3d");
}

// ------------------------------------------------------------------------------------------------ bilinear resize
// torch upsample_bilinear2d semantics (aten/native/UpSample.h area_pixel_compute_source_index)
__global__ __launch_bounds__(256) void bilinear_kernel(const float* __restrict__ src, float* __restrict__ dst, int H, int W, int OH, int OW,
                                                       int align, float sh, float sw, float mul, int64_t n) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const int ox = (int)(idx % OW);
    const int oy = (int)((idx / OW) % OH);
    const int64_t plane = idx / ((int64_t)OW * OH);
    float fy = align ? sh * oy : fmaxf(sh * (oy + 0.5f) - 0.5f, 0.0f);
    float fx = align ? sw * ox : fmaxf(sw * (ox + 0.5f) - 0.5f, 0.0f);
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + ((y0 < H - 1) ? 1 : 0), x1 = x0 + ((x0 < W - 1) ? 1 : 0);
    const float ly = fy - y0, lx = fx - x0;
    const float hy = 1.0f - ly, hx = 1.0f - lx;
    const float* s = src + plane * H * W;
    const float v = hy * (hx * s[y0 * W + x0] + lx * s[y0 * W + x1]) + ly * (hx * s[y1 * W + x0] + lx * s[y1 * W + x1]);
    dst[idx] = mul * v;
}
extern "C" int ppms_bilinear(const float* src, float* dst, int N, int C, int H, int W, int OH, int OW, int align_corners, float mul,
                             void* stream) {
    PPMS_REQUIRE(src && dst && N > 0 && C > 0 && H > 0 && W > 0 && OH > 0 && OW > 0, "bilinear: bad arguments");
    float sh, sw;
    if (align_corners) {
        sh = OH > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.0f;
        sw = OW > 1 ? (float)(W - 1) / (float)(OW - 1) : 0.0f;
    } else {
        sh = (float)H / (float)OH;
        sw = (float)W / (float)OW;
    }
    const int64_t n = (int64_t)N * C * OH * OW;
    hipLaunchKernelGGL(bilinear_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, H, W, OH, OW, align_corners,
                       sh, sw, mul, n);
    return ppms_check_launch("bilinear");
}

// Scale-to-scale hand-over of the cascade on SP (channel-last split-bf16) tensors, no NCHW round trip:
// dst = a * dst + b * interp(src) with F.interpolate(mode="bilinear", align_corners=True) semantics per frame
// (ppmstereo.py:726-732,763-767: hidden state x2, net = (net_s + interp(net_2s)) / 2).  One thread = one output pixel x 8 channels.
__global__ __launch_bounds__(256) void sp_resize_blend_kernel(ppms_sp src, ppms_sp dst, int H, int W, int OH, int OW, int C8, float sh, float sw,
                                                              float a, float b, int64_t n) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const int c = (int)(idx % C8) * 8;
    const int64_t opix = idx / C8;
    const int ox = (int)(opix % OW);
    const int oy = (int)((opix / OW) % OH);
    const int64_t frame = opix / ((int64_t)OW * OH);
    const float fy = sh * oy, fx = sw * ox;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + ((y0 < H - 1) ? 1 : 0), x1 = x0 + ((x0 < W - 1) ? 1 : 0);
    const float ly = fy - y0, lx = fx - x0;
    const float hy = 1.0f - ly, hx = 1.0f - lx;
    const int64_t base = frame * H * W;
    auto ld8 = [&](int y, int x, float* v) {
        const int64_t o = (base + (int64_t)y * W + x) * src.ld + c;
        const bf16x8 h8 = *(const bf16x8*)((const bf16_t*)src.hi + o), l8 = *(const bf16x8*)((const bf16_t*)src.lo + o);
#pragma unroll
        for (int j = 0; j < 8; ++j) v[j] = join_bf16(h8[j], l8[j]);
    };
    float v00[8], v01[8], v10[8], v11[8];
    ld8(y0, x0, v00);
    ld8(y0, x1, v01);
    ld8(y1, x0, v10);
    ld8(y1, x1, v11);
    const int64_t od = opix * dst.ld + c;
    float d[8];
    if (a != 0.0f) {
        const bf16x8 h8 = *(const bf16x8*)((const bf16_t*)dst.hi + od), l8 = *(const bf16x8*)((const bf16_t*)dst.lo + od);
#pragma unroll
        for (int j = 0; j < 8; ++j) d[j] = join_bf16(h8[j], l8[j]);
    }
    bf16x8 oh, ol;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float v = hy * (hx * v00[j] + lx * v01[j]) + ly * (hx * v10[j] + lx * v11[j]);
        const float y = (a != 0.0f) ? a * d[j] + b * v : b * v;
        bf16_t hh, ll;
        split_bf16(y, hh, ll);
        oh[j] = hh;
        ol[j] = ll;
    }
    *(bf16x8*)((bf16_t*)dst.hi + od) = oh;
    *(bf16x8*)((bf16_t*)dst.lo + od) = ol;
}
extern "C" int ppms_sp_resize_blend(ppms_sp src, ppms_sp dst, int N, int H, int W, int OH, int OW, float a, float b, void* stream) {
    PPMS_REQUIRE(src.hi && src.lo && dst.hi && dst.lo && N > 0 && H > 0 && W > 0 && OH > 0 && OW > 0, "sp_resize_blend: bad arguments");
    PPMS_REQUIRE(src.c == dst.c && src.c % 8 == 0 && src.ld % 8 == 0 && dst.ld % 8 == 0, "sp_resize_blend: channel views must match, multiples of 8");
    PPMS_REQUIRE((((uintptr_t)src.hi | (uintptr_t)src.lo | (uintptr_t)dst.hi | (uintptr_t)dst.lo) & 15) == 0, "sp_resize_blend: views must be 16-B aligned");
    const float sh = OH > 1 ? (float)(H - 1) / (float)(OH - 1) : 0.0f, sw = OW > 1 ? (float)(W - 1) / (float)(OW - 1) : 0.0f;
    const int64_t n = (int64_t)N * OH * OW * (src.c / 8);
    hipLaunchKernelGGL(sp_resize_blend_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, H, W, OH, OW, src.c / 8, sh, sw, a, b, n);
    return ppms_check_launch("sp_resize_blend");
}

// ------------------------------------------------------------------------------------------------ pre-loop glue of PPMStereo.forward
// (ppmstereo.py:620-682, between the encoders and the loop: the caller of the hot path)
// F.avg_pool2d(x, k, stride=k) on NCHW fp32 planes (k = 4: 1/4 -> 1/16 features :649-650, k = 2: :666-671)
__global__ __launch_bounds__(256) void avgpool_kernel(const float* __restrict__ src, float* __restrict__ dst, int H, int W, int k, int64_t n) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const int OW = W / k, OH = H / k;
    const int ox = (int)(idx % OW), oy = (int)((idx / OW) % OH);
    const int64_t plane = idx / ((int64_t)OW * OH);
    const float* s = src + plane * H * W + (int64_t)oy * k * W + ox * k;
    float acc = 0.0f;
    for (int dy = 0; dy < k; ++dy)
        for (int dx = 0; dx < k; ++dx) acc += s[dy * W + dx];
    dst[idx] = acc / (float)(k * k);
}
extern "C" int ppms_avgpool(const float* src, float* dst, int planes, int H, int W, int k, void* stream) {
    PPMS_REQUIRE(src && dst && planes > 0 && k > 0 && H >= k && W >= k, "avgpool: bad arguments");
    const int64_t n = (int64_t)planes * (H / k) * (W / k);
    hipLaunchKernelGGL(avgpool_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, src, dst, H, W, k, n);
    return ppms_check_launch("avgpool");
}
// out = a * x + b * y[i mod period]  (period = n: plain axpby; period = C*H*W: y broadcast over frames, the positional encoding
// of :327-333; a = b = 0.5: the feature / context averages of :627-628, :666-671)
__global__ __launch_bounds__(256) void axpby_kernel(const float* __restrict__ x, const float* __restrict__ y, float* __restrict__ out, float a, float b,
                                                    int64_t period, int64_t n) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (idx < n) out[idx] = a * x[idx] + b * y[idx % period];
}
extern "C" int ppms_axpby(const float* x, const float* y, float* out, float a, float b, int64_t period, int64_t n, void* stream) {
    PPMS_REQUIRE(x && y && out && n > 0 && period > 0, "axpby: bad arguments");
    hipLaunchKernelGGL(axpby_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, x, y, out, a, b, period, n);
    return ppms_check_launch("axpby");
}
// net = tanh((f[:, :128] + c[:, :128]) / 2), inp = relu((f[:, 128:] + c[:, 128:]) / 2)   (:620-632, :660-664, :673-682);
// f, c: NCHW (N, 256, HW); net, inp: NCHW (N, 128, HW).  tanhf: the library-accurate form (one-off per scale).
__global__ __launch_bounds__(256) void ctx_mix_kernel(const float* __restrict__ f, const float* __restrict__ c, float* __restrict__ net,
                                                      float* __restrict__ inp, int64_t chw, int64_t n) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;      // over N * 128 * HW
    if (idx >= n) return;
    const int64_t frame = idx / chw, rem = idx - frame * chw;
    const int64_t src = frame * 2 * chw + rem;
    net[idx] = tanhf((f[src] + c[src]) / 2.0f);
    const float v = (f[src + chw] + c[src + chw]) / 2.0f;
    inp[idx] = v < 0.0f ? 0.0f : v;
}
extern "C" int ppms_ctx_mix(const float* fmap, const float* ctx, float* net, float* inp, int N, int HW, void* stream) {
    PPMS_REQUIRE(fmap && ctx && net && inp && N > 0 && HW > 0, "ctx_mix: bad arguments");
    const int64_t chw = (int64_t)128 * HW, n = (int64_t)N * chw;
    hipLaunchKernelGGL(ctx_mix_kernel, dim3(ceil_div(n, 256)), dim3(256), 0, (hipStream_t)stream, fmap, ctx, net, inp, chw, n);
    return ppms_check_launch("ctx_mix");
}

// ------------------------------------------------------------------------------------------------ frame similarity
// pooled[which][t][cell] = mean_c max_{adaptive window} x_t[c]  (AdaptiveMaxPool2d(h//4,w//4) + mean over channels)
__global__ __launch_bounds__(128) void qk_pool_kernel(const float* __restrict__ q, const float* __restrict__ k, int ld,
                                                      float* __restrict__ pooled, int T, int H, int W, int OH, int OW) {
    __shared__ float red[128];
    const int cell = blockIdx.x, t = blockIdx.y, which = blockIdx.z;
    const int oy = cell / OW, ox = cell - oy * OW;
    const int ys = (oy * H) / OH, ye = ((oy + 1) * H + OH - 1) / OH;
    const int xs = (ox * W) / OW, xe = ((ox + 1) * W + OW - 1) / OW;
    const float* src = (which ? k : q) + (int64_t)t * H * W * ld + threadIdx.x;
    float m = -INFINITY;
    for (int y = ys; y < ye; ++y)
        for (int x = xs; x < xe; ++x) m = fmaxf(m, src[((int64_t)y * W + x) * ld]);
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 64; s > 0; s >>= 1) {
        if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) pooled[((int64_t)which * T + t) * OH * OW + cell] = red[0] / 128.0f;
}
// sim[i][j] = cos(kbar_i, qbar_j), eps 1e-8 (F.cosine_similarity)
__global__ __launch_bounds__(64) void qk_cos_kernel(const float* __restrict__ pooled, float* __restrict__ sim, int T, int L) {
    const int i = blockIdx.y, j = blockIdx.x;
    const float* qv = pooled + (int64_t)j * L;
    const float* kv = pooled + ((int64_t)T + i) * L;
    float dot = 0.0f, nq = 0.0f, nk = 0.0f;
    for (int e = threadIdx.x; e < L; e += 64) {
        const float a = qv[e], b = kv[e];
        dot += a * b;
        nq += a * a;
        nk += b * b;
    }
    for (int s = 32; s > 0; s >>= 1) {
        dot += __shfl_xor(dot, s);
        nq += __shfl_xor(nq, s);
        nk += __shfl_xor(nk, s);
    }
    if (threadIdx.x == 0) sim[i * T + j] = dot / (fmaxf(sqrtf(nq), 1e-8f) * fmaxf(sqrtf(nk), 1e-8f));
}
extern "C" int ppms_qk_pool(const float* q, const float* k, int ld, float* pooled, int T, int H, int W, void* stream) {
    PPMS_REQUIRE(q && k && pooled && T > 0 && H >= 4 && W >= 4 && ld >= 128, "qk_pool: bad arguments (H,W >= 4)");
    const int OH = H / 4, OW = W / 4;
    hipLaunchKernelGGL(qk_pool_kernel, dim3(OH * OW, T, 2), dim3(128), 0, (hipStream_t)stream, q, k, ld, pooled, T, H, W, OH, OW);
    return ppms_check_launch("qk_pool");
}
extern "C" int ppms_qk_cos(const float* pooled, float* sim, int T, int cells, void* stream) {
    PPMS_REQUIRE(pooled && sim && T > 0 && cells > 0, "qk_cos: bad arguments");
    hipLaunchKernelGGL(qk_cos_kernel, dim3(T, T), dim3(64), 0, (hipStream_t)stream, pooled, sim, T, cells);
    return ppms_check_launch("qk_cos");
}
extern "C" int ppms_qk_similarity(const float* q, const float* k, int ld, float* pooled, float* sim, int T, int H, int W, void* stream) {
    PPMS_REQUIRE(q && k && pooled && sim && T > 0 && H >= 4 && W >= 4 && ld >= 128, "qk_similarity: bad arguments (H,W >= 4)");
    const int OH = H / 4, OW = W / 4;
    hipLaunchKernelGGL(qk_pool_kernel, dim3(OH * OW, T, 2), dim3(128), 0, (hipStream_t)stream, q, k, ld, pooled, T, H, W, OH, OW);
    hipLaunchKernelGGL(qk_cos_kernel, dim3(T, T), dim3(64), 0, (hipStream_t)stream, pooled, sim, T, OH * OW);
    return ppms_check_launch("qk_similarity");
}

// ------------------------------------------------------------------------------------------------ QAM pick
// one lane per clip row i: score[i][j] = exp(-S_ij / (sum_j S_ij + T)) * sim[i][j] + conf[j]; top-5 by score;
// S[i][sel] += 1; picked frames ascending + normalised scores s_hat = score / mean(score over picked)
__global__ __launch_bounds__(64) void qam_select_kernel(const float* __restrict__ sim, float* __restrict__ strive,
                                                        const float* __restrict__ partial, int nblk, int HW, int32_t* __restrict__ sel,
                                                        float* __restrict__ shat, float* __restrict__ score_out, int T) {
    __shared__ float conf[64];
    const int i = threadIdx.x;
    // frame confidences = mean of the uncertainty map: the block sums of frame f are added by all 64 lanes (lane l takes blocks l, l + 64,
    // ...; then a butterfly) -- a fixed order, the same on every rank of a sharded window; one lane per frame walking nblk sums in a
    // dependent chain cost most of this kernel's 11 us
    for (int f = 0; f < T; ++f) {
        float s = 0.0f;
        for (int b = i; b < nblk; b += 64) s += partial[f * nblk + b];
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
        if (i == 0) conf[f] = s / (float)HW;
    }
    __syncthreads();
    if (i >= T) return;
    float ssum = 0.0f;
    for (int j = 0; j < T; ++j) ssum += strive[i * T + j];
    const int ksel = T < 5 ? T : 5;
    unsigned long long picked = 0ull;
    float sc[64];
    for (int j = 0; j < T; ++j) {
        const float pen = expf(-strive[i * T + j] / (ssum + (float)T));
        sc[j] = pen * sim[i * T + j] + conf[j];
        if (score_out) score_out[i * T + j] = sc[j];
    }
    for (int n = 0; n < ksel; ++n) {
        int best = -1;
        float bv = -INFINITY;
        for (int j = 0; j < T; ++j) {
            if ((picked >> j) & 1ull) continue;
            const float v = sc[j];
            if (best < 0 || v > bv || (v != v && bv == bv)) {   // NaN scores (T == 1) still pick a frame
                best = j;
                bv = v;
            }
        }
        picked |= 1ull << best;
    }
    float mean = 0.0f;
    for (int j = 0; j < T; ++j)
        if ((picked >> j) & 1ull) mean += sc[j];
    mean /= (float)ksel;
    int n = 0;
    for (int j = 0; j < T; ++j)
        if ((picked >> j) & 1ull) {
            strive[i * T + j] += 1.0f;
            sel[i * 5 + n] = j;
            shat[i * 5 + n] = sc[j] / mean;
            ++n;
        }
    for (; n < 5; ++n) {
        sel[i * 5 + n] = 0;
        shat[i * 5 + n] = 0.0f;
    }
}
extern "C" int ppms_qam_select(const float* sim, float* strive, const float* conf_partial, int nblk, int HW, int32_t* sel, float* shat,
                               float* score, int T, void* stream) {
    PPMS_REQUIRE(sim && strive && conf_partial && sel && shat && T > 0 && T <= 64, "qam_select: bad arguments (1 <= T <= 64)");
    hipLaunchKernelGGL(qam_select_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, sim, strive, conf_partial, nblk, HW, sel, shat, score, T);
    return ppms_check_launch("qam_select");
}

// ------------------------------------------------------------------------------------------------ attention operands
__global__ __launch_bounds__(256) void attn_prep_q_kernel(const float* __restrict__ q, int ld, const float* __restrict__ pe,
                                                          bf16_t* __restrict__ qb, int n, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;     // (frame, pixel, 8-channel group)
    if (idx >= total) return;
    const int g = (int)(idx & 15);
    const int64_t pix = idx >> 4;
    const int frame = (int)(pix / n);
    const f32x4 a = *(const f32x4*)(q + pix * ld + g * 8), b = *(const f32x4*)(q + pix * ld + g * 8 + 4);
    const f32x4 pa = *(const f32x4*)(pe + frame * 128 + g * 8), pb = *(const f32x4*)(pe + frame * 128 + g * 8 + 4);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        o[j] = (bf16_t)(a[j] + pa[j]);
        o[4 + j] = (bf16_t)(b[j] + pb[j]);
    }
    *(bf16x8*)(qb + pix * 128 + g * 8) = o;
}
extern "C" int ppms_attn_prep_q(const float* q, int ld, const float* pe, void* qb, int T, int n, void* stream) {
    PPMS_REQUIRE(q && pe && qb && ld % 4 == 0, "attn_prep_q: bad arguments");
    const int64_t total = (int64_t)T * n * 16;
    hipLaunchKernelGGL(attn_prep_q_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, q, ld, pe, (bf16_t*)qb, n, total);
    return ppms_check_launch("attn_prep_q");
}

__global__ __launch_bounds__(256) void attn_prep_k_kernel(const float* __restrict__ key, int ld, const float* __restrict__ pe,
                                                          const int32_t* __restrict__ sel, const float* __restrict__ shat,
                                                          bf16_t* __restrict__ kb, int ksel, int n, int64_t total) {
    const int64_t idx = (int64_t)blockIdx.x * 256 + threadIdx.x;     // (clip, slot, pixel, 8-channel group)
    if (idx >= total) return;
    const int g = (int)(idx & 15);
    const int64_t r = idx >> 4;
    const int pin = (int)(r % n);
    const int cs = (int)(r / n);                // clip*ksel + slot
    const int clip = cs / ksel, slot = cs - clip * ksel;
    const int frame = sel[clip * 5 + slot];
    const float s = shat[clip * 5 + slot];
    const float* kp = key + ((int64_t)frame * n + pin) * ld + g * 8;
    const f32x4 a = *(const f32x4*)kp, b = *(const f32x4*)(kp + 4);
    const f32x4 pa = *(const f32x4*)(pe + frame * 128 + g * 8), pb = *(const f32x4*)(pe + frame * 128 + g * 8 + 4);
    bf16x8 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        o[j] = (bf16_t)(a[j] * s + pa[j]);
        o[4 + j] = (bf16_t)(b[j] * s + pb[j]);
    }
    *(bf16x8*)(kb + r * 128 + g * 8) = o;
}
extern "C" int ppms_attn_prep_k(const float* key, int ld, const float* pe, const int32_t* sel, const float* shat, void* kb, int T, int ksel,
                                int n, void* stream) {
    PPMS_REQUIRE(key && pe && sel && shat && kb && ksel >= 1 && ksel <= 5 && ld % 4 == 0, "attn_prep_k: bad arguments");
    const int64_t total = (int64_t)T * ksel * n * 16;
    hipLaunchKernelGGL(attn_prep_k_kernel, dim3(ceil_div(total, 256)), dim3(256), 0, (hipStream_t)stream, key, ld, pe, sel, shat,
                       (bf16_t*)kb, ksel, n, total);
    return ppms_check_launch("attn_prep_k");
}
