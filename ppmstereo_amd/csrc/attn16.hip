// update_block16's TimeAttnBlock / SpaceAttnBlock pieces that are not plain GEMMs
// (/root/reference/models/core/ppmtereo_update.py:593-631 with Attention :400-420, and the LoFTR linear attention of
// /root/reference/models/core/attention.py:73-100,164-190) -- C = 384 there; the same kernels with C = 256 serve the SST block
// of the 1/16 features (ppmstereo.py:322-395: LoFTR self / cross layers and TimeAttnBlock(256)).  The Linear layers run on the implicit-GEMM kernel; these
// kernels are the LayerNorms, the per-pixel T x T temporal attention and the per-(frame, head) linear-attention sums.
#include "common.h"

// ------------------------------------------------------------------------------------------------ time attention core
// One wave per pixel.  tokens = the T frames of that pixel; y = LayerNorm(x) (eps 1e-5, affine); per head (dh = C/heads
// channels): o_t = sum_t2 softmax_t2(y_t . y_t2 / sqrt(dh)) y_t2   (q = k = v = y: the reference never applies qkv).
// Lane l owns channels [l*CPL, (l+1)*CPL), CPL = C/64; a head = 64/heads consecutive lanes.
template <int CPL>
__global__ __launch_bounds__(64) void time_attn_kernel(ppms_sp x, const float* __restrict__ lnw, const float* __restrict__ lnb, ppms_sp out,
                                                       int T, int n, int lanes_per_head, float scale) {
    extern __shared__ float ysh[];                     // [T][64*CPL]
    const int lane = threadIdx.x;
    const int pin = blockIdx.x;
    const int C = 64 * CPL;
    float w[CPL], b[CPL];
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        w[j] = lnw[lane * CPL + j];
        b[j] = lnb[lane * CPL + j];
    }
    // all T frames of the pixel are requested before anything is consumed (the loop used to wait out one memory round trip per frame, with
    // six 2-byte loads per plane and lane): 32-bit loads of channel pairs into LDS, then LayerNorm frame by frame from LDS
    {
        constexpr int PAIRS = CPL / 2;
#pragma unroll 8
        for (int t = 0; t < T; ++t) {
            const int64_t pix = (int64_t)t * n + pin;
            const unsigned* ph = (const unsigned*)((const bf16_t*)x.hi + pix * x.ld + lane * CPL);
            const unsigned* pl = (const unsigned*)((const bf16_t*)x.lo + pix * x.ld + lane * CPL);
#pragma unroll
            for (int j = 0; j < PAIRS; ++j) {
                const unsigned uh = ph[j], ul = pl[j];
                // bf16 -> fp32 is a 16-bit shift: low half = even channel, high half = odd channel
                ysh[t * C + lane * CPL + 2 * j] = __uint_as_float(uh << 16) + __uint_as_float(ul << 16);
                ysh[t * C + lane * CPL + 2 * j + 1] = __uint_as_float(uh & 0xffff0000u) + __uint_as_float(ul & 0xffff0000u);
            }
        }
    }
    for (int t = 0; t < T; ++t) {
        float v[CPL], s = 0.0f;
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            v[j] = ysh[t * C + lane * CPL + j];
            s += v[j];
        }
        for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
        const float mean = s / (float)C;
        float q = 0.0f;
#pragma unroll
        for (int j = 0; j < CPL; ++j) q += (v[j] - mean) * (v[j] - mean);
        for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
        const float rstd = 1.0f / sqrtf(q / (float)C + 1e-5f);
#pragma unroll
        for (int j = 0; j < CPL; ++j) ysh[t * C + lane * CPL + j] = (v[j] - mean) * rstd * w[j] + b[j];
    }
    __syncthreads();
    for (int t1 = 0; t1 < T; ++t1) {
        float y1[CPL], o[CPL];
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            y1[j] = ysh[t1 * C + lane * CPL + j];
            o[j] = 0.0f;
        }
        // two passes over t2 (max, then exp / sum): T is small, the scores are recomputed instead of stored
        float mx = -INFINITY;
        for (int t2 = 0; t2 < T; ++t2) {
            float d = 0.0f;
#pragma unroll
            for (int j = 0; j < CPL; ++j) d += y1[j] * ysh[t2 * C + lane * CPL + j];
            for (int of = 1; of < lanes_per_head; of <<= 1) d += __shfl_xor(d, of);
            mx = fmaxf(mx, d * scale);
        }
        float den = 0.0f;
        for (int t2 = 0; t2 < T; ++t2) {
            float d = 0.0f;
#pragma unroll
            for (int j = 0; j < CPL; ++j) d += y1[j] * ysh[t2 * C + lane * CPL + j];
            for (int of = 1; of < lanes_per_head; of <<= 1) d += __shfl_xor(d, of);
            const float e = expf(d * scale - mx);
            den += e;
#pragma unroll
            for (int j = 0; j < CPL; ++j) o[j] += e * ysh[t2 * C + lane * CPL + j];
        }
        const int64_t pix = (int64_t)t1 * n + pin;
#pragma unroll
        for (int j = 0; j < CPL; ++j) {
            bf16_t hi, lo;
            split_bf16(o[j] / den, hi, lo);
            ((bf16_t*)out.hi)[pix * out.ld + lane * CPL + j] = hi;
            ((bf16_t*)out.lo)[pix * out.ld + lane * CPL + j] = lo;
        }
    }
}

extern "C" int ppms_time_attn(ppms_sp x, const float* ln_w, const float* ln_b, ppms_sp out, int T, int n, int heads, void* stream) {
    PPMS_REQUIRE(x.hi && x.lo && out.hi && out.lo && ln_w && ln_b, "time_attn: null operand");
    PPMS_REQUIRE((x.c == 384 || x.c == 256) && out.c == x.c && heads == 8,
                 "time_attn: C = 384 (update_block16) or 256 (SST block), 8 heads expected, got C=%d heads=%d", x.c, heads);
    PPMS_REQUIRE(T >= 1 && T <= 64 && n >= 1, "time_attn: bad T=%d n=%d", T, n);
    // the kernel reads a lane's channels as 32-bit pairs: rows must start on 4-byte boundaries
    PPMS_REQUIRE(x.ld % 2 == 0 && out.ld % 2 == 0 && ((uintptr_t)x.hi & 3) == 0 && ((uintptr_t)x.lo & 3) == 0,
                 "time_attn: x must be a 4-byte aligned SP view with an even row stride (ld=%d)", x.ld);
    const int lanes_per_head = 64 / heads;
    const float scale = 1.0f / sqrtf((float)(x.c / heads));
    if (x.c == 384)
        hipLaunchKernelGGL(time_attn_kernel<6>, dim3(n), dim3(64), (size_t)T * 384 * 4, (hipStream_t)stream, x, ln_w, ln_b, out, T, n, lanes_per_head,
                           scale);
    else
        hipLaunchKernelGGL(time_attn_kernel<4>, dim3(n), dim3(64), (size_t)T * 256 * 4, (hipStream_t)stream, x, ln_w, ln_b, out, T, n, lanes_per_head,
                           scale);
    return ppms_check_launch("time_attn");
}

// ------------------------------------------------------------------------------------------------ layer norm (+ residual)
// out = resid + LN(x) (resid optional); x: fp32 [pixel][ld], one wave per pixel, C = 64*CPL channels
template <int CPL>
__global__ __launch_bounds__(256) void layernorm_kernel(const float* __restrict__ x, int ld, const float* __restrict__ w, const float* __restrict__ b,
                                                        ppms_sp resid, ppms_sp out, int64_t pixels) {
    const int lane = threadIdx.x & 63;
    const int64_t pix = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (pix >= pixels) return;
    const int C = 64 * CPL;
    float v[CPL], s = 0.0f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        v[j] = x[pix * ld + lane * CPL + j];
        s += v[j];
    }
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    const float mean = s / (float)C;
    float q = 0.0f;
#pragma unroll
    for (int j = 0; j < CPL; ++j) q += (v[j] - mean) * (v[j] - mean);
    for (int o = 32; o > 0; o >>= 1) q += __shfl_xor(q, o);
    const float rstd = 1.0f / sqrtf(q / (float)C + 1e-5f);
#pragma unroll
    for (int j = 0; j < CPL; ++j) {
        const int c = lane * CPL + j;
        float y = (v[j] - mean) * rstd * w[c] + b[c];
        if (resid.hi) y += join_bf16(((const bf16_t*)resid.hi)[pix * resid.ld + c], ((const bf16_t*)resid.lo)[pix * resid.ld + c]);
        bf16_t hi, lo;
        split_bf16(y, hi, lo);
        ((bf16_t*)out.hi)[pix * out.ld + c] = hi;
        ((bf16_t*)out.lo)[pix * out.ld + c] = lo;
    }
}

extern "C" int ppms_layernorm(const float* x, int ld, const float* w, const float* b, ppms_sp resid, ppms_sp out, int64_t pixels, int C,
                              void* stream) {
    PPMS_REQUIRE(x && w && b && out.hi && out.lo && (C == 384 || C == 256) && ld >= C, "layernorm: C = 384 or 256 expected (got %d)", C);
    if (C == 384)
        hipLaunchKernelGGL(layernorm_kernel<6>, dim3(ceil_div(pixels, 4)), dim3(256), 0, (hipStream_t)stream, x, ld, w, b, resid, out, pixels);
    else
        hipLaunchKernelGGL(layernorm_kernel<4>, dim3(ceil_div(pixels, 4)), dim3(256), 0, (hipStream_t)stream, x, ld, w, b, resid, out, pixels);
    return ppms_check_launch("layernorm");
}

// ------------------------------------------------------------------------------------------------ linear attention
// kv[f][hd][d][v] = sum_s K[f,s,hd,d] V[f,s,hd,v];  ksum[f][hd][d] = sum_s K[f,s,hd,d].
// Grid (head, frame, pixel split): each workgroup reduces n/NSPLIT pixels into its own partial (fixed order: deterministic),
// each thread owns a B x B block of the DH x DH outer product, B = DH / 16 (DH = 48: 6 LDS reads per 9 FMAs; DH = 32: 4 per 4).
// Four pixel splits, 64-pixel passes.  More splits shorten this kernel (10 splits: 26 -> 11 us at n = 640) but every workgroup of the apply
// kernel sums all the splits' partial blocks (17 -> 38 us): measured best in total at 4 (round 3; the split count is a function so that the
// workspace size follows it).
constexpr int LA_MAXSPLIT = 32;
static inline int la_nsplit(int n) {
    (void)n;
    return 4;
}
template <int DH>
__global__ __launch_bounds__(256) void linattn_kv_kernel(const float* __restrict__ K, int ldk, const float* __restrict__ V, int ldv,
                                                         float* __restrict__ kv, float* __restrict__ ksum, int n, int heads) {
    // Round 4: a pass stages 160 pixels (the whole share of a workgroup at n = 640) with 16-byte loads, all of them in flight before the
    // first is consumed -- the kernel used to walk 64-pixel passes of 4-byte loads, i.e. three memory round trips in a row (26 -> 13 us at
    // n = 640).  The sums run over the pixels in the same order as before: same bits.
    constexpr int CH = 160, B = DH / 16, Q4 = DH / 4;  // pixels staged per pass; block edge per thread; 16-byte pieces per pixel row
    const int LA_NSPLIT = gridDim.z;
    __shared__ __attribute__((aligned(16))) float ks[CH][DH], vs[CH][DH];
    const int hd = blockIdx.x, f = blockIdx.y, sp_id = blockIdx.z;
    const int tid = threadIdx.x;
    const int d0 = (tid >> 4) * B, v0 = (tid & 15) * B;
    float acc[B][B];
#pragma unroll
    for (int i = 0; i < B; ++i)
#pragma unroll
        for (int j = 0; j < B; ++j) acc[i][j] = 0.0f;
    float ksacc = 0.0f;                                // threads 0..DH-1
    const int per = (n + LA_NSPLIT - 1) / LA_NSPLIT;
    const int s_begin = sp_id * per, s_end = (s_begin + per < n) ? s_begin + per : n;
    for (int s0 = s_begin; s0 < s_end; s0 += CH) {
        const int cnt = (s_end - s0 < CH) ? s_end - s0 : CH;
        constexpr int NIT = (CH * Q4 + 255) / 256;
        f32x4 kr[NIT], vr[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {             // every load of the pass is requested here ...
            const int i = tid + it * 256;
            const int sp = i / Q4, c4 = i - sp * Q4;
            kr[it] = vr[it] = (f32x4){0.0f, 0.0f, 0.0f, 0.0f};
            if (i < CH * Q4 && sp < cnt) {
                const int64_t pix = (int64_t)f * n + s0 + sp;
                kr[it] = gld<f32x4>(K + pix * ldk + hd * DH + c4 * 4);
                vr[it] = gld<f32x4>(V + pix * ldv + hd * DH + c4 * 4);
            }
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {             // ... and written to LDS here
            const int i = tid + it * 256;
            if (i < CH * Q4) {
                const int sp = i / Q4, c4 = i - sp * Q4;
                *(f32x4*)&ks[sp][c4 * 4] = kr[it];
                *(f32x4*)&vs[sp][c4 * 4] = vr[it];
            }
        }
        __syncthreads();
        for (int sp = 0; sp < cnt; ++sp) {
            float kk[B], aa[B];
#pragma unroll
            for (int i = 0; i < B; ++i) {
                kk[i] = ks[sp][d0 + i];
                aa[i] = vs[sp][v0 + i];
            }
#pragma unroll
            for (int i = 0; i < B; ++i)
#pragma unroll
                for (int j = 0; j < B; ++j) acc[i][j] += kk[i] * aa[j];
        }
        if (tid < DH)
            for (int sp = 0; sp < cnt; ++sp) ksacc += ks[sp][tid];
        __syncthreads();
    }
    float* o = kv + (((int64_t)sp_id * gridDim.y + f) * heads + hd) * DH * DH;
#pragma unroll
    for (int i = 0; i < B; ++i)
#pragma unroll
        for (int j = 0; j < B; ++j) o[(d0 + i) * DH + v0 + j] = acc[i][j];
    if (tid < DH) ksum[(((int64_t)sp_id * gridDim.y + f) * heads + hd) * DH + tid] = ksacc;
}

// msg[f,l,hd,v] = (sum_d Q[f,l,hd,d] kv[f,hd,d,v]) * (1 / (Q . ksum + eps)) * n   (one workgroup = 32 pixels of one frame x head)
template <int DH>
__global__ __launch_bounds__(256) void linattn_apply_kernel(const float* __restrict__ Q, int ldq, const float* __restrict__ kv,
                                                            const float* __restrict__ ksum, ppms_sp out, int n, int heads, float eps, int LA_NSPLIT) {
    constexpr int PX = 32, Q4 = DH / 4;
    __shared__ __attribute__((aligned(16))) float kvs[DH][DH], kss[DH], qs[PX][DH];
    const int hd = blockIdx.y, f = blockIdx.z, p0 = blockIdx.x * PX;
    const int tid = threadIdx.x;
    const int64_t pstride_kv = (int64_t)gridDim.z * heads * DH * DH, pstride_ks = (int64_t)gridDim.z * heads * DH;
    const float* kvp = kv + ((int64_t)f * heads + hd) * DH * DH;
    // the splits' partial blocks, 16 bytes per load, summed in split order (as before: same bits)
    for (int i = tid; i < DH * Q4; i += 256) {
        f32x4 a = {0.0f, 0.0f, 0.0f, 0.0f};
        for (int sp = 0; sp < LA_NSPLIT; ++sp) {
            const f32x4 t = gld<f32x4>(kvp + sp * pstride_kv + i * 4);
#pragma unroll
            for (int j = 0; j < 4; ++j) a[j] += t[j];
        }
        *(f32x4*)&kvs[i / Q4][(i % Q4) * 4] = a;
    }
    if (tid < DH) {
        float a = 0.0f;
        for (int sp = 0; sp < LA_NSPLIT; ++sp) a += ksum[sp * pstride_ks + ((int64_t)f * heads + hd) * DH + tid];
        kss[tid] = a;
    }
    for (int i = tid; i < PX * Q4; i += 256) {
        const int sp = i / Q4, c4 = i % Q4;
        f32x4 q = {0.0f, 0.0f, 0.0f, 0.0f};
        if (p0 + sp < n) q = gld<f32x4>(Q + ((int64_t)f * n + p0 + sp) * ldq + hd * DH + c4 * 4);
        *(f32x4*)&qs[sp][c4 * 4] = q;
    }
    __syncthreads();
    for (int i = tid; i < PX * DH; i += 256) {
        const int sp = i / DH, v = i % DH;
        if (p0 + sp >= n) continue;
        float z = 0.0f, a = 0.0f;
        for (int d = 0; d < DH; ++d) {
            z += qs[sp][d] * kss[d];
            a += qs[sp][d] * kvs[d][v];
        }
        const float y = a * (1.0f / (z + eps)) * (float)n;
        bf16_t hi, lo;
        split_bf16(y, hi, lo);
        const int64_t pix = (int64_t)f * n + p0 + sp;
        ((bf16_t*)out.hi)[pix * out.ld + hd * DH + v] = hi;
        ((bf16_t*)out.lo)[pix * out.ld + hd * DH + v] = lo;
    }
}

extern "C" int ppms_linear_attention(const float* Q, int ldq, const float* K, int ldk, const float* V, int ldv, float* kv_ws, ppms_sp out,
                                     int T, int n, int heads, int dh, void* stream) {
    PPMS_REQUIRE(Q && K && V && kv_ws && out.hi && out.lo && heads == 8 && (dh == 48 || dh == 32),
                 "linear_attention: 8 heads x 48 (update_block16) or x 32 (SST block) channels expected");
    // the kernels read rows in 16-byte pieces
    PPMS_REQUIRE(ldq % 4 == 0 && ldk % 4 == 0 && ldv % 4 == 0 && (((uintptr_t)Q | (uintptr_t)K | (uintptr_t)V | (uintptr_t)kv_ws) & 15) == 0,
                 "linear_attention: Q / K / V / workspace must be 16-byte aligned with row strides that are multiples of 4 floats");
    const int nsplit = la_nsplit(n);
    float* kv = kv_ws;
    float* ksum = kv_ws + (size_t)nsplit * T * heads * dh * dh;
    if (dh == 48) {
        hipLaunchKernelGGL(linattn_kv_kernel<48>, dim3(heads, T, nsplit), dim3(256), 0, (hipStream_t)stream, K, ldk, V, ldv, kv, ksum, n, heads);
        hipLaunchKernelGGL(linattn_apply_kernel<48>, dim3(ceil_div(n, 32), heads, T), dim3(256), 0, (hipStream_t)stream, Q, ldq, kv, ksum, out, n, heads,
                           1e-6f, nsplit);
    } else {
        hipLaunchKernelGGL(linattn_kv_kernel<32>, dim3(heads, T, nsplit), dim3(256), 0, (hipStream_t)stream, K, ldk, V, ldv, kv, ksum, n, heads);
        hipLaunchKernelGGL(linattn_apply_kernel<32>, dim3(ceil_div(n, 32), heads, T), dim3(256), 0, (hipStream_t)stream, Q, ldq, kv, ksum, out, n, heads,
                           1e-6f, nsplit);
    }
    return ppms_check_launch("linear_attention");
}

// floats the caller must provide as kv_ws: the pixel splits' partial K^T V blocks and K sums
extern "C" int64_t ppms_linear_attention_workspace_floats(int T, int n, int heads, int dh) {
    return (int64_t)la_nsplit(n) * T * heads * dh * (dh + 1);
}
