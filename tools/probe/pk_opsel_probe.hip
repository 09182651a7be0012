// Minimal form of the packed-fp32 finding (docs/LOG_r01_r05.md section 5; tools/probe/bilinear_pk_probe.hip narrowed it to the swizzled packed add):
// a register-only victim -- no loads -- evaluates one VOP3P fp32 instruction per round on values derived from the thread index and checks
// it against scalar arithmetic.  It keeps to < 24 VGPRs so that one of its waves still fits on a SIMD beside two 240-register waves of the
// library's conv kernels (the high-register probes of earlier rounds never shared a SIMD with them, which is why they saw nothing).
//   OP 0: v_pk_add_f32 op_sel:[0,1] op_sel_hi:[1,0]   (lo = x.lo + y.hi, hi = x.hi + y.lo: the failing instruction)
//   OP 1: v_pk_add_f32                                 (no swizzle)
//   OP 2: v_pk_mul_f32 op_sel:[0,1] op_sel_hi:[1,0]
//   OP 3: v_pk_fma_f32 op_sel:[0,1,0] op_sel_hi:[1,0,1]
//   OP 4: v_pk_add_f32 op_sel:[1,1] op_sel_hi:[0,0]   (both sources swapped)
//   OP 5: v_pk_add_f32 op_sel:[0,0] op_sel_hi:[0,0]   (lo halves broadcast)
//   OP 6: v_pk_mov_b32 op_sel:[1,0]                    (the swizzle alone: lo = x.hi, hi = y.lo)
//   OP 7-10: v_pk_add_f32 with op_sel:[1,0] op_sel_hi:[0,1] / [0,1],[0,1] / [0,0],[1,0] / [0,1],[1,1]  (which result, which source)
// Synthetic aggressors for the other stream: MFMA only, LDS reads only, global loads only, VALU only.
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int OP>
__global__ __launch_bounds__(256) void opsel_victim(unsigned* __restrict__ bad, float* __restrict__ first, int rounds) {
    const int idx = blockIdx.x * 256 + threadIdx.x;
    f32x2 x = {1.0f + (float)(idx & 1023) * 0.0009765625f, 2.0f + (float)(idx & 511) * 0.001953125f};
    f32x2 y = {0.5f + (float)(idx & 255) * 0.00390625f, 4.0f + (float)(idx & 127) * 0.0078125f};
    const f32x2 one = {1.0f, 1.0f};
    unsigned nbad = 0;
    for (int r = 0; r < rounds; ++r) {
        f32x2 z, e;
        // the reference is formed by scalar instructions written out here: left to the compiler it becomes the same kind of swizzled packed op
#define SC2(INS, D, A, B) asm volatile(INS " %0, %1, %2" : "=v"(D) : "v"(A), "v"(B))
        if (OP == 0) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(z) : "v"(x), "v"(y)); SC2("v_add_f32", e[0], x[0], y[1]); SC2("v_add_f32", e[1], x[1], y[0]); }
        if (OP == 1) { asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(z) : "v"(x), "v"(y)); SC2("v_add_f32", e[0], x[0], y[0]); SC2("v_add_f32", e[1], x[1], y[1]); }
        if (OP == 2) { asm volatile("v_pk_mul_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,0]" : "=v"(z) : "v"(x), "v"(y)); SC2("v_mul_f32", e[0], x[0], y[1]); SC2("v_mul_f32", e[1], x[1], y[0]); }
        if (OP == 3) {
            asm volatile("v_pk_fma_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[1,0,1]" : "=v"(z) : "v"(x), "v"(y), "v"(one));
            asm volatile("v_fma_f32 %0, %1, %2, 1.0" : "=v"(e[0]) : "v"(x[0]), "v"(y[1]));
            asm volatile("v_fma_f32 %0, %1, %2, 1.0" : "=v"(e[1]) : "v"(x[1]), "v"(y[0]));
        }
        if (OP == 4) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,1] op_sel_hi:[0,0]" : "=v"(z) : "v"(x), "v"(y)); SC2("v_add_f32", e[0], x[1], y[1]); SC2("v_add_f32", e[1], x[0], y[0]); }
        if (OP == 5) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[0,0]" : "=v"(z) : "v"(x), "v"(y)); SC2("v_add_f32", e[0], x[0], y[0]); SC2("v_add_f32", e[1], x[0], y[0]); }
        if (OP == 6) { asm volatile("v_pk_mov_b32 %0, %1, %2 op_sel:[1,0]" : "=v"(z) : "v"(x), "v"(y)); e = (f32x2){x[1], y[0]}; }
        if (OP == 7) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[1,0] op_sel_hi:[0,1]" : "=v"(z) : "v"(x), "v"(y)); SC2("v_add_f32", e[0], x[1], y[0]); SC2("v_add_f32", e[1], x[0], y[1]); }
        if (OP == 8) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[0,1]" : "=v"(z) : "v"(x), "v"(y)); SC2("v_add_f32", e[0], x[0], y[1]); SC2("v_add_f32", e[1], x[0], y[1]); }
        if (OP == 9) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0]" : "=v"(z) : "v"(x), "v"(y)); SC2("v_add_f32", e[0], x[0], y[0]); SC2("v_add_f32", e[1], x[1], y[0]); }
        if (OP == 10) { asm volatile("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1]" : "=v"(z) : "v"(x), "v"(y)); SC2("v_add_f32", e[0], x[0], y[1]); SC2("v_add_f32", e[1], x[1], y[1]); }
        asm volatile("" : "+v"(z));
        if (z[0] != e[0] || z[1] != e[1]) {
            if (nbad == 0 && first != nullptr && atomicAdd(bad + 4, 1u) == 0) {
                first[0] = x[0], first[1] = x[1], first[2] = y[0], first[3] = y[1], first[4] = z[0], first[5] = z[1], first[6] = e[0], first[7] = e[1];
                first[8] = (float)(threadIdx.x & 63);
            }
            ++nbad;
        }
        x[0] += 0.25f, x[1] -= 0.125f, y[0] += 0.0625f, y[1] -= 0.5f;
    }
    if (nbad) atomicAdd(bad + ((threadIdx.x & 63) >> 4), nbad);
}

// ---- aggressors -------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void agg_mfma(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[2] = {(f32x16){0}, (f32x16){0}};
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) a[j] = (__bf16)(1.0f + lane * 0.001f), b[j] = (__bf16)(0.5f);
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int m = 0; m < 8; ++m) acc[m & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 1], 0, 0, 0);
    out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][7];
}
__global__ __launch_bounds__(256) void agg_lds(float* out, int iters) {
    __shared__ __attribute__((aligned(16))) float sm[8192];
    for (int i = threadIdx.x; i < 8192; i += 256) sm[i] = (float)i;
    __syncthreads();
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 s = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int m = 0; m < 8; ++m) s += *(const f32x4*)(sm + (((threadIdx.x + m * 256 + it * 64) * 4) & 8188));
    out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
__global__ __launch_bounds__(256) void agg_gload(const float* __restrict__ src, float* out, int iters, long nsrc) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    f32x4 s = {0, 0, 0, 0};
    long o = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    for (int it = 0; it < iters; ++it) {
        s += *(const f32x4*)(src + (o % (nsrc - 4)));
        o += 1 << 20;
    }
    out[blockIdx.x * 256 + threadIdx.x] = s[0] + s[1] + s[2] + s[3];
}
__global__ __launch_bounds__(256) void agg_valu(float* out, int iters) {
    float a = threadIdx.x * 0.001f, b = 1.0001f, c = 0.5f, d = 0.25f;
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int m = 0; m < 8; ++m) a = __builtin_fmaf(a, b, c), d = __builtin_fmaf(d, b, a);
    out[blockIdx.x * 256 + threadIdx.x] = a + d;
}

// "holder": does nothing but OCCUPY registers -- 8 waves per workgroup (2 per SIMD) of R VGPRs each, one workgroup per CU (100 KiB of LDS pins
// that), sleeping for `ticks` of the 100 MHz wall clock.  A victim wave that lands beside them gets its VGPRs at physical offset 2 * R of the SIMD's
// 512-register file: is the fault about WHERE the victim's registers are rather than what the neighbour executes?
template <int R>
__global__ __launch_bounds__(512) void holder_kernel(float* out, long ticks) {
    extern __shared__ float hold[];
    if (R == 64) asm volatile("v_mov_b32 v63, 0" ::: "v63");
    if (R == 128) asm volatile("v_mov_b32 v127, 0" ::: "v127");
    if (R == 192) asm volatile("v_mov_b32 v191, 0" ::: "v191");
    if (R == 224) asm volatile("v_mov_b32 v223, 0" ::: "v223");
    if (R == 232) asm volatile("v_mov_b32 v231, 0" ::: "v231");
    if (R == 240) asm volatile("v_mov_b32 v239, 0" ::: "v239");
    const long t0 = wall_clock64();
    while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(20);
    if (ticks < 0) out[threadIdx.x] = hold[threadIdx.x];
}
template <int R>
static void launch_holder(float* sink, long ticks, hipStream_t st) {
    (void)hipFuncSetAttribute((const void*)holder_kernel<R>, hipFuncAttributeMaxDynamicSharedMemorySize, 100 * 1024);
    hipLaunchKernelGGL(holder_kernel<R>, dim3(256), dim3(512), 100 * 1024, st, sink, ticks);
}
extern "C" int opsel_holder_launch(int regs, float* sink, long ticks, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    switch (regs) {
        case 64: launch_holder<64>(sink, ticks, st); break;
        case 128: launch_holder<128>(sink, ticks, st); break;
        case 192: launch_holder<192>(sink, ticks, st); break;
        case 224: launch_holder<224>(sink, ticks, st); break;
        case 232: launch_holder<232>(sink, ticks, st); break;
        case 240: launch_holder<240>(sink, ticks, st); break;
        default: return -1;
    }
    return (int)hipGetLastError();
}

// composite aggressor, the conv kernels' ingredients one flag each: 1 global loads, 2 LDS writes, 4 barriers, 8 LDS reads, 16 MFMA,
// 32 fp32 -> bf16 hi/lo split (v_cvt_pk_bf16_f32 and friends), 64 global stores
template <int F>
__global__ __launch_bounds__(256) void agg_mix(const float* __restrict__ src, float* __restrict__ out, int iters, long nsrc) {
    typedef float f32x4 __attribute__((ext_vector_type(4)));
    __shared__ __attribute__((aligned(16))) float sm[8192];
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += 256) sm[i] = (float)(i & 7);
    __syncthreads();
    f32x16 acc = (f32x16){0};
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) a[j] = (__bf16)(1.0f + lane * 0.001f), b[j] = (__bf16)(0.5f);
    f32x4 g = {1.0f, 2.0f, 3.0f, 4.0f}, l = {0.5f, 0.25f, 0.125f, 1.0f};
    long o = ((long)blockIdx.x * 256 + threadIdx.x) * 4;
    for (int it = 0; it < iters; ++it) {
        if (F & 1) {
            g = *(const f32x4*)(src + (o % (nsrc - 4)));
            o += 1 << 18;
        }
        if (F & 2) *(f32x4*)(sm + ((threadIdx.x * 4 + it * 1024) & 8188)) = g + l;
        if (F & 4) __syncthreads();
        if (F & 8) l = *(const f32x4*)(sm + ((threadIdx.x * 4 + 2048 + it * 1024) & 8188));
        if (F & 32) {
            // hi = bf16(v), lo = bf16(v - hi): the SP split of the conv epilogues, feeding the MFMA operands
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const __bf16 hi = (__bf16)l[j];
                const __bf16 lo = (__bf16)(l[j] - (float)hi);
                a[2 * j] = hi, a[2 * j + 1] = lo;
            }
        }
        if (F & 16)
#pragma unroll
            for (int mm = 0; mm < 4; ++mm) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        if (F & 64) *(f32x4*)(out + (((long)blockIdx.x * 256 + threadIdx.x) * 4 + (long)(it & 15) * (1 << 22))) = (f32x4){acc[0], acc[1], l[0], g[1]};
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[5] + l[0] + g[0] + (float)a[3];
}
extern "C" int opsel_mix_launch(int flags, const float* src, long nsrc, float* sink, int nblocks, int iters, void* stream) {
    hipStream_t st = (hipStream_t)stream;
#define MIX(FL) case FL: hipLaunchKernelGGL(agg_mix<FL>, dim3(nblocks), dim3(256), 0, st, src, sink, iters, nsrc); break;
    switch (flags) {
        MIX(127) MIX(126) MIX(125) MIX(123) MIX(119) MIX(111) MIX(95) MIX(63) MIX(1) MIX(2) MIX(6) MIX(4) MIX(8) MIX(16) MIX(32) MIX(48) MIX(64) MIX(24) MIX(30)
        default: return -1;
    }
#undef MIX
    return (int)hipGetLastError();
}

extern "C" int opsel_victim_launch(int op, unsigned* bad, float* first, int nblocks, int rounds, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    switch (op) {
        case 0: hipLaunchKernelGGL(opsel_victim<0>, dim3(nblocks), dim3(256), 0, st, bad, first, rounds); break;
        case 1: hipLaunchKernelGGL(opsel_victim<1>, dim3(nblocks), dim3(256), 0, st, bad, first, rounds); break;
        case 2: hipLaunchKernelGGL(opsel_victim<2>, dim3(nblocks), dim3(256), 0, st, bad, first, rounds); break;
        case 3: hipLaunchKernelGGL(opsel_victim<3>, dim3(nblocks), dim3(256), 0, st, bad, first, rounds); break;
        case 4: hipLaunchKernelGGL(opsel_victim<4>, dim3(nblocks), dim3(256), 0, st, bad, first, rounds); break;
        case 5: hipLaunchKernelGGL(opsel_victim<5>, dim3(nblocks), dim3(256), 0, st, bad, first, rounds); break;
        case 6: hipLaunchKernelGGL(opsel_victim<6>, dim3(nblocks), dim3(256), 0, st, bad, first, rounds); break;
        case 7: hipLaunchKernelGGL(opsel_victim<7>, dim3(nblocks), dim3(256), 0, st, bad, first, rounds); break;
        case 8: hipLaunchKernelGGL(opsel_victim<8>, dim3(nblocks), dim3(256), 0, st, bad, first, rounds); break;
        case 9: hipLaunchKernelGGL(opsel_victim<9>, dim3(nblocks), dim3(256), 0, st, bad, first, rounds); break;
        case 10: hipLaunchKernelGGL(opsel_victim<10>, dim3(nblocks), dim3(256), 0, st, bad, first, rounds); break;
        default: return -1;
    }
    return (int)hipGetLastError();
}
// kind 0: MFMA, 1: LDS reads, 2: global loads (src: nsrc floats), 3: VALU
extern "C" int opsel_aggressor_launch(int kind, const float* src, long nsrc, float* sink, int nblocks, int iters, void* stream) {
    hipStream_t st = (hipStream_t)stream;
    if (kind == 0) hipLaunchKernelGGL(agg_mfma, dim3(nblocks), dim3(256), 0, st, sink, iters);
    else if (kind == 1) hipLaunchKernelGGL(agg_lds, dim3(nblocks), dim3(256), 0, st, sink, iters);
    else if (kind == 2) hipLaunchKernelGGL(agg_gload, dim3(nblocks), dim3(256), 0, st, src, sink, iters, nsrc);
    else hipLaunchKernelGGL(agg_valu, dim3(nblocks), dim3(256), 0, st, sink, iters);
    return (int)hipGetLastError();
}
