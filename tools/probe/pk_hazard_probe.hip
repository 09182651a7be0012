// Minimal reproducer attempt for the packed-fp32 finding (docs/LOG_r01_r05.md section 5): does a wave's v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32
// return wrong values while ANOTHER stream's MFMA-streaming kernel shares its SIMD?
//   kernel A (victim): every thread evaluates a short chain of packed fp32 ops (or the same math with scalar ops, VAR = 0) on values
//                      derived from its index and writes the result;
//   kernel B (aggressor): streams v_mfma_f32_32x32x16_bf16 (optionally with LDS traffic) for a few hundred microseconds.
// A runs alone first (reference), then repeatedly while B is resident; mismatching elements are counted per lane quarter.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int VAR>
__global__ __launch_bounds__(256) void victim(const float* __restrict__ src, float* __restrict__ dst, long n, int rounds) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const float a = src[idx], b = src[(idx * 7 + 3) % n], c = src[(idx * 13 + 5) % n], d = src[(idx * 29 + 11) % n];
    f32x2 x = {a, b}, y = {c, d}, w = {0.25f, 0.75f}, acc = {0.0f, 0.0f};
    for (int r = 0; r < rounds; ++r) {
        if (VAR == 1) {
            asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(x) : "v"(x), "v"(w));
            asm volatile("v_pk_fma_f32 %0, %1, %2, %0" : "+v"(acc) : "v"(x), "v"(y));
            asm volatile("v_pk_add_f32 %0, %1, %2" : "=v"(y) : "v"(y), "v"(w));
        } else {
            x = x * w;
            acc = x * y + acc;
            y = y + w;
        }
    }
    dst[idx] = acc[0] + acc[1];
}

// the failing kernel's shape: four independent gather loads, consumed pairwise by packed ops right behind the counted vmcnt waits
template <int VAR>
__global__ __launch_bounds__(256) void victim_gather(const float* __restrict__ src, float* __restrict__ dst, int W, long n) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const int x = (int)(idx % W);
    const long row = idx / W;
    const int x1 = x + (x < W - 1 ? 1 : 0);
    const long r1 = row + ((row + 1) * W < n ? 1 : 0);
    const float fx = 0.25f + 0.5f * (float)(x & 1), fy = 0.75f - 0.5f * (float)(row & 1);
    const float s00 = src[row * W + x], s01 = src[row * W + x1], s10 = src[r1 * W + x], s11 = src[r1 * W + x1];
    float out;
    if (VAR == 1) {
        f32x2 top = {s00, s01}, bot = {s10, s11}, wx = {1.0f - fx, fx}, t, b2;
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(t) : "v"(top), "v"(wx));
        asm volatile("v_pk_mul_f32 %0, %1, %2" : "=v"(b2) : "v"(bot), "v"(wx));
        out = (1.0f - fy) * (t[0] + t[1]) + fy * (b2[0] + b2[1]);
    } else {
        out = (1.0f - fy) * ((1.0f - fx) * s00 + fx * s01) + fy * ((1.0f - fx) * s10 + fx * s11);
    }
    dst[idx] = out;
}

// aggressor 2: LDS-DMA streaming (global_load_lds_dwordx4, the conv kernels' window gather), optionally with MFMAs
template <int MFMA>
__global__ __launch_bounds__(256, 2) void aggressor_dma(const float* __restrict__ src, float* out, int iters, long nsrc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    f32x16 acc = (f32x16){0};
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) a[j] = (__bf16)(1.0f + lane * 0.001f), b[j] = (__bf16)(0.5f);
    for (int it = 0; it < iters; ++it) {
        const char* g = (const char*)src + (((long)blockIdx.x * 4096 + it * 65536 + threadIdx.x * 16) % (nsrc * 4 - 65536));
        char* l = smem + wave * 1024 + (it & 7) * 4096;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(g + k * 4096), (__attribute__((address_space(3))) void*)(l), 16, 0, 0);
        if (MFMA)
            for (int m = 0; m < 4; ++m) acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc, 0, 0, 0);
        if ((it & 7) == 7) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    out[blockIdx.x * 256 + threadIdx.x] = acc[0] + ((float*)smem)[threadIdx.x];
}

template <int LDS>
__global__ __launch_bounds__(256, 2) void aggressor(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    if (LDS)
        for (int i = threadIdx.x; i < 8192; i += 256) ((int*)smem)[i] = 0x3f803f80 + i;
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f32x16){0};
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) a[j] = (__bf16)(1.0f + lane * 0.001f), b[j] = (__bf16)(0.5f);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (LDS) a = *(const bf16x8*)(smem + ((lane * 16 + m * 1024 + it * 64) & 32752));
            acc[m & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[m & 3], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int VAR, int LDS>
void run(const char* name) {
    const long n = 1 << 22;
    float *src, *dst, *ref, *sink;
    (void)hipMalloc(&src, n * 4);
    (void)hipMalloc(&dst, n * 4);
    (void)hipMalloc(&ref, n * 4);
    (void)hipMalloc(&sink, 2048 * 256 * 4);
    std::vector<float> h(n);
    for (long i = 0; i < n; ++i) h[i] = 0.5f + (float)((i * 2654435761u) & 0xffff) / 65536.0f;
    (void)hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice);
    hipStream_t s0, s1;
    (void)hipStreamCreate(&s0);
    (void)hipStreamCreate(&s1);
    (void)hipFuncSetAttribute((const void*)aggressor<LDS>, hipFuncAttributeMaxDynamicSharedMemorySize, 32768);
    victim<VAR><<<(n + 255) / 256, 256, 0, s0>>>(src, ref, n, 8);
    (void)hipDeviceSynchronize();
    std::vector<float> r(n), o(n);
    (void)hipMemcpy(r.data(), ref, n * 4, hipMemcpyDeviceToHost);
    long bad = 0, q[4] = {0, 0, 0, 0};
    int bad_runs = 0;
    for (int rep = 0; rep < 60; ++rep) {
        aggressor<LDS><<<1024, 256, 32768, s1>>>(sink, 3000);           // ~0.5 ms of MFMA streaming on every CU, 2 workgroups per CU
        victim<VAR><<<(n + 255) / 256, 256, 0, s0>>>(src, dst, n, 8);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(o.data(), dst, n * 4, hipMemcpyDeviceToHost);
        long b = 0;
        for (long i = 0; i < n; ++i)
            if (o[i] != r[i]) ++b, ++q[(i & 63) >> 4];
        bad += b;
        bad_runs += b != 0;
    }
    printf("%-34s: %d of 60 runs differ, %ld elements; by lane quarter (0-15, 16-31, 32-47, 48-63): %ld %ld %ld %ld\n", name, bad_runs, bad, q[0], q[1], q[2],
           q[3]);
    (void)hipFree(src); (void)hipFree(dst); (void)hipFree(ref); (void)hipFree(sink);
}

template <int VAR, int MFMA>
void run_gather(const char* name) {
    const long n = 1 << 22;
    const int W = 2048;
    float *src, *dst, *ref, *sink, *big;
    const long nbig = 1 << 26;
    (void)hipMalloc(&src, n * 4);
    (void)hipMalloc(&dst, n * 4);
    (void)hipMalloc(&ref, n * 4);
    (void)hipMalloc(&sink, 2048 * 256 * 4);
    (void)hipMalloc(&big, nbig * 4);
    (void)hipMemset(big, 0, nbig * 4);
    std::vector<float> h(n);
    for (long i = 0; i < n; ++i) h[i] = 0.5f + (float)((i * 2654435761u) & 0xffff) / 65536.0f;
    (void)hipMemcpy(src, h.data(), n * 4, hipMemcpyHostToDevice);
    hipStream_t s0, s1;
    (void)hipStreamCreate(&s0);
    (void)hipStreamCreate(&s1);
    (void)hipFuncSetAttribute((const void*)aggressor_dma<MFMA>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    victim_gather<VAR><<<(n + 255) / 256, 256, 0, s0>>>(src, ref, W, n);
    (void)hipDeviceSynchronize();
    std::vector<float> r(n), o(n);
    (void)hipMemcpy(r.data(), ref, n * 4, hipMemcpyDeviceToHost);
    long bad = 0, q[4] = {0, 0, 0, 0};
    int bad_runs = 0;
    for (int rep = 0; rep < 60; ++rep) {
        aggressor_dma<MFMA><<<512, 256, 65536, s1>>>(big, sink, 4000, nbig);
        victim_gather<VAR><<<(n + 255) / 256, 256, 0, s0>>>(src, dst, W, n);
        (void)hipDeviceSynchronize();
        (void)hipMemcpy(o.data(), dst, n * 4, hipMemcpyDeviceToHost);
        long b = 0;
        for (long i = 0; i < n; ++i)
            if (o[i] != r[i]) ++b, ++q[(i & 63) >> 4];
        bad += b;
        bad_runs += b != 0;
    }
    printf("%-44s: %d of 60 runs differ, %ld elements; by lane quarter: %ld %ld %ld %ld\n", name, bad_runs, bad, q[0], q[1], q[2], q[3]);
    (void)hipFree(src); (void)hipFree(dst); (void)hipFree(ref); (void)hipFree(sink); (void)hipFree(big);
}

int main() {
    run_gather<0, 0>("gather + scalar fp32, LDS-DMA aggressor");
    run_gather<1, 0>("gather + packed fp32, LDS-DMA aggressor");
    run_gather<1, 1>("gather + packed fp32, LDS-DMA + MFMA aggressor");
    run_gather<0, 1>("gather + scalar fp32, LDS-DMA + MFMA aggressor");
    run<0, 0>("scalar fp32, MFMA aggressor");
    run<1, 0>("packed fp32, MFMA aggressor");
    run<1, 1>("packed fp32, MFMA + LDS aggressor");
    run<0, 1>("scalar fp32, MFMA + LDS aggressor");
    return 0;
}
