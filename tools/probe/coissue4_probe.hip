// One wave per SIMD: what does a wave's own VALU work cost behind 16x16x32 MFMAs (16 cycles each), by instruction kind?
//   each wave repeats  { v_mfma_f32_16x16x32_bf16 (accumulators in AGPRs) ; NV x <op> on independent registers }
// op: 0 = v_fma_f32, 1 = v_exp_f32, 2 = v_cvt_pk_bf16_f32, 3 = one v_exp_f32 + (NV - 1) v_fma_f32.  Zero operands (cycles, not clocks).
// Reported: shader cycles (s_memtime) per MFMA slot.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NV, int OP>
__global__ __launch_bounds__(256, 1) void probe(float* out, long long* cyc, int iters) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63;
    f32x4 acc[16];
    for (int i = 0; i < 16; ++i) acc[i] = (f32x4){0};
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (__bf16)0.0f;
        b[j] = (__bf16)0.0f;
    }
    float x[16], y[16];
    for (int i = 0; i < 16; ++i) x[i] = lane * 0.01f + i, y[i] = 0;
    const float c = 0.999f, d = 0.0001f;
    const long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[m]) : "v"(a), "v"(b));
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                const int r = (m * NV + k) & 15;
                if (OP == 0 || (OP == 3 && k > 0)) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[r]) : "v"(c), "v"(d));
                if (OP == 1 || (OP == 3 && k == 0)) asm volatile("v_exp_f32 %0, %1" : "=v"(y[r]) : "v"(x[r]));
                if (OP == 2) asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(y[r]) : "v"(x[r]), "v"(x[(r + 1) & 15]));
            }
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < 16; ++i) s += acc[i][0] + x[i] + y[i];
    out[blockIdx.x * 256 + threadIdx.x] = s + smem[threadIdx.x];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NV, int OP>
void run() {
    const int blocks = 256, iters = 20000, lds = 100 * 1024;
    float* out;
    long long* cyc;
    (void)hipMalloc(&out, blocks * 256 * 4);
    (void)hipMalloc(&cyc, blocks * 8);
    (void)hipFuncSetAttribute((const void*)probe<NV, OP>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    probe<NV, OP><<<blocks, 256, lds>>>(out, cyc, 10);
    probe<NV, OP><<<blocks, 256, lds>>>(out, cyc, iters);
    (void)hipDeviceSynchronize();
    long long h[256];
    (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < blocks; ++i) s += (double)h[i];
    static const char* nm[] = {"v_fma_f32", "v_exp_f32", "v_cvt_pk_bf16_f32", "1 exp + fma"};
    printf("MFMA 16x16x32 + %d x %-18s: %6.2f cycles per MFMA slot\n", NV, nm[OP], s / blocks / iters / 16);
    (void)hipFree(out);
    (void)hipFree(cyc);
}

int main() {
    run<0, 0>();
    run<1, 0>(); run<2, 0>(); run<3, 0>(); run<4, 0>();
    run<1, 1>(); run<2, 1>();
    run<1, 2>(); run<2, 2>();
    run<2, 3>(); run<3, 3>();
    return 0;
}
