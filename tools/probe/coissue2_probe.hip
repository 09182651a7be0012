// One wave per SIMD (or two): does a wave's own VALU work issue in the shadow of its MFMAs on gfx950, and does it matter whether the
// MFMA accumulators live in the VGPR or the AGPR half of the register file?
//   each wave repeats  { MFMA 32x32x16 bf16 ; NV x v_fma_f32 (independent registers) }   8 accumulators / 4 accumulators rotated
// Reported: ns and cycles (at 2.25 GHz) per MFMA slot.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int ACC_A, int NACC>
__global__ __launch_bounds__(256, 1) void probe(float* out, int iters) {
    extern __shared__ char smem[];
    const int lane = threadIdx.x & 63;
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) acc[i] = (f32x16){0};
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (__bf16)(1.0f + lane * 0.001f);
        b[j] = (__bf16)(0.5f);
    }
    float x[16];
    for (int i = 0; i < 16; ++i) x[i] = lane * 0.01f + i;
    const float c = 0.999f, d = 0.0001f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (ACC_A)
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[m % NACC]) : "v"(a), "v"(b));
            else
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m % NACC]) : "v"(a), "v"(b));
#pragma unroll
            for (int k = 0; k < NV; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[(m * NV + k) & 15]) : "v"(c), "v"(d));
        }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
    float s = 0;
    for (int i = 0; i < NACC; ++i) s += acc[i][0] + acc[i][9];
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s + smem[threadIdx.x];
}

template <int NV, int ACC_A, int NACC>
void run(int wg_per_cu) {
    const int blocks = 256 * wg_per_cu;
    float* out;
    hipMalloc(&out, blocks * 256 * 4);
    const int lds = wg_per_cu == 1 ? 100 * 1024 : 60 * 1024;      // 1 or 2 workgroups per CU
    hipFuncSetAttribute((const void*)probe<NV, ACC_A, NACC>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    probe<NV, ACC_A, NACC><<<blocks, 256, lds>>>(out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<NV, ACC_A, NACC><<<blocks, 256, lds>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / ((double)iters * 8);
    printf("waves/SIMD %d  acc in %s x%d  NV=%d : %.2f ns per MFMA slot = %.1f cycles @2.25GHz\n", wg_per_cu, ACC_A ? "AGPR" : "VGPR", NACC, NV, ns, ns * 2.25);
    hipFree(out);
}

int main() {
    run<0, 0, 8>(1); run<1, 0, 8>(1); run<2, 0, 8>(1); run<4, 0, 8>(1); run<6, 0, 8>(1); run<8, 0, 8>(1);
    run<0, 1, 8>(1); run<1, 1, 8>(1); run<2, 1, 8>(1); run<4, 1, 8>(1); run<6, 1, 8>(1); run<8, 1, 8>(1);
    run<4, 0, 4>(1); run<4, 1, 4>(1); run<4, 0, 2>(1); run<4, 1, 2>(1);
    run<0, 0, 4>(2); run<4, 0, 4>(2); run<4, 1, 4>(2); run<6, 1, 4>(2);
    return 0;
}
