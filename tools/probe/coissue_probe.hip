// Does the VALU (incl. quarter-rate v_exp_f32) run under the shadow of a bf16 MFMA on gfx950?
//  same-wave:  each wave issues  { MFMA 32x32x16 ; NV x v_fma_f32 ; NE x v_exp_f32 }  repeatedly (independent registers)
//  two-wave:   workgroups alternate roles (even: MFMA only, odd: VALU only), two workgroups per CU share each SIMD
// Reported: cycles per loop trip at the measured SCLK-agnostic wall time (ns per trip per wave).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NV, int NE, int ROLE>   // ROLE 0: everything in one wave; 1: even blocks MFMA, odd blocks VALU
__global__ __launch_bounds__(256, 2) void probe(float* out, int iters) {
    const int lane = threadIdx.x & 63;
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f32x16){0};
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (__bf16)(1.0f + lane * 0.001f);
        b[j] = (__bf16)(0.5f);
    }
    float x[16], e[4];
    for (int i = 0; i < 16; ++i) x[i] = lane * 0.01f + i;
    for (int i = 0; i < 4; ++i) e[i] = -0.001f * lane - i;
    const float c = 0.999f, d = 0.0001f;
    const bool do_m = ROLE == 0 || (blockIdx.x & 1) == 0;
    const bool do_v = ROLE == 0 || (blockIdx.x & 1) == 1;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            if (do_m) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(acc[m & 3]) : "v"(a), "v"(b));
            if (do_v) {
#pragma unroll
                for (int k = 0; k < NV; ++k) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[(m * NV + k) & 15]) : "v"(c), "v"(d));
#pragma unroll
                for (int k = 0; k < NE; ++k) asm volatile("v_exp_f32 %0, %0" : "+v"(e[(m * NE + k) & 3]));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][9] + e[i];
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV, int NE, int ROLE>
void run(int blocks) {
    float* out;
    hipMalloc(&out, blocks * 256 * 4);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 20000;
    probe<NV, NE, ROLE><<<blocks, 256>>>(out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<NV, NE, ROLE><<<blocks, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double ns_per_mfma_slot = ms * 1e6 / ((double)iters * 8);
    printf("role=%d blocks=%4d  per MFMA: %2d v_fma + %d v_exp   -> %7.2f ns per {MFMA + VALU group}   (%.3f ms)\n", ROLE, blocks, NV, NE,
           ns_per_mfma_slot, ms);
    hipFree(out);
}

int main() {
    for (int blocks : {256, 512}) {
        run<0, 0, 0>(blocks);
        run<4, 0, 0>(blocks);
        run<6, 0, 0>(blocks);
        run<8, 0, 0>(blocks);
        run<12, 0, 0>(blocks);
        run<16, 0, 0>(blocks);
        run<0, 1, 0>(blocks);
        run<0, 2, 0>(blocks);
        run<0, 4, 0>(blocks);
        run<4, 1, 0>(blocks);
    }
    // two workgroups per CU with different roles: one wave per SIMD does MFMA, the other VALU
    run<8, 0, 1>(512);
    run<16, 0, 1>(512);
    run<0, 2, 1>(512);
    run<0, 4, 1>(512);
    run<8, 2, 1>(512);
    return 0;
}
