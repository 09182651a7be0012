// What keeps a wave's VALU work from hiding under its MFMAs?  One wave per SIMD; a slot = one MFMA 32x32x16 bf16 plus:
//   F: NV v_fma_f32 on private registers    E: one v_exp_f32    R: the fmas READ registers of a VGPR tile written by earlier MFMAs
//   S: half of the MFMAs accumulate into VGPR tiles (4 tuples), the other half into AGPR tiles (8 tuples)
//   L: one ds_read_b128 every second slot, waited for two reads later (s_waitcnt lgkmcnt(2))
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int NV, bool E, bool R, bool S, bool L>
__global__ __launch_bounds__(256, 1) void probe(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63;
    f32x16 acc[8], st[4];
    for (int i = 0; i < 8; ++i) acc[i] = (f32x16){0};
    for (int i = 0; i < 4; ++i) st[i] = (f32x16){0};
    bf16x8 a, b;
    for (int j = 0; j < 8; ++j) {
        a[j] = (__bf16)(1.0f + lane * 0.001f);
        b[j] = (__bf16)(0.5f);
    }
    u32x4 fr[4] = {};
    const unsigned la = (unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)smem + threadIdx.x * 16;
    float x[16], e[4], t[4];
    for (int i = 0; i < 16; ++i) x[i] = lane * 0.01f + i;
    for (int i = 0; i < 4; ++i) e[i] = -0.001f * lane - i, t[i] = 0;
    const float c = 0.999f, d = 0.0001f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < 16; ++m) {
            if (L && (m & 1) == 0) asm volatile("s_waitcnt lgkmcnt(2)" ::: "memory");
            if (S && (m & 1))
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+v"(st[(m >> 1) & 3]) : "v"(L ? __builtin_bit_cast(bf16x8, fr[(m >> 1) & 3]) : a), "v"(b));
            else
                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[m & 7]) : "v"(L ? __builtin_bit_cast(bf16x8, fr[(m >> 1) & 3]) : a), "v"(b));
            if (L && (m & 1)) asm volatile("ds_read_b128 %0, %1 offset:%2" : "+v"(fr[((m >> 1) + 3) & 3]) : "v"(la), "n"(m * 4096));
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                if (R)
                    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(t[k & 3]) : "v"(st[(m + 2) & 3][(k * 5 + m) & 15]), "v"(c), "v"(d));
                else
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[(m * NV + k) & 15]) : "v"(c), "v"(d));
            }
            if (E) asm volatile("v_exp_f32 %0, %0" : "+v"(e[m & 3]));
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_nop 15\n\ts_nop 15" ::: "memory");
    float s = 0;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][9];
    for (int i = 0; i < 4; ++i) s += st[i][3] + e[i] + t[i] + fr[i][0];
    for (int i = 0; i < 16; ++i) s += x[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV, bool E, bool R, bool S, bool L>
void run() {
    const int blocks = 256;
    float* out;
    (void)hipMalloc(&out, blocks * 256 * 4);
    const int lds = 100 * 1024;
    (void)hipFuncSetAttribute((const void*)probe<NV, E, R, S, L>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const int iters = 10000;
    probe<NV, E, R, S, L><<<blocks, 256, lds>>>(out, 10);
    (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0);
    probe<NV, E, R, S, L><<<blocks, 256, lds>>>(out, iters);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    const double ns = ms * 1e6 / ((double)iters * 16);
    printf("NV=%d exp=%d read-MFMA-tile=%d VGPR-S-tiles=%d lds=%d : %.2f ns per slot = %.1f cycles @2.25GHz\n", NV, E, R, S, L, ns, ns * 2.25);
    (void)hipFree(out);
}

int main() {
    run<3, false, false, false, false>();
    run<3, true, false, false, false>();
    run<2, true, false, false, false>();
    run<3, false, false, true, false>();
    run<3, false, true, true, false>();
    run<3, true, true, true, false>();
    run<3, false, false, false, true>();
    run<3, true, true, true, true>();
    run<2, true, true, true, true>();
    run<0, false, false, true, false>();
    run<0, false, false, false, true>();
    return 0;
}
