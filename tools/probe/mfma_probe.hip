// Micro-benchmarks that calibrate what the conv / attention inner loops can reach on this MI355X:
//  mode 0: bf16 MFMA 32x32x16 only (registers)          mode 1: + 16 ds_read_b128 per 24 MFMA (conv2 ratio)
//  mode 2: like 1 but fragments for step i+1 are read before the MFMAs of step i (software pipelined)
//  mode 3: 16x16x32 MFMA only
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void probe(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16384; i += 256) ((int*)smem)[i] = 0x3f803f80 + i;
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f32x16){0};
    const char* base = smem + wave * 8192 + (lane & 31) * 64 + (lane >> 5) * 16;
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) {
        a[i] = *(const bf16x8*)(base + i * 2048);
        b[i] = *(const bf16x8*)(base + 32768 + i * 2048);
    }
    if (MODE == 0) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[i], b[(i + r) & 3], acc[i], 0, 0, 0);
        }
    } else if (MODE == 1) {
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k16 = 0; k16 < 2; ++k16) {
                const char* p = base + ((it + k16) & 1) * 1024;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    a[i] = *(const bf16x8*)(p + i * 2048);
                    b[i] = *(const bf16x8*)(p + 32768 + i * 2048);
                }
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + r) & 3], b[i], acc[i], 0, 0, 0);
            }
        }
    } else if (MODE == 2) {
        bf16x8 a2[4], b2[4];
        for (int it = 0; it < iters; ++it) {
            const char* p = base + (it & 1) * 1024;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a2[i] = *(const bf16x8*)(p + i * 2048);
                b2[i] = *(const bf16x8*)(p + 32768 + i * 2048);
            }
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + r) & 3], b[i], acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                a[i] = *(const bf16x8*)(p + 512 + i * 2048);
                b[i] = *(const bf16x8*)(p + 512 + 32768 + i * 2048);
            }
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2[(i + r) & 3], b2[i], acc[i], 0, 0, 0);
        }
    } else if (MODE == 4 || MODE == 5) {
        // 64x128 wave tile: 8 A-side + 16 B-side reads (24 ds_read_b128) per 48 MFMA; MODE 5 adds a barrier per step
        f32x16 acc2[4];
        for (int i = 0; i < 4; ++i) acc2[i] = (f32x16){0};
        bf16x8 bb[8];
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k16 = 0; k16 < 2; ++k16) {
                const char* p = base + ((it + k16) & 1) * 1024;
#pragma unroll
                for (int i = 0; i < 4; ++i) a[i] = *(const bf16x8*)(p + i * 2048);
#pragma unroll
                for (int i = 0; i < 8; ++i) bb[i] = *(const bf16x8*)(p + 32768 + i * 2048 + (i >> 2) * 512);
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + r) & 3], bb[i], acc[i], 0, 0, 0);
                        acc2[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + r) & 3], bb[4 + i], acc2[i], 0, 0, 0);
                    }
            }
            if (MODE == 5) __syncthreads();
        }
        for (int i = 0; i < 4; ++i) acc[i][0] += acc2[i][3];
    } else if (MODE == 6) {
        // conv2 ratio (16 reads / 24 MFMA) with one workgroup barrier per step
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int k16 = 0; k16 < 2; ++k16) {
                const char* p = base + ((it + k16) & 1) * 1024;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    a[i] = *(const bf16x8*)(p + i * 2048);
                    b[i] = *(const bf16x8*)(p + 32768 + i * 2048);
                }
#pragma unroll
                for (int r = 0; r < 3; ++r)
#pragma unroll
                    for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + r) & 3], b[i], acc[i], 0, 0, 0);
            }
            __syncthreads();
        }
    } else {
        f32x4 c[8];
        for (int i = 0; i < 8; ++i) c[i] = (f32x4){0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int r = 0; r < 6; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[i & 3], b[(i + r) & 3], c[i], 0, 0, 0);
        }
        for (int i = 0; i < 8; ++i) acc[i & 3][0] += c[i][0];
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int blocks, int iters, double flop_per_iter_per_wave) {
    float* out;
    hipMalloc(&out, blocks * 256 * 4);
    hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    probe<MODE><<<blocks, 256, 65536>>>(out, 10);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    probe<MODE><<<blocks, 256, 65536>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double fl = (double)blocks * 4 * iters * flop_per_iter_per_wave;
    printf("%-44s blocks=%4d  %8.3f ms  %8.1f TFLOP/s\n", name, blocks, ms, fl / ms / 1e9);
    hipFree(out);
}

int main() {
    const double f32 = 2.0 * 32 * 32 * 16, f16 = 2.0 * 16 * 16 * 32;
    for (int blocks : {256, 512, 768}) {
        run<0>("mfma32x32x16 only (24/iter)", blocks, 4000, 24 * f32);
        run<3>("mfma16x16x32 only (48/iter)", blocks, 4000, 48 * f16);
        run<1>("16 ds_read_b128 + 24 mfma, read-then-compute", blocks, 4000, 24 * f32);
        run<2>("16 ds_read_b128 + 24 mfma, pipelined", blocks, 4000, 24 * f32);
        run<6>("16 ds_read_b128 + 24 mfma + barrier/step", blocks, 4000, 24 * f32);
        run<4>("24 ds_read_b128 + 48 mfma (64x128 wave tile)", blocks, 2000, 48 * f32);
        run<5>("24 ds_read_b128 + 48 mfma + barrier/step", blocks, 2000, 48 * f32);
    }
    return 0;
}
