// Which bf16 MFMA shape is faster BY WALL CLOCK on random operands (the chip lowers its clock under MFMA load, and the clock it holds
// depends on the shape: MI355X_MICROARCH.md, DVFS give-back item 7)?  One wave per SIMD, every CU busy, the same 128 x 64 fp32 output
// tile per wave (the attention kernel's O^T tile; 128 accumulator registers) and the same operand fragments per 32-deep k-step
// (8 A fragments + 4 B fragments of 16 B per lane):
//    32x32x16:  4 x 2 blocks x 2 k-halves = 16 MFMAs of 32 cycles per k-step
//    16x16x32:  8 x 4 blocks              = 32 MFMAs of 16 cycles per k-step
// LDS = 1: the 12 fragments are re-read from LDS (ds_read_b128) every k-step; 0: they stay in registers.
// Prints wall time per k-step, the in-kernel clock (s_memtime / s_memrealtime) and TFLOP/s.  Each variant runs ~2 s back to back first.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned hash32(unsigned x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
__device__ __forceinline__ bf16x8 rnd_frag(unsigned seed, bool zero) {
    bf16x8 f;
    for (int j = 0; j < 8; ++j) {
        const unsigned u = hash32(seed * 8 + j);
        const float v = ((int)(u & 0xffff) - 32768) * (1.0f / 32768.0f);        // uniform in [-1, 1)
        f[j] = (__bf16)(zero ? 0.0f : v);
    }
    return f;
}

template <int SHAPE, int LDS>
__global__ __launch_bounds__(256, 1) void probe(float* out, long long* stamps, int iters, int zero) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    bf16x8* frag = (bf16x8*)smem + wave * 12 * 64;         // [12][64 lanes] per wave: lane-linear, conflict free
    bf16x8 a[8], b[4];
    for (int i = 0; i < 8; ++i) a[i] = rnd_frag((blockIdx.x * 256 + tid) * 12 + i, zero);
    for (int i = 0; i < 4; ++i) b[i] = rnd_frag((blockIdx.x * 256 + tid) * 12 + 8 + i, zero);
    for (int i = 0; i < 8; ++i) frag[i * 64 + lane] = a[i];
    for (int i = 0; i < 4; ++i) frag[(8 + i) * 64 + lane] = b[i];
    __syncthreads();
    f32x16 acc32[SHAPE == 32 ? 8 : 1];
    f32x4 acc16[SHAPE == 16 ? 32 : 1];
    for (auto& x : acc32) x = (f32x16){0};
    for (auto& x : acc16) x = (f32x4){0};
    const long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (LDS) {
#pragma unroll
            for (int i = 0; i < 8; ++i) a[i] = frag[i * 64 + lane];
#pragma unroll
            for (int i = 0; i < 4; ++i) b[i] = frag[(8 + i) * 64 + lane];
        }
        if (SHAPE == 32) {
#pragma unroll
            for (int kh = 0; kh < 2; ++kh)
#pragma unroll
                for (int d = 0; d < 4; ++d)
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc32[d * 2 + q]) : "v"(a[d * 2 + kh]), "v"(b[q * 2 + kh]));
        } else {
#pragma unroll
            for (int d = 0; d < 8; ++d)
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc16[d * 4 + q]) : "v"(a[d]), "v"(b[q]));
        }
        if (LDS) asm volatile("" ::: "memory");
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // (inline-asm MFMAs: the compiler pads nothing behind them)
    const long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (auto& x : acc32) s += x[0] + x[9];
    for (auto& x : acc16) s += x[0] + x[3];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0) {
        stamps[blockIdx.x * 2] = c1 - c0;
        stamps[blockIdx.x * 2 + 1] = r1 - r0;
    }
}

template <int SHAPE, int LDS>
void run(int zero) {
    const int blocks = 256;
    float* out;
    long long* st;
    hipMalloc(&out, blocks * 256 * 4);
    hipMalloc(&st, blocks * 16);
    const int lds = 100 * 1024;                 // one workgroup per CU
    hipFuncSetAttribute((const void*)probe<SHAPE, LDS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    const int iters = 40000;                    // ~10 ms per launch
    for (int i = 0; i < 200; ++i) probe<SHAPE, LDS><<<blocks, 256, lds>>>(out, st, iters, zero);      // ~2 s of load first
    hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) probe<SHAPE, LDS><<<blocks, 256, lds>>>(out, st, iters, zero);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    ms /= 20;
    std::vector<long long> h(blocks * 2);
    hipMemcpy(h.data(), st, blocks * 16, hipMemcpyDeviceToHost);
    std::vector<double> clk;
    for (int i = 0; i < blocks; ++i) clk.push_back((double)h[2 * i] / (double)h[2 * i + 1] * 0.1);      // GHz (s_memrealtime: 100 MHz)
    std::sort(clk.begin(), clk.end());
    const double flops = 2.0 * 128 * 64 * 32 * (double)iters * 4 * blocks;
    printf("shape %dx%d  operands %-9s data %-6s : %7.2f ns per k-step, clock %.2f GHz (median; s_memtime ticks per 10 ns: see note), %7.1f TFLOP/s\n", SHAPE, SHAPE,
           LDS ? "from LDS" : "registers", zero ? "zeros" : "random", ms * 1e6 / iters, clk[blocks / 2], flops / (ms * 1e-3) * 1e-12);
    hipFree(out);
    hipFree(st);
}

int main() {
    for (int zero = 0; zero < 2; ++zero) {
        run<32, 0>(zero);
        run<16, 0>(zero);
        run<32, 1>(zero);
        run<16, 1>(zero);
    }
    run<32, 1>(0);      // once more in the other order: drift check
    run<16, 1>(0);
    return 0;
}
