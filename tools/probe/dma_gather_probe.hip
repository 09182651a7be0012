// What does an LDS-DMA gather cost?  conv_gemm6 fills its activation windows with global_load_lds_dwordx4 whose 64 lanes read 16 B each from
// 8 pixels x 2 planes (16 segments of 64 B, pixels ~100 KB apart); the K loop showed 100 ... 360 cycles of wave time per such instruction.
// This probe issues N LDS-DMA instructions per wave (4 waves per workgroup, one workgroup per CU, every CU busy) with different lane -> address
// patterns and address modes and reports cycles per instruction: issue only (the wave's own clock until the last one is issued) and until
// everything has landed (vmcnt(0)).
//   pattern 0: 1 KiB contiguous per instruction            pattern 1: 16 segments of 64 B (conv_gemm6's windows)
//   pattern 2: 8 segments of 128 B                         pattern 3: 32 segments of 32 B (conv_gemm5's 16-channel windows)
//   mode 0: 64-bit address per lane (global_load_lds_dwordx4 v[a:a+1], off)     mode 1: scalar base + 32-bit lane offset
// build: hipcc --offload-arch=gfx950 -O3 -o tools/probe/dma_gather_probe tools/probe/dma_gather_probe.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <vector>
#include <algorithm>

constexpr int NI = 14;            // DMA instructions per wave and round
constexpr int ROUNDS = 64;

template <int MODE>
__global__ __launch_bounds__(256, 1) void probe(const char* buf, size_t bytes, int pattern, long long* out, int gap_nops) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned lds_wave = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) const char*)smem + wave * 1024);
    // lane -> byte offset inside one instruction's footprint
    const size_t pix_stride = 128 * 384 * 2;                  // one image row of 128 pixels x 384 channels x 2 B below (y-neighbours of a window column)
    unsigned off[NI];
    for (int i = 0; i < NI; ++i) {
        size_t o;
        const int q = lane;                                    // 16-B piece index inside the instruction
        if (pattern == 0) o = (size_t)q * 16;
        else if (pattern == 1) o = (size_t)(q >> 3) * pix_stride + (size_t)((q >> 2) & 1) * (bytes / 2) + (size_t)(q & 3) * 16;          // 8 pixels x 2 planes x 64 B
        else if (pattern == 2) o = (size_t)(q >> 3) * pix_stride + (size_t)(q & 7) * 16;                                                  // 8 pixels x 128 B
        else o = (size_t)(q >> 2) * pix_stride + (size_t)((q >> 1) & 1) * (bytes / 2) + (size_t)(q & 1) * 16;                             // 16 pixels x 2 planes x 32 B
        o += (size_t)(blockIdx.x * 4 + wave) * 4096 + (size_t)i * 8 * pix_stride + (size_t)i * 64;
        off[i] = (unsigned)(o % (bytes / 2 - (1 << 20)));       // stay inside the first half (plane 1 adds bytes / 2)
        off[i] &= ~15u;
    }
    long long t_issue = 0, t_land = 0;
    for (int r = 0; r < ROUNDS; ++r) {
        const char* base = buf + (size_t)(r & 7) * 65536;
        __builtin_amdgcn_s_barrier();
        const long long c0 = __builtin_amdgcn_s_memtime();
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const unsigned dst = lds_wave + (unsigned)(i * 4096);
            if (MODE == 0) {
                const char* p = base + off[i];
                asm volatile("s_mov_b32 m0, %1\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, off" ::"v"(p), "s"(dst) : "memory");
            } else {
                asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(off[i]), "s"(base), "s"(dst) : "memory");
            }
            for (int k = 0; k < gap_nops; ++k) asm volatile("s_nop 15" ::: "memory");     // (spacing between the instructions: 16 cycles each)
        }
        const long long c1 = __builtin_amdgcn_s_memtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const long long c2 = __builtin_amdgcn_s_memtime();
        if (r >= 8) t_issue += c1 - c0, t_land += c2 - c0;
    }
    if (lane == 0) {
        out[(blockIdx.x * 4 + wave) * 2] = t_issue / (ROUNDS - 8);
        out[(blockIdx.x * 4 + wave) * 2 + 1] = t_land / (ROUNDS - 8);
    }
}

int main() {
    const size_t bytes = (size_t)1 << 30;
    char* buf;
    long long* out;
    hipMalloc(&buf, bytes);
    hipMemset(buf, 1, bytes);
    const int ncu = 256;
    hipMalloc(&out, ncu * 4 * 2 * sizeof(long long));
    hipFuncSetAttribute((const void*)probe<0>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    hipFuncSetAttribute((const void*)probe<1>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
    std::vector<long long> h(ncu * 8);
    for (int gap : {0, 12}) {
        for (int mode = 0; mode < 2; ++mode) {
            for (int pat = 0; pat < 4; ++pat) {
                for (int rep = 0; rep < 2; ++rep) {
                    if (mode == 0) hipLaunchKernelGGL(probe<0>, dim3(ncu), dim3(256), 64 * 1024, 0, buf, bytes, pat, out, gap);
                    else hipLaunchKernelGGL(probe<1>, dim3(ncu), dim3(256), 64 * 1024, 0, buf, bytes, pat, out, gap);
                    hipDeviceSynchronize();
                }
                hipMemcpy(h.data(), out, h.size() * sizeof(long long), hipMemcpyDeviceToHost);
                std::vector<long long> a, b;
                for (int i = 0; i < ncu * 4; ++i) a.push_back(h[2 * i]), b.push_back(h[2 * i + 1]);
                std::sort(a.begin(), a.end());
                std::sort(b.begin(), b.end());
                printf("gap %3d cycles, mode %d (%s), pattern %d: issue %6.1f cycles / instruction (median wave; max %6.1f), landed %6.1f (max %6.1f); %d instructions per wave and round\n",
                       gap * 16, mode, mode ? "saddr + 32-bit lane offset" : "64-bit lane address", pat, (double)a[a.size() / 2] / NI, (double)a.back() / NI,
                       (double)b[b.size() / 2] / NI, (double)b.back() / NI, NI);
            }
        }
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) {
        printf("HIP error: %s\n", hipGetErrorString(e));
        return 1;
    }
    return 0;
}
