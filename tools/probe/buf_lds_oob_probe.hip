// Does an out-of-range lane of `buffer_load_dwordx4 ... offen lds` (LDS-DMA through a buffer resource) write ZEROS to its LDS slot?  conv_gemm6 would use that for
// the padding of its activation windows (no zero page, no per-lane select).  LDS is pre-filled with 0xdeadbeef; lanes with (lane & 3) == 1 get an offset beyond
// num_records.  Prints what those lanes' 16 bytes hold afterwards and what the in-range lanes hold.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
__global__ void k(const char* p, unsigned nrec, unsigned* out) {
    extern __shared__ unsigned smem[];
    for (int i = threadIdx.x; i < 64 * 4; i += 64) smem[i] = 0xdeadbeefu;
    __syncthreads();
    u32x4 srd;
    const unsigned long long a = (unsigned long long)p;
    srd[0] = __builtin_amdgcn_readfirstlane((unsigned)a);
    srd[1] = __builtin_amdgcn_readfirstlane((unsigned)(a >> 32) & 0xffffu);
    srd[2] = nrec;
    srd[3] = 0x00020000u;
    const unsigned off = (threadIdx.x & 3) == 1 ? 0xFFFFFFF0u : threadIdx.x * 16;
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned*)smem);
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, 0 offen lds" ::"v"(off), "s"(srd), "s"(dst) : "memory");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 64 * 4; i += 64) out[i] = smem[i];
}
int main() {
    char* buf; unsigned* out;
    hipMalloc(&buf, 1 << 20); hipMalloc(&out, 1024 * 4);
    unsigned h[1 << 18];
    for (int i = 0; i < (1 << 18); ++i) h[i] = 0x1000000u + i;
    hipMemcpy(buf, h, 1 << 20, hipMemcpyHostToDevice);
    for (unsigned nrec : {0xFFFFFF00u, 4096u}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, buf, nrec, out);
        unsigned r[256];
        hipMemcpy(r, out, sizeof(r), hipMemcpyDeviceToHost);
        int oob_zero = 0, oob_other = 0, in_ok = 0, in_bad = 0;
        for (int lane = 0; lane < 64; ++lane)
            for (int j = 0; j < 4; ++j) {
                const unsigned v = r[lane * 4 + j];
                if ((lane & 3) == 1) { if (v == 0) ++oob_zero; else ++oob_other; }
                else { if (v == 0x1000000u + lane * 4 + j) ++in_ok; else ++in_bad; }
            }
        printf("num_records 0x%x: out-of-range lanes: %d dwords zero, %d other (first: 0x%x); in-range lanes: %d right, %d wrong\n", nrec, oob_zero, oob_other, r[4], in_ok, in_bad);
    }
    return hipGetLastError() != hipSuccess;
}
