// Do four global loads of a wave land IN ORDER with respect to s_waitcnt vmcnt (what the compiler's counted waits assume on gfx9)?
// Each thread issues loads 0..3 (inline asm, so that the order and the waits are exactly these), pre-fills the destination of load 2 with
// a sentinel, waits vmcnt(1) -- "loads 0, 1, 2 have landed" -- and immediately copies register 2 with a plain v_mov.  A sentinel in
// the output means load 2's data was not there although the counter said so.  Built as a shared library so that a Python harness can
// run it beside the library's convolution kernels (tools/vmcnt_order_probe.py).
#include <hip/hip_runtime.h>
#include <stdint.h>

__global__ __launch_bounds__(256) void order_kernel(const float* __restrict__ src, float* __restrict__ out, long n, int W, int gap) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const long a0 = idx, a1 = (idx + 1) % n, a2 = (idx + W) % n, a3 = (idx + W + 1) % n;       // the four taps of a bilinear stencil
    const float* p0 = src + a0;
    const float* p1 = src + a1;
    const float* p2 = src + a2;
    const float* p3 = src + a3;
    float v0, v1, v2 = -12345.0f, v3, c2;
    asm volatile(
        "global_load_dword %0, %5, off\n\t"
        "global_load_dword %1, %6, off\n\t"
        "global_load_dword %2, %7, off\n\t"
        "global_load_dword %3, %8, off\n\t"
        "s_waitcnt vmcnt(1)\n\t"
        "v_mov_b32 %4, %2\n\t"
        "s_waitcnt vmcnt(0)"
        : "=&v"(v0), "=&v"(v1), "+&v"(v2), "=&v"(v3), "=&v"(c2)
        : "v"(p0), "v"(p1), "v"(p2), "v"(p3)
        : "memory");
    // out: 1 if the early copy of load 2 disagrees with its final value
    out[idx] = (c2 != v2) ? 1.0f : 0.0f;
    if (gap == 12345) out[idx] += v0 + v1 + v3;        // keep the other loads alive
}

// The failing pattern of the packed-fp32 finding (docs/LOG_r01_r05.md section 5), instruction for instruction: loads 0..3 land in the register
// pairs (l0, l3) and (l2, l1); behind vmcnt(1) a v_pk_mul_f32 reads the pair (l2, l1), behind vmcnt(0) another reads (l0, l3).
typedef float f32x2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void pk_after_wait_kernel(const float* __restrict__ src, float* __restrict__ out, long n, int W, int scalar_ops) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const float* p0 = src + idx;
    const float* p1 = src + (idx + 1) % n;
    const float* p2 = src + (idx + W) % n;
    const float* p3 = src + (idx + W + 1) % n;
    f32x2 ra, rb;
    float l0, l1, l2, l3;
    const f32x2 w = {0.75f, 0.25f};
    if (!scalar_ops) {
        // fixed registers so that the packed ops read the load destinations themselves: (v40, v41) = loads (0, 3), (v42, v43) = loads (2, 1)
        asm volatile(
            "global_load_dword v40, %6, off\n\t"
            "global_load_dword v43, %7, off\n\t"
            "global_load_dword v42, %8, off\n\t"
            "global_load_dword v41, %9, off\n\t"
            "s_waitcnt vmcnt(1)\n\t"
            "v_pk_mul_f32 %1, v[42:43], %10\n\t"
            "s_waitcnt vmcnt(0)\n\t"
            "v_pk_mul_f32 %0, v[40:41], %10\n\t"
            "v_mov_b32 %2, v40\n\t"
            "v_mov_b32 %3, v43\n\t"
            "v_mov_b32 %4, v42\n\t"
            "v_mov_b32 %5, v41"
            : "=&v"(ra), "=&v"(rb), "=&v"(l0), "=&v"(l1), "=&v"(l2), "=&v"(l3)
            : "v"(p0), "v"(p1), "v"(p2), "v"(p3), "v"(w)
            : "memory", "v40", "v41", "v42", "v43");
    } else {
        l0 = *p0, l1 = *p1, l2 = *p2, l3 = *p3;
        ra = (f32x2){l0, l3} * w, rb = (f32x2){l2, l1} * w;
    }
    const bool bad = ra[0] != l0 * w[0] || ra[1] != l3 * w[1] || rb[0] != l2 * w[0] || rb[1] != l1 * w[1];
    out[idx] = bad ? 1.0f : 0.0f;
}

extern "C" int pk_after_wait_launch(const float* src, float* out, long n, int W, int scalar_ops, void* stream) {
    hipLaunchKernelGGL(pk_after_wait_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, out, n, W, scalar_ops);
    return (int)hipGetLastError();
}

extern "C" int vmcnt_order_launch(const float* src, float* out, long n, int W, void* stream) {
    hipLaunchKernelGGL(order_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, (hipStream_t)stream, src, out, n, W, 0);
    return (int)hipGetLastError();
}
