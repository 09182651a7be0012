// Reproducer: a small gather kernel (4 dependent-free global loads per thread, bilinear resize) running on one stream while an
// MFMA + LDS heavy kernel runs on another.  Counts output elements that differ from the single-stream result.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int VAR>
__global__ __launch_bounds__(256) void bilinear_k(const float* __restrict__ src, float* __restrict__ dst, int H, int W, int OH, int OW, float sh, float sw,
                                                  long n) {
    const long idx = (long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const int ox = (int)(idx % OW), oy = (int)((idx / OW) % OH);
    const long plane = idx / ((long)OW * OH);
    float fy = fmaxf(sh * (oy + 0.5f) - 0.5f, 0.0f), fx = fmaxf(sw * (ox + 0.5f) - 0.5f, 0.0f);
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = y0 + ((y0 < H - 1) ? 1 : 0), x1 = x0 + ((x0 < W - 1) ? 1 : 0);
    const float ly = fy - y0, lx = fx - x0, hy = 1.0f - ly, hx = 1.0f - lx;
    const float* s = src + plane * H * W;
    float s00, s01, s10, s11;
    if (VAR == 0) {
        s00 = s[y0 * W + x0]; s01 = s[y0 * W + x1]; s10 = s[y1 * W + x0]; s11 = s[y1 * W + x1];
    } else if (VAR == 1) {
        s00 = __builtin_nontemporal_load(s + y0 * W + x0); s01 = __builtin_nontemporal_load(s + y0 * W + x1);
        s10 = __builtin_nontemporal_load(s + y1 * W + x0); s11 = __builtin_nontemporal_load(s + y1 * W + x1);
    } else {
        const volatile float* vs = s;
        s00 = vs[y0 * W + x0]; s01 = vs[y0 * W + x1]; s10 = vs[y1 * W + x0]; s11 = vs[y1 * W + x1];
    }
    dst[idx] = hy * (hx * s00 + lx * s01) + ly * (hx * s10 + lx * s11);
}

__global__ __launch_bounds__(256, 2) void heavy_k(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16384; i += 256) ((int*)smem)[i] = 0x3f803f80 + i;
    __syncthreads();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = (f32x16){0};
    const char* base = smem + wave * 8192 + (lane & 31) * 64 + (lane >> 5) * 16;
    bf16x8 a[4], b[4];
    for (int it = 0; it < iters; ++it) {
        const char* p = base + (it & 1) * 1024;
        for (int i = 0; i < 4; ++i) { a[i] = *(const bf16x8*)(p + i * 2048); b[i] = *(const bf16x8*)(p + 32768 + i * 2048); }
        for (int r = 0; r < 3; ++r)
            for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + r) & 3], b[i], acc[i], 0, 0, 0);
        __syncthreads();
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) s += acc[i][0] + acc[i][7];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int VAR>
int trial(bool concurrent, const float* src, float* dst, float* ref, float* hout, int T, int H, int W, hipStream_t s1, hipStream_t s2, std::vector<float>& a,
          std::vector<float>& b) {
    const int OH = 4 * H, OW = 4 * W;
    const long n = (long)T * OH * OW;
    if (concurrent) heavy_k<<<512, 256, 65536, s2>>>(hout, 3000);
    bilinear_k<VAR><<<(n + 255) / 256, 256, 0, s1>>>(src, dst, H, W, OH, OW, 0.25f, 0.25f, n);
    hipDeviceSynchronize();
    hipMemcpy(a.data(), dst, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), ref, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (long i = 0; i < n; ++i) bad += a[i] != b[i];
    return bad;
}

int main() {
    const int T = 5, H = 80, W = 128;
    const long nin = (long)T * H * W, n = nin * 16;
    float *src, *dst, *ref, *hout;
    hipMalloc(&src, nin * 4); hipMalloc(&dst, n * 4); hipMalloc(&ref, n * 4); hipMalloc(&hout, 512 * 256 * 4);
    std::vector<float> h(nin, 1.0f), a(n), b(n);
    hipMemcpy(src, h.data(), nin * 4, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void*)heavy_k, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipStream_t s1, s2;
    hipStreamCreate(&s1); hipStreamCreate(&s2);
    bilinear_k<0><<<(n + 255) / 256, 256, 0, s1>>>(src, ref, H, W, 4 * H, 4 * W, 0.25f, 0.25f, n);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 6; ++rep) {
        printf("rep %d  alone: %d   concurrent: plain %d  nontemporal %d  volatile %d\n", rep, trial<0>(false, src, dst, ref, hout, T, H, W, s1, s2, a, b),
               trial<0>(true, src, dst, ref, hout, T, H, W, s1, s2, a, b), trial<1>(true, src, dst, ref, hout, T, H, W, s1, s2, a, b),
               trial<2>(true, src, dst, ref, hout, T, H, W, s1, s2, a, b));
    }
    return 0;
}
