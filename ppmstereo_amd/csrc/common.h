// Shared device/host helpers for libppms (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>
#include "ppms.h"

typedef __bf16 bf16_t;
typedef bf16_t bf16x8 __attribute__((ext_vector_type(8)));
typedef bf16_t bf16x4 __attribute__((ext_vector_type(4)));
typedef bf16_t bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// pointers fetched from a descriptor in memory are generic to the compiler; these casts make the access a
// global_load / global_store (address space 1) instead of a flat one
#define PPMS_GLOBAL __attribute__((address_space(1)))
__device__ __forceinline__ u32x4 gload16(const void* p) { return *(const PPMS_GLOBAL u32x4*)(uintptr_t)p; }
__device__ __forceinline__ void gstore16(void* p, u32x4 v) { *(PPMS_GLOBAL u32x4*)(uintptr_t)p = v; }
// typed forms.  A FLAT access also counts in lgkmcnt, so the next LDS wait (the epilogues' staging reads) would have to sit out the
// whole memory round trip of every earlier store: the conv epilogues ran at one HBM latency per 8-row step until these were used.
template <class T> __device__ __forceinline__ T gld(const void* p) { return *(const PPMS_GLOBAL T*)(uintptr_t)p; }
template <class T> __device__ __forceinline__ void gst(void* p, const T& v) { *(PPMS_GLOBAL T*)(uintptr_t)p = v; }

// ---- error plumbing (no C++ exceptions cross the ABI) ----------------------------------------
void ppms_set_error(const char* fmt, ...);
int ppms_check_launch(const char* what);

#define PPMS_REQUIRE(cond, ...)                  \
    do {                                         \
        if (!(cond)) {                           \
            ppms_set_error(__VA_ARGS__);         \
            return PPMS_EINVAL;                  \
        }                                        \
    } while (0)

static inline int ceil_div(int64_t a, int64_t b) { return (int)((a + b - 1) / b); }

// Per-device one-time initialisation (dynamic-LDS limits of the kernels): the only mutable state of the library is this
// lazily-initialised, immutable-afterwards table, guarded by std::call_once -- one flag per device, so a second GPU used
// from the same process gets its own hipFuncSetAttribute calls.  Usage:
//     static ppms_device_once once;  once.run([] { hipFuncSetAttribute(...); });
#ifdef __cplusplus
#include <mutex>
struct ppms_device_once {
    static constexpr int MAX_DEV = 64;
    std::once_flag flag[MAX_DEV];
    template <typename F>
    void run(F&& f) {
        int dev = 0;
        (void)hipGetDevice(&dev);
        std::call_once(flag[(unsigned)dev % MAX_DEV], f);
    }
};

// compute units of the current device (cached per device; 256 if the query fails)
inline int ppms_num_cus() {
    static int cached[64] = {0};
    int dev = 0;
    (void)hipGetDevice(&dev);
    int& c = cached[(unsigned)dev % 64];
    if (c == 0) {
        int n = 0;
        c = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
    }
    return c;
}

#endif

// ---- device helpers ----------------------------------------------------------------------------
// x ~= hi + lo with hi = bf16(x) (round to nearest even: identical to torch's .to(bfloat16)) and lo = bf16(x - hi)
__device__ __forceinline__ void split_bf16(float x, bf16_t& hi, bf16_t& lo) {
    hi = (bf16_t)x;
    lo = (bf16_t)(x - (float)hi);
}
__device__ __forceinline__ float join_bf16(bf16_t hi, bf16_t lo) { return (float)hi + (float)lo; }
// One element of the attention's transposed value operand V^T (ppms_epilogue.out_vt / vt_f16): bf16(y) as the reference casts V
// (ppmstereo.py:550) -- or, f16 != 0, the fp16 number equal to that bf16 value (saturated at fp16's largest finite value), returned as
// the 16 bits a bf16_t store writes.
__device__ __forceinline__ bf16_t vt_enc(float y, int f16) {
    const bf16_t b = (bf16_t)y;
    if (!f16) return b;
    float v = (float)b;
    v = v > 65504.0f ? 65504.0f : (v < -65504.0f ? -65504.0f : v);      // (NaN passes)
    return __builtin_bit_cast(bf16_t, (_Float16)v);
}

__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }      // library-accurate (QAM confidence path)

// Branch-free transcendentals of the fused epilogues (GRU gates, GELU): the ocml erff / tanhf / expf + IEEE division cost
// 25-60 instructions per element with divergent range branches, which made the activation the longest part of the
// per-pixel layer chains and ~3 % of an iteration.  These use the raw v_exp_f32 / v_rcp_f32 (1 ulp each); absolute
// errors: sigmoid, tanh <= ~1e-7, erf <= 1.5e-7 (Abramowitz & Stegun 7.1.26), i.e. the rounding level of fp32 values
// of magnitude 1, far inside the 3e-5 tolerance the bf16x3 convolutions are tested to.  NaN propagates.
__device__ __forceinline__ float exp_fast(float x) { return __builtin_amdgcn_exp2f(x * 1.4426950408889634f); }
__device__ __forceinline__ float sigmoid_fast(float x) { return __builtin_amdgcn_rcpf(1.0f + exp_fast(-x)); }
__device__ __forceinline__ float tanh_fast(float x) {
    const float big = 1.0f - 2.0f * __builtin_amdgcn_rcpf(1.0f + exp_fast(2.0f * x));       // +-1 at the infinities
    const float x2 = x * x;
    const float small = x * (1.0f + x2 * (-0.33333333333f + x2 * (0.13333333333f - 0.05396825397f * x2)));   // |x| < 1/8: rel. 2e-10
    return __builtin_fabsf(x) < 0.125f ? small : big;
}
__device__ __forceinline__ float erf_fast(float x) {
    const float ax = __builtin_fabsf(x);
    const float t = __builtin_amdgcn_rcpf(1.0f + 0.3275911f * ax);
    const float poly = t * (0.254829592f + t * (-0.284496736f + t * (1.421413741f + t * (-1.453152027f + t * 1.061405429f))));
    const float y = 1.0f - poly * exp_fast(-ax * ax);
    return __builtin_copysignf(y, x);
}
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_fast(x * 0.70710678118654752440f)); }

__device__ __forceinline__ float apply_act(float x, int act) {
    switch (act) {
        case PPMS_ACT_RELU: return x < 0.0f ? 0.0f : x;   // not fmaxf: NaN must propagate like torch.relu (T == 1 case)
        case PPMS_ACT_GELU: return gelu_erf(x);
        case PPMS_ACT_SIGMOID: return sigmoid_fast(x);
        case PPMS_ACT_TANH: return tanh_fast(x);
        case PPMS_ACT_ELU1: return x > 0.0f ? x + 1.0f : expf(x);      // elu(x) + 1 (attention.py:14-15)
        default: return x;
    }
}

// The same on N values with ONE dispatch: called per element, the switch above becomes a chain of scalar branches around every value
// (8 chains per 8-cout row: most of the 0.8 us an epilogue row step took before this form existed).
template <int N>
__device__ __forceinline__ void apply_act_n(float (&y)[N], int act, float scale) {
    switch (act) {
        case PPMS_ACT_RELU:
#pragma unroll
            for (int j = 0; j < N; ++j) y[j] = (y[j] < 0.0f ? 0.0f : y[j]) * scale;
            break;
        case PPMS_ACT_GELU:
#pragma unroll
            for (int j = 0; j < N; ++j) y[j] = gelu_erf(y[j]) * scale;
            break;
        case PPMS_ACT_SIGMOID:
#pragma unroll
            for (int j = 0; j < N; ++j) y[j] = sigmoid_fast(y[j]) * scale;
            break;
        case PPMS_ACT_TANH:
#pragma unroll
            for (int j = 0; j < N; ++j) y[j] = tanh_fast(y[j]) * scale;
            break;
        case PPMS_ACT_ELU1:
#pragma unroll
            for (int j = 0; j < N; ++j) y[j] = (y[j] > 0.0f ? y[j] + 1.0f : expf(y[j])) * scale;
            break;
        default:
#pragma unroll
            for (int j = 0; j < N; ++j) y[j] = y[j] * scale;
            break;
    }
}

// conv_gemm2.hip: the reduce half of a K-sliced convolution launch (shared with conv_gemm5.hip)
struct ppms_conv;
int ppms_launch_slice_reduce(const ppms_conv* d, const ppms_conv* dev_desc, const float* workspace, int nslice, void* stream);
