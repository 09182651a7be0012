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

// ---- device helpers ----------------------------------------------------------------------------
// x ~= hi + lo with hi = bf16(x) (round to nearest even: identical to torch's .to(bfloat16)) and lo = bf16(x - hi)
__device__ __forceinline__ void split_bf16(float x, bf16_t& hi, bf16_t& lo) {
    hi = (bf16_t)x;
    lo = (bf16_t)(x - (float)hi);
}
__device__ __forceinline__ float join_bf16(bf16_t hi, bf16_t lo) { return (float)hi + (float)lo; }

__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }
__device__ __forceinline__ float sigmoid_f(float x) { return 1.0f / (1.0f + expf(-x)); }

__device__ __forceinline__ float apply_act(float x, int act) {
    switch (act) {
        case PPMS_ACT_RELU: return x < 0.0f ? 0.0f : x;   // not fmaxf: NaN must propagate like torch.relu (T == 1 case)
        case PPMS_ACT_GELU: return gelu_erf(x);
        case PPMS_ACT_SIGMOID: return sigmoid_f(x);
        case PPMS_ACT_TANH: return tanhf(x);
        case PPMS_ACT_ELU1: return x > 0.0f ? x + 1.0f : expf(x);      // elu(x) + 1 (attention.py:14-15)
        default: return x;
    }
}
