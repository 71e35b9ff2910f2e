// pk_common.h - shared host/device helpers for libpk_amd (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/pk_amd.h"

// ---- error plumbing -------------------------------------------------------
void pk_set_error(const char* fmt, ...);
// A/B levers of past experiments live behind ONE environment variable, PK_EXPERIMENT="key=value,key=value" (pk_lib.hip;
// the keys are listed in INTEGRATION.md): value of `key`, or nullptr.  Callers cache what they read.
const char* pk_experiment(const char* key);

#define PK_CHECK_HIP(expr)                                                                  \
    do {                                                                                    \
        hipError_t _e = (expr);                                                             \
        if (_e != hipSuccess) {                                                             \
            pk_set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return 1000 + (int)_e;                                                          \
        }                                                                                   \
    } while (0)

#define PK_LAUNCH_CHECK() PK_CHECK_HIP(hipGetLastError())

#define PK_REQUIRE(cond, ...)        \
    do {                             \
        if (!(cond)) {               \
            pk_set_error(__VA_ARGS__); \
            return 2;                \
        }                            \
    } while (0)

static inline hipStream_t pk_stream(void* s) { return (hipStream_t)s; }

// ---- device helpers -------------------------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef short bf16x8 __attribute__((ext_vector_type(8)));

#define PK_WAVE 64

// round-to-nearest-even fp32 -> bf16: the gfx950 conversion instruction (v_cvt_pk_bf16_f32; NaN stays a quiet NaN)
typedef __bf16 pk_bf16x2 __attribute__((ext_vector_type(2)));
typedef float pk_f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned short pk_f2bf(float f) { return __builtin_bit_cast(unsigned short, (__bf16)f); }
__device__ __forceinline__ float pk_bf2f(unsigned short h) { return __uint_as_float(((unsigned)h) << 16); }
__device__ __forceinline__ unsigned pk_pack_bf2(float lo, float hi) {
    const pk_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, pk_bf16x2));
}

__device__ __forceinline__ float pk_sigmoid(float x) { return 1.0f / (1.0f + __expf(-x)); }
__device__ __forceinline__ float pk_tanh(float x) {
    // tanh(x) = 1 - 2/(exp(2x)+1); exact limits for |x| large, ~1e-7 relative
    float e = __expf(2.0f * x);
    return 1.0f - 2.0f / (e + 1.0f);
}

// act_fun of the reference, neural_networks.py:36-57.  A translation unit that defines PK_CELL_FAST_MATH
// (the bf16 perf-mode kernels) gets hardware-rate exp / reciprocal (1-2 ulp) instead of the precise forms.
#ifdef PK_CELL_FAST_MATH
__device__ __forceinline__ float pk_fast_sigmoid(float x) { return __builtin_amdgcn_rcpf(1.0f + __expf(-x)); }
__device__ __forceinline__ float pk_fast_tanh(float x) {
    const float e = __expf(2.0f * fminf(x, 15.0f));  // tanh(15) == 1 in fp32; keeps exp finite
    return 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
}
#endif
__device__ __forceinline__ float pk_act(int act, float x) {
    switch (act) {
        case PK_ACT_RELU: return x > 0.f ? x : 0.f;
#ifdef PK_CELL_FAST_MATH
        case PK_ACT_TANH: return pk_fast_tanh(x);
        case PK_ACT_SIGMOID: return pk_fast_sigmoid(x);
#else
        case PK_ACT_TANH: return tanhf(x);
        case PK_ACT_SIGMOID: return 1.0f / (1.0f + expf(-x));
#endif
        case PK_ACT_LEAKY_RELU: return x > 0.f ? x : 0.2f * x;
        case PK_ACT_ELU: return x > 0.f ? x : (expf(x) - 1.0f);
        default: return x;
    }
}
// derivative expressed through the OUTPUT y = act(x)
__device__ __forceinline__ float pk_act_grad_from_out(int act, float y) {
    switch (act) {
        case PK_ACT_RELU: return y > 0.f ? 1.f : 0.f;
        case PK_ACT_TANH: return 1.f - y * y;
        case PK_ACT_SIGMOID: return y * (1.f - y);
        case PK_ACT_LEAKY_RELU: return y > 0.f ? 1.f : 0.2f;
        case PK_ACT_ELU: return y > 0.f ? 1.f : (y + 1.0f);
        default: return 1.f;
    }
}
// derivative expressed through the INPUT x (pre-activation)
__device__ __forceinline__ float pk_act_grad_from_in(int act, float x) {
    switch (act) {
        case PK_ACT_RELU: return x > 0.f ? 1.f : 0.f;
        case PK_ACT_TANH: { float t = tanhf(x); return 1.f - t * t; }
        case PK_ACT_SIGMOID: { float s = 1.0f / (1.0f + expf(-x)); return s * (1.f - s); }
        case PK_ACT_LEAKY_RELU: return x > 0.f ? 1.f : 0.2f;
        case PK_ACT_ELU: return x > 0.f ? 1.f : expf(x);
        default: return 1.f;
    }
}

__device__ __forceinline__ float pk_wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float pk_wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}
