// Shared device/host helpers for libzigma_hip.so (gfx950 only; wave = 64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

#include "zigma_hip.h"

namespace zigma {

constexpr int kWave = 64;

typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

// ---- element types -----------------------------------------------------------------------------
// I/O element wrappers: 16-bit types are carried as raw uint16_t and widened by bit ops (bf16) or
// the hardware converter (f16); arithmetic is always float32.
struct F32 { using raw = float;    static constexpr int id = ZIGMA_F32;  };
struct F16 { using raw = uint16_t; static constexpr int id = ZIGMA_F16;  };
struct BF16 { using raw = uint16_t; static constexpr int id = ZIGMA_BF16; };

template <typename T> __device__ __forceinline__ float to_float(typename T::raw v);
template <> __device__ __forceinline__ float to_float<F32>(float v) { return v; }
template <> __device__ __forceinline__ float to_float<BF16>(uint16_t v) {
    return __uint_as_float(static_cast<uint32_t>(v) << 16);
}
template <> __device__ __forceinline__ float to_float<F16>(uint16_t v) {
    _Float16 h;
    __builtin_memcpy(&h, &v, 2);
    return static_cast<float>(h);
}

template <typename T> __device__ __forceinline__ typename T::raw from_float(float f);
template <> __device__ __forceinline__ float from_float<F32>(float f) { return f; }
template <> __device__ __forceinline__ uint16_t from_float<BF16>(float f) {
    // round-to-nearest-even (same as at::BFloat16); one v_cvt_pk_bf16_f32 on gfx950
    const __bf16 h = static_cast<__bf16>(f);
    uint16_t v;
    __builtin_memcpy(&v, &h, 2);
    return v;
}
template <> __device__ __forceinline__ uint16_t from_float<F16>(float f) {
    _Float16 h = static_cast<_Float16>(f);
    uint16_t v;
    __builtin_memcpy(&v, &h, 2);
    return v;
}

template <typename T> __device__ __forceinline__ float ld(const void *base, int64_t idx) {
    return to_float<T>(reinterpret_cast<const typename T::raw *>(base)[idx]);
}
template <typename T> __device__ __forceinline__ void st(void *base, int64_t idx, float v) {
    reinterpret_cast<typename T::raw *>(base)[idx] = from_float<T>(v);
}

// ---- math (reference numerics: selective_scan_fwd_kernel.cuh:153-156,216,293) ------------------
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }   // v_exp_f32
__device__ __forceinline__ float fast_log2(float x) { return __builtin_amdgcn_logf(x); }    // v_log_f32
__device__ __forceinline__ float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }     // v_rcp_f32

// softplus with pass-through above 20 (reference: x <= 20 ? log1pf(expf(x)) : x).
// log1p(t), t = e^x: short alternating series while t is small (keeps full relative accuracy for
// very negative x, where 1+t would round t away), hardware log2 otherwise.
__device__ __forceinline__ float softplus20(float x) {
    const float t = fast_exp2(fminf(x, 20.f) * kLog2e);
    const float series = t * (1.f + t * (-0.5f + t * 0.33333334f));   // |err| < t^4/4 <= 6e-8 t  for t < 2^-6
    const float big = fast_log2(1.f + t) * kLn2;                       // abs err ~1e-7 on a result >= 0.0155
    const float sp = t < 0.015625f ? series : big;
    return x > 20.f ? x : sp;
}

// The same function for results that are rounded to 16 bits right away (dt_proj's epilogue): softplus(x) = max(x, 0) +
// log1p(exp(-|x|)), 10 instructions instead of 14 (no clamp, no pass-through select: above 20 the log term is below 2e-9
// and vanishes in the rounding, like the reference's pass-through).  t = exp(-|x|) <= 1; below 2^-8 the log1p is the
// two-term series (relative error < 2^-17), above it log2(1 + t) (absolute error 2^-24 on a term >= 2^-8.5).
__device__ __forceinline__ float softplus20_r16(float x) {
    const float t = fast_exp2(-fabsf(x) * kLog2e);
    const float series = __builtin_fmaf(t * -0.5f, t, t);
    const float big = fast_log2(1.f + t) * kLn2;
    return fmaxf(x, 0.f) + (t < 0.00390625f ? series : big);
}

// x * sigmoid(x) written as z / (1 + exp(-z)) like the reference kernel.
__device__ __forceinline__ float silu(float z) { return z * fast_rcp(1.f + fast_exp2(-z * kLog2e)); }

// ---- host side -----------------------------------------------------------------------------------
inline int check_launch() {
    const hipError_t e = hipGetLastError();
    if (e == hipSuccess) return ZIGMA_OK;
    fprintf(stderr, "libzigma_hip: kernel launch failed: %s (%s)\n", hipGetErrorName(e), hipGetErrorString(e));
    return ZIGMA_ERR_LAUNCH;
}
void set_last_kernel(const char *name);

#define ZIGMA_DISPATCH_DTYPE(DT, T, ...)                                   \
    switch (DT) {                                                          \
        case ZIGMA_F32: { using T = ::zigma::F32; __VA_ARGS__; break; }    \
        case ZIGMA_F16: { using T = ::zigma::F16; __VA_ARGS__; break; }    \
        case ZIGMA_BF16: { using T = ::zigma::BF16; __VA_ARGS__; break; }  \
        default: return ZIGMA_ERR_DTYPE;                                   \
    }

}  // namespace zigma
