// 4-channel packing and buffer-addressed row access shared by the causal conv1d forward and backward kernels.
#pragma once
#include "zigma_common.h"

namespace zigma {

template <typename T, int VEC> struct Pack;
template <> struct Pack<BF16, 4> { using type = uint2; };
template <> struct Pack<F16, 4> { using type = uint2; };
template <> struct Pack<F32, 4> { using type = uint4; };

template <typename T> __device__ __forceinline__ void unpack4(const typename Pack<T, 4>::type &r, float (&f)[4]);
template <> __device__ __forceinline__ void unpack4<BF16>(const uint2 &r, float (&f)[4]) {
    f[0] = __uint_as_float(r.x << 16); f[1] = __uint_as_float(r.x & 0xffff0000u);
    f[2] = __uint_as_float(r.y << 16); f[3] = __uint_as_float(r.y & 0xffff0000u);
}
template <> __device__ __forceinline__ void unpack4<F16>(const uint2 &r, float (&f)[4]) {
    f[0] = to_float<F16>(r.x & 0xffffu); f[1] = to_float<F16>(r.x >> 16);
    f[2] = to_float<F16>(r.y & 0xffffu); f[3] = to_float<F16>(r.y >> 16);
}
template <> __device__ __forceinline__ void unpack4<F32>(const uint4 &r, float (&f)[4]) {
    f[0] = __uint_as_float(r.x); f[1] = __uint_as_float(r.y); f[2] = __uint_as_float(r.z); f[3] = __uint_as_float(r.w);
}
template <typename T> __device__ __forceinline__ typename Pack<T, 4>::type pack4(const float (&f)[4]);
template <> __device__ __forceinline__ uint2 pack4<BF16>(const float (&f)[4]) {
    return make_uint2(from_float<BF16>(f[0]) | (uint32_t(from_float<BF16>(f[1])) << 16),
                      from_float<BF16>(f[2]) | (uint32_t(from_float<BF16>(f[3])) << 16));
}
template <> __device__ __forceinline__ uint2 pack4<F16>(const float (&f)[4]) {
    return make_uint2(from_float<F16>(f[0]) | (uint32_t(from_float<F16>(f[1])) << 16),
                      from_float<F16>(f[2]) | (uint32_t(from_float<F16>(f[3])) << 16));
}
template <> __device__ __forceinline__ uint4 pack4<F32>(const float (&f)[4]) {
    return make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
}

using rsrc_t = __amdgpu_buffer_rsrc_t;
template <typename IO> __device__ __forceinline__ typename Pack<IO, 4>::type buf_ld4(rsrc_t r, unsigned voff, int soff) {
    if constexpr (sizeof(typename IO::raw) == 2) {
        const auto v = __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0);
        return make_uint2(v[0], v[1]);
    } else {
        const auto v = __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0);
        return make_uint4(v[0], v[1], v[2], v[3]);
    }
}
template <typename IO> __device__ __forceinline__ void buf_st4(const typename Pack<IO, 4>::type &v, rsrc_t r, unsigned voff, int soff) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    typedef unsigned u4 __attribute__((ext_vector_type(4)));
    if constexpr (sizeof(typename IO::raw) == 2) __builtin_amdgcn_raw_buffer_store_b64(u2{v.x, v.y}, r, voff, soff, 0);
    else __builtin_amdgcn_raw_buffer_store_b128(u4{v.x, v.y, v.z, v.w}, r, voff, soff, 0);
}

}  // namespace zigma
