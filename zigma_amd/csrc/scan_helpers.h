// DPP broadcast / buffer-addressing helpers shared by the forward (scan_tok.inc) and backward (scan_bwd.hip)
// token-major selective-scan kernels.  gfx950, wave64.
#pragma once
#include "zigma_common.h"

namespace zigma {

template <int M>
__device__ __forceinline__ float row_bcast(float v) {   // lane M of every 16-lane row -> whole row (DPP)
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x150 + M, 0xf, 0xf, true));
}
// y += row_bcast<M>(c) * h as ONE v_fmac_f32_dpp (hipcc folds DPP into v_mul but not into the tied-operand fmac).
// A VALU write of `c` needs 2 wait states before a DPP read of it and nothing inside an asm statement is padded
// by the compiler: pass every freshly produced `c` through dpp_settle() once.
template <int M>
__device__ __forceinline__ void fmac_bcast(float &y, float c, float h) {
    asm("v_fmac_f32_dpp %0, %1, %2 row_newbcast:%c3 row_mask:0xf bank_mask:0xf bound_ctrl:1"
        : "+v"(y) : "v"(c), "v"(h), "i"(M));
}
// One recurrence step of the 4 states of this wave, h_j = exp2(dv * a2_j) * h_j + B_j * du, as ONE scheduled block
// (the state-only pass has no y chain to hold the compiler's schedule together; left to itself it hoists the 64
// products of a tile, spills the prefetch registers and so serialises every tile behind its global loads).
// Bf holds B of 4 steps x 4 states in its 16-lane rows; it must be settled (dpp_settle) after its last VALU write.
template <int S>
__device__ __forceinline__ void step_h(float dv, float du, float Bf, const float (&a2)[4], float (&h)[4]) {
    float t0, t1, t2, t3, p0, p1, p2, p3;
    asm volatile(
        "v_mul_f32 %4, %12, %14\n\tv_mul_f32 %5, %12, %15\n\tv_mul_f32 %6, %12, %16\n\tv_mul_f32 %7, %12, %17\n\t"
        "v_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\t"
        "v_mul_f32_dpp %8, %18, %13 row_newbcast:%c19 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_mul_f32_dpp %9, %18, %13 row_newbcast:%c20 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_mul_f32_dpp %10, %18, %13 row_newbcast:%c21 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_mul_f32_dpp %11, %18, %13 row_newbcast:%c22 row_mask:0xf bank_mask:0xf bound_ctrl:1\n\t"
        "v_fma_f32 %0, %4, %0, %8\n\tv_fma_f32 %1, %5, %1, %9\n\tv_fma_f32 %2, %6, %2, %10\n\tv_fma_f32 %3, %7, %3, %11"
        : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3),
          "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3)
        : "v"(dv), "v"(du), "v"(a2[0]), "v"(a2[1]), "v"(a2[2]), "v"(a2[3]), "v"(Bf),
          "i"(S * 4 + 0), "i"(S * 4 + 1), "i"(S * 4 + 2), "i"(S * 4 + 3));
}
// The same step with B as plain (LDS-broadcast) operands — state-only pass of scan_tok2_kernel.
__device__ __forceinline__ void step_h_plain(float dv, float du, float b0, float b1, float b2, float b3, const float (&a2)[4],
                                             float (&h)[4]) {
    float t0, t1, t2, t3, p0, p1, p2, p3;
    asm volatile(
        "v_mul_f32 %4, %12, %14\n\tv_mul_f32 %5, %12, %15\n\tv_mul_f32 %6, %12, %16\n\tv_mul_f32 %7, %12, %17\n\t"
        "v_exp_f32 %4, %4\n\tv_exp_f32 %5, %5\n\tv_exp_f32 %6, %6\n\tv_exp_f32 %7, %7\n\t"
        "v_mul_f32 %8, %18, %13\n\tv_mul_f32 %9, %19, %13\n\tv_mul_f32 %10, %20, %13\n\tv_mul_f32 %11, %21, %13\n\t"
        "v_fma_f32 %0, %4, %0, %8\n\tv_fma_f32 %1, %5, %1, %9\n\tv_fma_f32 %2, %6, %2, %10\n\tv_fma_f32 %3, %7, %3, %11"
        : "+v"(h[0]), "+v"(h[1]), "+v"(h[2]), "+v"(h[3]), "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3),
          "=&v"(p0), "=&v"(p1), "=&v"(p2), "=&v"(p3)
        : "v"(dv), "v"(du), "v"(a2[0]), "v"(a2[1]), "v"(a2[2]), "v"(a2[3]), "v"(b0), "v"(b1), "v"(b2), "v"(b3));
}
__device__ __forceinline__ void dpp_settle(float &c) { asm volatile("s_nop 1" : "+v"(c)); }

using rsrc_t = __amdgpu_buffer_rsrc_t;
__device__ __forceinline__ rsrc_t make_rsrc(const void *base, int64_t bytes) {
    const unsigned n = bytes > 0x7fffffff ? 0x7fffffffu : static_cast<unsigned>(bytes < 0 ? 0 : bytes);
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void *>(base), 0, static_cast<int>(n), 0x00020000);
}
template <typename T> __device__ __forceinline__ typename T::raw buf_ld(rsrc_t r, unsigned voff, int soff) {
    if constexpr (sizeof(typename T::raw) == 2) return static_cast<uint16_t>(__builtin_amdgcn_raw_buffer_load_b16(r, voff, soff, 0));
    else return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0));
}
template <typename T> __device__ __forceinline__ void buf_st(typename T::raw v, rsrc_t r, unsigned voff, int soff) {
    if constexpr (sizeof(typename T::raw) == 2) __builtin_amdgcn_raw_buffer_store_b16(v, r, voff, soff, 0);
    else __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), r, voff, soff, 0);
}

}  // namespace zigma
