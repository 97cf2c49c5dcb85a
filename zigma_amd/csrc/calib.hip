// Box calibration for bench.py: two FIXED kernels timed in the same process as the headline run, so that a change of the
// headline between rounds can be told from a change of silicon (boxes of the pool differ by 3-4 %: DESIGN.md §0).  Not part of the drop-in
// surface — nothing of the reference maps to it.
//   mode 0: copy `bytes` (multiple of 16 x 256) from src to dst with 16-byte accesses, 2048 workgroups grid-stride  -> HBM GB/s (read + write)
//   mode 1: 1280 workgroups x 4 waves (= 5 waves per SIMD, the scan kernel's occupancy), `iters` x 16 independent v_fma_f32 per wave
//   mode 2: the same with 8 independent v_exp_f32 per iteration
// dst receives one float per thread in modes 1 / 2 (so that the loops are not dead code).
#include "zigma_common.h"

namespace zigma {

__global__ __launch_bounds__(256) void calib_copy_kernel(const uint4 *__restrict__ src, uint4 *__restrict__ dst, int64_t n16) {
    const int64_t stride = static_cast<int64_t>(gridDim.x) * 256;
    for (int64_t i = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x; i < n16; i += stride) dst[i] = src[i];
}

template <int MODE>
__global__ __launch_bounds__(256) void calib_valu_kernel(float *__restrict__ dst, int iters) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float m = 0.999f, c = 1e-3f;
    for (int i = 0; i < iters; ++i) {
        if constexpr (MODE == 1) {
            asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         "v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                         "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));
        } else {
            asm volatile("v_exp_f32 %0, %8\n v_exp_f32 %1, %8\n v_exp_f32 %2, %8\n v_exp_f32 %3, %8\n"
                         "v_exp_f32 %4, %8\n v_exp_f32 %5, %8\n v_exp_f32 %6, %8\n v_exp_f32 %7, %8\n"
                         : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m));
        }
    }
    dst[static_cast<size_t>(blockIdx.x) * 256 + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
}

}  // namespace zigma

using namespace zigma;

extern "C" int zigma_calib_launch(const zigma_calib_params_t *pp, void *stream_) {
    if (!pp) return ZIGMA_ERR_NULL;
    (void)hipGetLastError();
    const zigma_calib_params_t &p = *pp;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!p.dst) return ZIGMA_ERR_NULL;
    if (p.mode == 0) {
        if (!p.src) return ZIGMA_ERR_NULL;
        if (p.bytes <= 0 || p.bytes % 4096 != 0) return ZIGMA_ERR_SHAPE;
        if ((reinterpret_cast<uintptr_t>(p.src) | reinterpret_cast<uintptr_t>(p.dst)) % 16 != 0) return ZIGMA_ERR_STRIDE;
        hipLaunchKernelGGL(calib_copy_kernel, dim3(2048), dim3(256), 0, stream, static_cast<const uint4 *>(p.src), static_cast<uint4 *>(p.dst), p.bytes / 16);
        set_last_kernel("calib_copy");
    } else if (p.mode == 1 || p.mode == 2) {
        if (p.iters < 1 || p.bytes < static_cast<int64_t>(1280) * 256 * 4) return ZIGMA_ERR_SHAPE;       // dst: one float per thread
        if (p.mode == 1) hipLaunchKernelGGL(calib_valu_kernel<1>, dim3(1280), dim3(256), 0, stream, static_cast<float *>(p.dst), p.iters);
        else hipLaunchKernelGGL(calib_valu_kernel<2>, dim3(1280), dim3(256), 0, stream, static_cast<float *>(p.dst), p.iters);
        set_last_kernel(p.mode == 1 ? "calib_v_fma" : "calib_v_exp");
    } else {
        return ZIGMA_ERR_UNSUPPORTED;
    }
    return check_launch();
}
