// Fused (gated-branch add) + residual add + RMSNorm/LayerNorm (+ adaLN modulate) forward, gfx950.
// C ABI: zigma_add_norm_fwd (include/zigma_hip.h).
//
// Replaces the Triton kernel _layer_norm_fwd_1pass_kernel (reference dis_mamba/mamba_ssm/ops/triton/
// layernorm.py:65-120) and folds in the elementwise glue Block.forward wraps around it
// (model_zigma.py:53-54,441-458): the gate*branch residual of the previous sub-layer on the way in,
// the adaLN modulate on the way out.  Pure HBM streaming: one wave owns one row, the row lives in
// registers between the statistics and the normalisation, so every operand is read or written once.
#include "zigma_common.h"

namespace zigma {

// sum over the LPR lanes that share a row (LPR = 64: the whole wave)
template <int LPR> __device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int s = LPR / 2; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
    return v;
}

template <typename T> struct V4;  // 4 consecutive elements
template <> struct V4<F32> { using type = uint4; };
template <> struct V4<F16> { using type = uint2; };
template <> struct V4<BF16> { using type = uint2; };

template <typename T> __device__ __forceinline__ void load4(const void *base, int64_t idx, float (&f)[4]) {
    if constexpr (T::id == ZIGMA_F32) {
        const uint4 r = *reinterpret_cast<const uint4 *>(reinterpret_cast<const float *>(base) + idx);
        f[0] = __uint_as_float(r.x); f[1] = __uint_as_float(r.y); f[2] = __uint_as_float(r.z); f[3] = __uint_as_float(r.w);
    } else {
        const uint2 r = *reinterpret_cast<const uint2 *>(reinterpret_cast<const uint16_t *>(base) + idx);
        f[0] = to_float<T>(r.x & 0xffffu); f[1] = to_float<T>(r.x >> 16);
        f[2] = to_float<T>(r.y & 0xffffu); f[3] = to_float<T>(r.y >> 16);
    }
}
template <typename T> __device__ __forceinline__ void store4(void *base, int64_t idx, const float (&f)[4]) {
    if constexpr (T::id == ZIGMA_F32) {
        *reinterpret_cast<uint4 *>(reinterpret_cast<float *>(base) + idx) =
            make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
    } else {
        *reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(base) + idx) =
            make_uint2(from_float<T>(f[0]) | (uint32_t(from_float<T>(f[1])) << 16),
                       from_float<T>(f[2]) | (uint32_t(from_float<T>(f[3])) << 16));
    }
}
// VEC consecutive elements: 1 (scalar), 4, or 8 (two float4 / one 16-byte piece of 16-bit elements)
template <typename T, int VEC> __device__ __forceinline__ void loadv(const void *base, int64_t idx, float (&f)[VEC]) {
    if constexpr (VEC == 1) {
        f[0] = ld<T>(base, idx);
    } else if constexpr (VEC == 4) {
        load4<T>(base, idx, f);
    } else if constexpr (T::id == ZIGMA_F32) {
        float a[4], b[4];
        load4<T>(base, idx, a);
        load4<T>(base, idx + 4, b);
#pragma unroll
        for (int i = 0; i < 4; ++i) { f[i] = a[i]; f[4 + i] = b[i]; }
    } else {
        const uint4 r = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint16_t *>(base) + idx);
        const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) { f[2 * i] = to_float<T>(w[i] & 0xffffu); f[2 * i + 1] = to_float<T>(w[i] >> 16); }
    }
}
template <typename T, int VEC> __device__ __forceinline__ void storev(void *base, int64_t idx, const float (&f)[VEC]) {
    if constexpr (VEC == 1) {
        st<T>(base, idx, f[0]);
    } else if constexpr (VEC == 4) {
        store4<T>(base, idx, f);
    } else if constexpr (T::id == ZIGMA_F32) {
        const float a[4] = {f[0], f[1], f[2], f[3]}, b[4] = {f[4], f[5], f[6], f[7]};
        store4<T>(base, idx, a);
        store4<T>(base, idx + 4, b);
    } else {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) w[i] = from_float<T>(f[2 * i]) | (uint32_t(from_float<T>(f[2 * i + 1])) << 16);
        *reinterpret_cast<uint4 *>(reinterpret_cast<uint16_t *>(base) + idx) = make_uint4(w[0], w[1], w[2], w[3]);
    }
}
// value the tensor would hold after being stored in dtype T (the reference materialises these tensors)
template <typename T> __device__ __forceinline__ float rnd(float v) { return to_float<T>(from_float<T>(v)); }

// VEC = 8 / 4 (vector paths) or 1 (scalar path, any alignment); a row is shared by LPR lanes (64 / LPR rows per wave) that walk it in
// ITERS chunks of LPR * VEC columns.  LPR = 16 is for rows of a few hundred 16-bit elements (E = 640: 5 x 16 bytes per lane, every
// lane busy, 4 rows per wave instruction and per reduction step); LPR = 64 is one row per wave.
template <typename XT, typename RT, typename WT, typename MT, int VEC, int ITERS, int LPR = 64>
__global__ __launch_bounds__(256) void add_norm_kernel(const zigma_norm_params_t p) {
    const int lane = (threadIdx.x & 63) % LPR;
    const int64_t r = (static_cast<int64_t>(blockIdx.x) * 4 + (threadIdx.x >> 6)) * (64 / LPR) + (threadIdx.x & 63) / LPR;
    if (r >= p.rows) return;            // (rows % (64 / LPR) == 0 on the multi-row path: whole waves leave together)
    const int b = static_cast<int>(r / p.rows_per_batch);
    const int cols = p.cols;

    float v[ITERS][VEC];
    float sum = 0.f, sq = 0.f;
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c = (it * LPR + lane) * VEC;
        if (c < cols) {
            float x[VEC];
            loadv<XT, VEC>(p.x, r * p.x_row_stride + c, x);
            if (p.branch) {
                float br[VEC], g[VEC];
                loadv<XT, VEC>(p.branch, r * p.branch_row_stride + c, br);
                loadv<MT, VEC>(p.gate, b * p.mod_batch_stride + c, g);
#pragma unroll
                for (int i = 0; i < VEC; ++i) x[i] = rnd<XT>(x[i] + g[i] * br[i]);
                if (p.x_out) {
                    storev<XT, VEC>(p.x_out, r * p.x_out_row_stride + c, x);
                }
            }
            if (p.residual) {
                float rs[VEC];
                loadv<RT, VEC>(p.residual, r * p.res_row_stride + c, rs);
#pragma unroll
                for (int i = 0; i < VEC; ++i) x[i] += rs[i];
            }
            if (p.residual_out) {
                storev<RT, VEC>(p.residual_out, r * p.res_out_row_stride + c, x);
            }
#pragma unroll
            for (int i = 0; i < VEC; ++i) { v[it][i] = x[i]; sum += x[i]; sq += x[i] * x[i]; }
        } else {
#pragma unroll
            for (int i = 0; i < VEC; ++i) v[it][i] = 0.f;
        }
    }
    float mean = 0.f, rstd;
    if (p.is_rms) {
        rstd = rsqrtf(wave_sum<LPR>(sq) / cols + p.eps);
    } else {  // two-pass variance on the register copy, like the Triton kernel (layernorm.py:102-107)
        mean = wave_sum<LPR>(sum) / cols;
        float var = 0.f;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = (it * LPR + lane) * VEC;
            if (c < cols) {
#pragma unroll
                for (int i = 0; i < VEC; ++i) { const float dlt = v[it][i] - mean; var += dlt * dlt; }
            }
        }
        rstd = rsqrtf(wave_sum<LPR>(var) / cols + p.eps);
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c = (it * LPR + lane) * VEC;
        if (c < cols) {
            float y[VEC], w[VEC], bs[VEC];
#pragma unroll
            for (int i = 0; i < VEC; ++i) { w[i] = 1.f; bs[i] = 0.f; }
            if (p.weight) loadv<WT, VEC>(p.weight, c, w);
            if (p.bias) loadv<WT, VEC>(p.bias, c, bs);
#pragma unroll
            for (int i = 0; i < VEC; ++i) y[i] = (v[it][i] - mean) * rstd * w[i] + bs[i];
            if (p.y_out) {
                storev<XT, VEC>(p.y_out, r * p.y_row_stride + c, y);
            }
            if (p.shift) {
                float sh[VEC], sc[VEC];
                loadv<MT, VEC>(p.shift, b * p.mod_batch_stride + c, sh);
                loadv<MT, VEC>(p.scale, b * p.mod_batch_stride + c, sc);
#pragma unroll
                for (int i = 0; i < VEC; ++i) y[i] = rnd<XT>(y[i]) * (1.f + sc[i]) + sh[i];
                storev<XT, VEC>(p.y_mod, r * p.y_mod_row_stride + c, y);
            }
        }
    }
}

static bool al(const void *ptr, size_t a) { return ptr == nullptr || reinterpret_cast<uintptr_t>(ptr) % a == 0; }

template <typename XT, typename RT, typename WT, typename MT>
static int launch_norm(const zigma_norm_params_t &p, hipStream_t stream) {
    constexpr size_t xs = sizeof(typename XT::raw), rs = sizeof(typename RT::raw), ws = sizeof(typename WT::raw),
                     ms = sizeof(typename MT::raw);
    const bool vec = p.cols % 4 == 0 && p.x_row_stride % 4 == 0 && p.branch_row_stride % 4 == 0 && p.x_out_row_stride % 4 == 0 &&
                     p.res_row_stride % 4 == 0 && p.res_out_row_stride % 4 == 0 && p.y_row_stride % 4 == 0 &&
                     p.y_mod_row_stride % 4 == 0 && p.mod_batch_stride % 4 == 0 && al(p.x, 4 * xs) && al(p.branch, 4 * xs) &&
                     al(p.x_out, 4 * xs) && al(p.y_out, 4 * xs) && al(p.y_mod, 4 * xs) && al(p.residual, 4 * rs) &&
                     al(p.residual_out, 4 * rs) && al(p.weight, 4 * ws) && al(p.bias, 4 * ws) && al(p.gate, 4 * ms) &&
                     al(p.shift, 4 * ms) && al(p.scale, 4 * ms);
    const bool vec8 = vec && p.cols % 8 == 0 && p.x_row_stride % 8 == 0 && p.branch_row_stride % 8 == 0 && p.x_out_row_stride % 8 == 0 &&
                      p.res_row_stride % 8 == 0 && p.res_out_row_stride % 8 == 0 && p.y_row_stride % 8 == 0 &&
                      p.y_mod_row_stride % 8 == 0 && p.mod_batch_stride % 8 == 0 && al(p.x, 8 * xs) && al(p.branch, 8 * xs) &&
                      al(p.x_out, 8 * xs) && al(p.y_out, 8 * xs) && al(p.y_mod, 8 * xs) && al(p.residual, 16) &&
                      al(p.residual_out, 16) && al(p.weight, 8 * ws) && al(p.bias, 8 * ws) && al(p.gate, 8 * ms) &&
                      al(p.shift, 8 * ms) && al(p.scale, 8 * ms) && xs == 2;
    dim3 grid(static_cast<unsigned>((static_cast<int64_t>(p.rows) + 3) / 4)), block(256);
#define ZIGMA_NORM(V_, I_) hipLaunchKernelGGL((add_norm_kernel<XT, RT, WT, MT, V_, I_>), grid, block, 0, stream, p)
    // four rows per wave where a row is light (16-bit tensors only: 32 us against 44 for the pre-attention LayerNorm of the block);
    // with the fp32 residual stream in and out a row is 5-6 KB and one row per wave is the faster form (117 against 121 us)
    const bool light_rows = (p.residual == nullptr && p.residual_out == nullptr) || rs == 2;
    if (vec8 && light_rows && p.cols % 128 == 0 && p.cols <= 128 * 8 && p.rows % 4 == 0 && !(p.flags & 1)) {
        const dim3 grid4(static_cast<unsigned>((static_cast<int64_t>(p.rows) / 4 + 3) / 4));
#define ZIGMA_NORM4(I_) hipLaunchKernelGGL((add_norm_kernel<XT, RT, WT, MT, 8, I_, 16>), grid4, block, 0, stream, p)
        switch (p.cols / 128) {
            case 1: ZIGMA_NORM4(1); break;
            case 2: ZIGMA_NORM4(2); break;
            case 3: ZIGMA_NORM4(3); break;
            case 4: ZIGMA_NORM4(4); break;
            case 5: ZIGMA_NORM4(5); break;
            case 6: ZIGMA_NORM4(6); break;
            case 7: ZIGMA_NORM4(7); break;
            default: ZIGMA_NORM4(8); break;
        }
#undef ZIGMA_NORM4
        set_last_kernel("add_norm_v8x4");
        return check_launch();
    }
    if (vec8 && p.cols <= 512 * 2) ZIGMA_NORM(8, 2);       // 16-byte accesses for 16-bit activations
    else if (vec && p.cols <= 256 * 4) ZIGMA_NORM(4, 4);
    else if (vec && p.cols <= 256 * 16) ZIGMA_NORM(4, 16);
    else if (p.cols <= 64 * 16) ZIGMA_NORM(1, 16);
    else if (p.cols <= 64 * 64) ZIGMA_NORM(1, 64);
    else return ZIGMA_ERR_SHAPE;
#undef ZIGMA_NORM
    set_last_kernel(vec ? "add_norm_v4" : "add_norm_v1");
    return check_launch();
}

}  // namespace zigma

using namespace zigma;

extern "C" int zigma_add_norm_fwd(const zigma_norm_params_t *pp, void *stream_) {
    if (!pp) return ZIGMA_ERR_NULL;
    (void)hipGetLastError();  // a stale error of an unrelated earlier call is not ours to report
    const zigma_norm_params_t &p = *pp;
    if (!p.x) return ZIGMA_ERR_NULL;
    if ((p.branch == nullptr) != (p.gate == nullptr)) return ZIGMA_ERR_NULL;
    if ((p.shift == nullptr) != (p.scale == nullptr)) return ZIGMA_ERR_NULL;
    if (p.shift && !p.y_mod) return ZIGMA_ERR_NULL;
    if (!p.y_out && !p.y_mod) return ZIGMA_ERR_NULL;
    if (p.rows < 0 || p.cols < 1 || p.rows_per_batch < 1) return ZIGMA_ERR_SHAPE;
    if (p.flags & ~1) return ZIGMA_ERR_UNSUPPORTED;          // 1: one row per wave even where four fit (A/B probe)
    if (p.rows == 0) return ZIGMA_OK;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    // residual stream is f32 or the activation dtype; weights f32 or activation dtype; keep the
    // instantiation set small: (x, res, w, mod) in {(T,F32|T,F32|T,T)}
    ZIGMA_DISPATCH_DTYPE(p.x_dtype, XT, {
        if (p.mod_dtype != p.x_dtype && (p.gate || p.shift)) return ZIGMA_ERR_DTYPE;
        const bool res32 = p.res_dtype == ZIGMA_F32, w32 = p.w_dtype == ZIGMA_F32;
        if (!res32 && p.res_dtype != p.x_dtype) return ZIGMA_ERR_DTYPE;
        if (!w32 && p.w_dtype != p.x_dtype) return ZIGMA_ERR_DTYPE;
        if (res32 && w32) return launch_norm<XT, F32, F32, XT>(p, stream);
        if (res32) return launch_norm<XT, F32, XT, XT>(p, stream);
        if (w32) return launch_norm<XT, XT, F32, XT>(p, stream);
        return launch_norm<XT, XT, XT, XT>(p, stream);
    })
    return ZIGMA_ERR_DTYPE;
}
