// RMSNorm / LayerNorm (+ residual / prenorm) BACKWARD, gfx950.  C ABI: zigma_add_norm_bwd.
//
// Replaces the Triton kernel _layer_norm_bwd_kernel and its host _layer_norm_bwd (reference
// dis_mamba/mamba_ssm/ops/triton/layernorm.py:196-377).  With s = the tensor that was normalised (x, or x + residual =
// the forward's residual_out), xhat = (s - mean) * rstd (mean = 0 for RMSNorm), wdy = dy * weight:
//     c1 = mean(xhat * wdy),  c2 = mean(wdy)  (LayerNorm only)
//     ds = (wdy - xhat * c1 - c2) * rstd + dresidual_out          ->  dx (x dtype) and dresidual (residual dtype)
//     dweight = sum_rows dy * xhat,   dbias = sum_rows dy
// mean / rstd are recomputed from s (one extra pass over registers, no extra HBM traffic: the row is read once).
// One wave per row at a time, each wave walks rows with a fixed stride and keeps its partial dweight / dbias in
// registers; the partials go to the workspace and a finishing kernel adds them in a fixed order (the reference
// does the same with one partial per SM, layernorm.py:330-347).
#include "zigma_common.h"

namespace zigma {

__device__ __forceinline__ float nb_wave_sum(float v) {
#pragma unroll
    for (int s = 32; s > 0; s >>= 1) v += __shfl_xor(v, s, 64);
    return v;
}

constexpr int kNbWaves = 4, kNbMaxWg = 512;

// VEC consecutive columns per lane per iteration (8-16 byte accesses when VEC = 4)
template <typename T, int VEC> __device__ __forceinline__ void nb_load(const void *base, int64_t idx, float (&f)[VEC]) {
    if constexpr (VEC == 1) {
        f[0] = ld<T>(base, idx);
    } else if constexpr (T::id == ZIGMA_F32) {
        const uint4 r = *reinterpret_cast<const uint4 *>(reinterpret_cast<const float *>(base) + idx);
        f[0] = __uint_as_float(r.x); f[1] = __uint_as_float(r.y); f[2] = __uint_as_float(r.z); f[3] = __uint_as_float(r.w);
    } else {
        const uint2 r = *reinterpret_cast<const uint2 *>(reinterpret_cast<const uint16_t *>(base) + idx);
        f[0] = to_float<T>(r.x & 0xffffu); f[1] = to_float<T>(r.x >> 16);
        f[2] = to_float<T>(r.y & 0xffffu); f[3] = to_float<T>(r.y >> 16);
    }
}
template <typename T, int VEC> __device__ __forceinline__ void nb_store(void *base, int64_t idx, const float (&f)[VEC]) {
    if constexpr (VEC == 1) {
        st<T>(base, idx, f[0]);
    } else if constexpr (T::id == ZIGMA_F32) {
        *reinterpret_cast<uint4 *>(reinterpret_cast<float *>(base) + idx) =
            make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3]));
    } else {
        *reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(base) + idx) =
            make_uint2(from_float<T>(f[0]) | (uint32_t(from_float<T>(f[1])) << 16),
                       from_float<T>(f[2]) | (uint32_t(from_float<T>(f[3])) << 16));
    }
}

template <typename XT, typename RT, typename WT, int VEC, int ITERS>
__global__ __launch_bounds__(64 * kNbWaves) void add_norm_bwd_kernel(const zigma_norm_bwd_params_t p, float *ws) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gw = blockIdx.x * kNbWaves + wave, n_gw = gridDim.x * kNbWaves;
    const int cols = p.cols;
    float dw[ITERS][VEC], db[ITERS][VEC], w[ITERS][VEC];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c = (it * 64 + lane) * VEC;
#pragma unroll
        for (int v = 0; v < VEC; ++v) { dw[it][v] = 0.f; db[it][v] = 0.f; w[it][v] = 1.f; }
        if (p.weight && c < cols) nb_load<WT, VEC>(p.weight, c, w[it]);
    }
    for (int64_t r = gw; r < p.rows; r += n_gw) {
        float s[ITERS][VEC], dy[ITERS][VEC];
        float sum = 0.f, sq = 0.f;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = (it * 64 + lane) * VEC;
#pragma unroll
            for (int v = 0; v < VEC; ++v) { s[it][v] = 0.f; dy[it][v] = 0.f; }
            if (c < cols) {
                nb_load<RT, VEC>(p.xsum, r * p.xsum_row_stride + c, s[it]);
                nb_load<XT, VEC>(p.dy, r * p.dy_row_stride + c, dy[it]);
            }
#pragma unroll
            for (int v = 0; v < VEC; ++v) { sum += s[it][v]; sq += s[it][v] * s[it][v]; }
        }
        float mean = 0.f, rstd;
        if (p.is_rms) {
            rstd = rsqrtf(nb_wave_sum(sq) / cols + p.eps);
        } else {
            mean = nb_wave_sum(sum) / cols;
            float var = 0.f;
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int c = (it * 64 + lane) * VEC;
                if (c < cols) {
#pragma unroll
                    for (int v = 0; v < VEC; ++v) { const float d = s[it][v] - mean; var += d * d; }
                }
            }
            rstd = rsqrtf(nb_wave_sum(var) / cols + p.eps);
        }
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = (it * 64 + lane) * VEC;
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const float xhat = c < cols ? (s[it][v] - mean) * rstd : 0.f;
                const float wdy = dy[it][v] * w[it][v];
                c1 += xhat * wdy; c2 += wdy;
                dw[it][v] += dy[it][v] * xhat; db[it][v] += dy[it][v];
                s[it][v] = xhat;
                dy[it][v] = wdy;
            }
        }
        c1 = nb_wave_sum(c1) / cols;
        c2 = p.is_rms ? 0.f : nb_wave_sum(c2) / cols;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = (it * 64 + lane) * VEC;
            if (c < cols) {
                float ds[VEC];
#pragma unroll
                for (int v = 0; v < VEC; ++v) ds[v] = (dy[it][v] - s[it][v] * c1 - c2) * rstd;
                if (p.dresidual_out) {
                    float dr[VEC];
                    nb_load<RT, VEC>(p.dresidual_out, r * p.dres_out_row_stride + c, dr);
#pragma unroll
                    for (int v = 0; v < VEC; ++v) ds[v] += dr[v];
                }
                if (p.dx) nb_store<XT, VEC>(p.dx, r * p.dx_row_stride + c, ds);
                if (p.dresidual) nb_store<RT, VEC>(p.dresidual, r * p.dres_row_stride + c, ds);
            }
        }
    }
    // fold the 4 waves' partials through LDS: one partial per WORKGROUP goes to the workspace
    __shared__ float s_fold[kNbWaves][2][64 * VEC];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c = (it * 64 + lane) * VEC;
        __syncthreads();
#pragma unroll
        for (int v = 0; v < VEC; ++v) {
            s_fold[wave][0][lane * VEC + v] = dw[it][v];
            s_fold[wave][1][lane * VEC + v] = db[it][v];
        }
        __syncthreads();
        if (wave < 2 && c < cols) {
#pragma unroll
            for (int v = 0; v < VEC; ++v) {
                const int q = lane * VEC + v;
                ws[(static_cast<int64_t>(blockIdx.x) * 2 + wave) * cols + c + v] =
                    (s_fold[0][wave][q] + s_fold[1][wave][q]) + (s_fold[2][wave][q] + s_fold[3][wave][q]);
            }
        }
    }
}

// one block per 64 columns of dweight or dbias: 4 waves each add every 4th partial (coalesced 256-byte reads), then fold
__global__ __launch_bounds__(256) void add_norm_bwd_finish(const zigma_norm_bwd_params_t p, const float *ws, int n_parts) {
    __shared__ float s_acc[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int blocks_per = (p.cols + 63) / 64;
    const int which = blockIdx.x / blocks_per, c = (blockIdx.x % blocks_per) * 64 + lane;
    float acc = 0.f;
    if (c < p.cols)
        for (int q = wave; q < n_parts; q += 4) acc += ws[(static_cast<int64_t>(q) * 2 + which) * p.cols + c];
    s_acc[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && c < p.cols) {
        acc = (s_acc[0][lane] + s_acc[1][lane]) + (s_acc[2][lane] + s_acc[3][lane]);
        if (which == 0) { if (p.dweight) p.dweight[c] = acc; }
        else if (p.dbias) p.dbias[c] = acc;
    }
}

static int nb_grid(const zigma_norm_bwd_params_t &p) {
    const int64_t wg = (static_cast<int64_t>(p.rows) + kNbWaves - 1) / kNbWaves;
    return static_cast<int>(wg < kNbMaxWg ? (wg < 1 ? 1 : wg) : kNbMaxWg);
}

template <typename XT, typename RT, typename WT>
static int launch_norm_bwd(const zigma_norm_bwd_params_t &p, hipStream_t stream) {
    const int grid = nb_grid(p);
    float *ws = reinterpret_cast<float *>(p.workspace);
    constexpr size_t xs = sizeof(typename XT::raw), rs = sizeof(typename RT::raw), wsz = sizeof(typename WT::raw);
    auto al = [](const void *q, size_t a) { return q == nullptr || reinterpret_cast<uintptr_t>(q) % a == 0; };
    const bool vec = p.cols % 4 == 0 && p.xsum_row_stride % 4 == 0 && p.dy_row_stride % 4 == 0 && p.dres_out_row_stride % 4 == 0 &&
                     p.dx_row_stride % 4 == 0 && p.dres_row_stride % 4 == 0 && al(p.xsum, 4 * rs) && al(p.dy, 4 * xs) &&
                     al(p.dresidual_out, 4 * rs) && al(p.dx, 4 * xs) && al(p.dresidual, 4 * rs) && al(p.weight, 4 * wsz);
#define ZIGMA_NB(V_, I_) hipLaunchKernelGGL((add_norm_bwd_kernel<XT, RT, WT, V_, I_>), dim3(grid), dim3(64 * kNbWaves), 0, stream, p, ws)
    if (vec && p.cols <= 256 * 3) ZIGMA_NB(4, 3);
    else if (vec && p.cols <= 256 * 8) ZIGMA_NB(4, 8);
    else if (p.cols <= 64 * 4) ZIGMA_NB(1, 4);
    else if (p.cols <= 64 * 12) ZIGMA_NB(1, 12);
    else if (p.cols <= 64 * 32) ZIGMA_NB(1, 32);
    else return ZIGMA_ERR_SHAPE;
#undef ZIGMA_NB
    hipLaunchKernelGGL(add_norm_bwd_finish, dim3(2 * ((p.cols + 63) / 64)), dim3(256), 0, stream, p, ws, grid);
    set_last_kernel("add_norm_bwd");
    return check_launch();
}

}  // namespace zigma

using namespace zigma;

extern "C" int64_t zigma_add_norm_bwd_workspace_bytes(const zigma_norm_bwd_params_t *p) {
    if (!p || p->rows <= 0 || p->cols <= 0) return 0;
    return static_cast<int64_t>(nb_grid(*p)) * 2 * p->cols * static_cast<int64_t>(sizeof(float));
}

extern "C" int zigma_add_norm_bwd(const zigma_norm_bwd_params_t *pp, void *stream_) {
    if (!pp) return ZIGMA_ERR_NULL;
    (void)hipGetLastError();
    const zigma_norm_bwd_params_t &p = *pp;
    if (p.rows < 0 || p.cols < 1) return ZIGMA_ERR_SHAPE;
    if (p.flags != 0) return ZIGMA_ERR_UNSUPPORTED;
    if (p.rows == 0) return ZIGMA_OK;
    if (!p.xsum || !p.dy || (!p.dx && !p.dresidual)) return ZIGMA_ERR_NULL;
    if (!p.workspace || p.workspace_bytes < zigma_add_norm_bwd_workspace_bytes(pp)) return ZIGMA_ERR_NULL;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    ZIGMA_DISPATCH_DTYPE(p.x_dtype, XT, {
        const bool res32 = p.res_dtype == ZIGMA_F32, w32 = p.w_dtype == ZIGMA_F32;
        if (!res32 && p.res_dtype != p.x_dtype) return ZIGMA_ERR_DTYPE;
        if (!w32 && p.w_dtype != p.x_dtype) return ZIGMA_ERR_DTYPE;
        if (res32 && w32) return launch_norm_bwd<XT, F32, F32>(p, stream);
        if (res32) return launch_norm_bwd<XT, F32, XT>(p, stream);
        if (w32) return launch_norm_bwd<XT, XT, F32>(p, stream);
        return launch_norm_bwd<XT, XT, XT>(p, stream);
    })
    return ZIGMA_ERR_DTYPE;
}
