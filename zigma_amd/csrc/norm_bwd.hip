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

template <typename XT, typename RT, typename WT, int ITERS>
__global__ __launch_bounds__(64 * kNbWaves) void add_norm_bwd_kernel(const zigma_norm_bwd_params_t p, float *ws) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int gw = blockIdx.x * kNbWaves + wave, n_gw = gridDim.x * kNbWaves;
    const int cols = p.cols;
    float dw[ITERS], db[ITERS], w[ITERS];
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c = it * 64 + lane;
        dw[it] = 0.f; db[it] = 0.f;
        w[it] = (p.weight && c < cols) ? ld<WT>(p.weight, c) : 1.f;
    }
    for (int64_t r = gw; r < p.rows; r += n_gw) {
        float s[ITERS], dy[ITERS];
        float sum = 0.f, sq = 0.f;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = it * 64 + lane;
            s[it] = 0.f; dy[it] = 0.f;
            if (c < cols) {
                s[it] = ld<RT>(p.xsum, r * p.xsum_row_stride + c);
                dy[it] = ld<XT>(p.dy, r * p.dy_row_stride + c);
            }
            sum += s[it]; sq += s[it] * s[it];
        }
        float mean = 0.f, rstd;
        if (p.is_rms) {
            rstd = rsqrtf(nb_wave_sum(sq) / cols + p.eps);
        } else {
            mean = nb_wave_sum(sum) / cols;
            float var = 0.f;
#pragma unroll
            for (int it = 0; it < ITERS; ++it) {
                const int c = it * 64 + lane;
                if (c < cols) { const float d = s[it] - mean; var += d * d; }
            }
            rstd = rsqrtf(nb_wave_sum(var) / cols + p.eps);
        }
        float c1 = 0.f, c2 = 0.f;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = it * 64 + lane;
            const float xhat = c < cols ? (s[it] - mean) * rstd : 0.f;
            const float wdy = dy[it] * w[it];
            c1 += xhat * wdy; c2 += wdy;
            dw[it] += dy[it] * xhat; db[it] += dy[it];
            s[it] = xhat;
            dy[it] = wdy;
        }
        c1 = nb_wave_sum(c1) / cols;
        c2 = p.is_rms ? 0.f : nb_wave_sum(c2) / cols;
#pragma unroll
        for (int it = 0; it < ITERS; ++it) {
            const int c = it * 64 + lane;
            if (c < cols) {
                float ds = (dy[it] - s[it] * c1 - c2) * rstd;
                if (p.dresidual_out) ds += ld<RT>(p.dresidual_out, r * p.dres_out_row_stride + c);
                if (p.dx) st<XT>(p.dx, r * p.dx_row_stride + c, ds);
                if (p.dresidual) st<RT>(p.dresidual, r * p.dres_row_stride + c, ds);
            }
        }
    }
#pragma unroll
    for (int it = 0; it < ITERS; ++it) {
        const int c = it * 64 + lane;
        if (c < cols) {
            ws[(static_cast<int64_t>(gw) * 2 + 0) * cols + c] = dw[it];
            ws[(static_cast<int64_t>(gw) * 2 + 1) * cols + c] = db[it];
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
#define ZIGMA_NB(I_) hipLaunchKernelGGL((add_norm_bwd_kernel<XT, RT, WT, I_>), dim3(grid), dim3(64 * kNbWaves), 0, stream, p, ws)
    if (p.cols <= 64 * 4) ZIGMA_NB(4);
    else if (p.cols <= 64 * 12) ZIGMA_NB(12);
    else if (p.cols <= 64 * 32) ZIGMA_NB(32);
    else return ZIGMA_ERR_SHAPE;
#undef ZIGMA_NB
    hipLaunchKernelGGL(add_norm_bwd_finish, dim3(2 * ((p.cols + 63) / 64)), dim3(256), 0, stream, p, ws, grid * kNbWaves);
    set_last_kernel("add_norm_bwd");
    return check_launch();
}

}  // namespace zigma

using namespace zigma;

extern "C" int64_t zigma_add_norm_bwd_workspace_bytes(const zigma_norm_bwd_params_t *p) {
    if (!p || p->rows <= 0 || p->cols <= 0) return 0;
    return static_cast<int64_t>(nb_grid(*p)) * kNbWaves * 2 * p->cols * static_cast<int64_t>(sizeof(float));
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
