// Depthwise causal conv1d (+bias, +SiLU) BACKWARD, token-major operands, gfx950.  C ABI: zigma_causal_conv1d_bwd.
//
// Replaces causal_conv1d_bwd_kernel / causal_conv1d_channellast_bwd_kernel (reference
// dis_causal_conv1d/csrc/causal_conv1d_bwd.cu:46-240,301-470).  With x' the (optionally gathered) input sequence:
//     pre[k]  = bias + sum_t w[t] x'[k - (W-1-t)]                          (recomputed, not stored)
//     dpre[k] = dout[k] * sigmoid(pre) (1 + pre (1 - sigmoid(pre)))        (SiLU) | dout[k]
//     dx'[k]  = sum_t w[t] dpre[k + (W-1-t)]          dw[t] = sum_{b,k} dpre[k] x'[k - (W-1-t)]        db = sum dpre
// HBM streaming like the forward: a lane owns 4 adjacent channels; a wave walks a SEGMENT of the sequence in tiles
// of 16 positions (all 16+2(W-1) x' rows and 16+(W-1) dout rows of a tile in flight before the first FMA) and keeps
// its partial dw / db in registers; one partial per (sample, segment) goes to the workspace and a finishing kernel
// adds them in a fixed order (the reference uses float atomics: not reproducible).
// dx is scattered through the same row table the forward gathered with (a permutation): dx[row[k]] = dx'[k].
#include "conv_helpers.h"

namespace zigma {

constexpr int kCbLT = 16, kCbSeg = 128;   // positions per tile / per wave

template <typename IO, typename WT, int W, bool SILU>
__global__ __launch_bounds__(64) void conv_bwd_tok_kernel(const zigma_conv_bwd_params_t p, float *ws) {
    using P = typename Pack<IO, 4>::type;
    constexpr int ES = static_cast<int>(sizeof(typename IO::raw)), LT = kCbLT;
    constexpr int NX = LT + 2 * (W - 1), ND = LT + (W - 1);       // x' rows k0-(W-1) .. k0+LT+W-2, dout rows k0 .. k0+LT+W-2
    static_assert(NX <= 64, "row table of a tile must fit one wave");
    const int lane = threadIdx.x;
    const int c0 = (blockIdx.x * 64 + lane) * 4;
    const int seg = blockIdx.y, b = blockIdx.z, L = p.seqlen;
    const bool live = c0 < p.dim;
    const int cc = live ? c0 : 0;

    float w[4][W], bias[4], dw[4][W], db[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < W; ++j) { w[i][j] = ld<WT>(p.weight, (cc + i) * p.weight_c_stride + j * p.weight_width_stride); dw[i][j] = 0.f; }
        bias[i] = p.bias ? ld<WT>(p.bias, cc + i) : 0.f;
        db[i] = 0.f;
    }
    const int64_t span = static_cast<int64_t>(L - 1);
    const int x_ls = static_cast<int>(p.x_l_stride) * ES, g_ls = static_cast<int>(p.dout_l_stride) * ES, o_ls = static_cast<int>(p.dx_l_stride) * ES;
    auto mk = [&](const void *base, int64_t bs, int ls) {
        return __builtin_amdgcn_make_buffer_rsrc(const_cast<typename IO::raw *>(reinterpret_cast<const typename IO::raw *>(base) + b * bs), 0,
                                                  static_cast<int>(span * ls + static_cast<int64_t>(p.dim) * ES), 0x00020000);
    };
    const rsrc_t x_rs = mk(p.x, p.x_batch_stride, x_ls), g_rs = mk(p.dout, p.dout_batch_stride, g_ls), o_rs = mk(p.dx, p.dx_batch_stride, o_ls);
    const unsigned lane_off = static_cast<unsigned>(cc) * ES;
    const int k_seg = seg * kCbSeg, k_end = k_seg + kCbSeg < L ? k_seg + kCbSeg : L;

#pragma unroll 1
    for (int k0 = k_seg; k0 < k_end; k0 += LT) {
        // the sequence this tile belongs to (LT divides reset_period): positions outside [k_lo, k_hi) are another sequence — their x' reads
        // as the zero padding, their dout does not reach this tile's dx
        const int k_lo = p.reset_period > 0 ? (k0 / p.reset_period) * p.reset_period : 0;
        const int k_hi = p.reset_period > 0 && k_lo + p.reset_period < L ? k_lo + p.reset_period : L;
        int rowv;      // lane i <- row of scan position k0 - (W-1) + i   (input gather AND dx scatter table)
        {
            int k = k0 - (W - 1) + lane;
            k = k < 0 ? 0 : (k < L ? k : L - 1);
            rowv = p.x_row_index ? p.x_row_index[k] : k;
        }
        P xr[NX], gr[ND];
#pragma unroll
        for (int i = 0; i < NX; ++i) {
            const int k = k0 - (W - 1) + i;
            const int row = __builtin_amdgcn_readlane(rowv, i);
            xr[i] = P{};
            if (k >= k_lo && k < k_hi) xr[i] = buf_ld4<IO>(x_rs, lane_off, row * x_ls);
        }
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            const int k = k0 + i;
            gr[i] = P{};
            if (k < k_hi) gr[i] = buf_ld4<IO>(g_rs, lane_off, k * g_ls);     // dout is in scan order
        }
        // dpre at positions k0 .. k0+LT+W-2
        float dp[ND][4];
#pragma unroll
        for (int i = 0; i < ND; ++i) {
            float go[4];
            unpack4<IO>(gr[i], go);
            if (SILU) {
                float pre[4] = {bias[0], bias[1], bias[2], bias[3]};
#pragma unroll
                for (int t = 0; t < W; ++t) {
                    float xin[4];
                    unpack4<IO>(xr[i + t], xin);       // x'[k0 + i - (W-1-t)] sits at index i + t
#pragma unroll
                    for (int q = 0; q < 4; ++q) pre[q] += w[q][t] * xin[q];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float sg = fast_rcp(1.f + fast_exp2(-pre[q] * kLog2e));
                    go[q] *= sg * (1.f + pre[q] * (1.f - sg));
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) dp[i][q] = go[q];
        }
#pragma unroll
        for (int j = 0; j < LT; ++j) {
            const int k = k0 + j;
            if (k < L) {
                float o[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int t = 0; t < W; ++t) {
                    float xin[4];
                    unpack4<IO>(xr[j + t], xin);
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        o[q] += w[q][t] * dp[j + (W - 1 - t)][q];
                        dw[q][t] += dp[j][q] * xin[q];
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q) db[q] += dp[j][q];
                const int row = __builtin_amdgcn_readlane(rowv, j + (W - 1));
                if (live) buf_st4<IO>(pack4<IO>(o), o_rs, lane_off, row * o_ls);
            }
        }
    }
    if (live) {
        float *out = ws + ((static_cast<int64_t>(b) * gridDim.y + seg) * p.dim + c0) * (W + 1);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
#pragma unroll
            for (int t = 0; t < W; ++t) out[q * (W + 1) + t] = dw[q][t];
            out[q * (W + 1) + W] = db[q];
        }
    }
}

// one block per 64 (channel, tap) entries: 4 waves each add every 4th partial, then fold
__global__ __launch_bounds__(256) void conv_bwd_finish(const zigma_conv_bwd_params_t p, const float *ws, int n_parts) {
    __shared__ float s_acc[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int W1 = p.width + 1, total = p.dim * W1;
    const int i = blockIdx.x * 64 + lane;                     // (c, t')  t' in [0, W]
    float acc = 0.f;
    if (i < total)
        for (int q = wave; q < n_parts; q += 4) acc += ws[static_cast<int64_t>(q) * total + i];
    s_acc[wave][lane] = acc;
    __syncthreads();
    if (wave == 0 && i < total) {
        acc = (s_acc[0][lane] + s_acc[1][lane]) + (s_acc[2][lane] + s_acc[3][lane]);
        const int c = i / W1, t = i % W1;
        if (t < p.width) p.dweight[c * p.width + t] = acc;
        else if (p.dbias) p.dbias[c] = acc;
    }
}

template <typename IO, typename WT>
static void launch_conv_bwd(const zigma_conv_bwd_params_t &p, int n_seg, hipStream_t stream) {
    dim3 grid((p.dim / 4 + 63) / 64, n_seg, p.batch), block(64);
    float *ws = reinterpret_cast<float *>(p.workspace);
#define ZIGMA_CB(W_)                                                                                                 \
    if (p.silu_activation) hipLaunchKernelGGL((conv_bwd_tok_kernel<IO, WT, W_, true>), grid, block, 0, stream, p, ws); \
    else hipLaunchKernelGGL((conv_bwd_tok_kernel<IO, WT, W_, false>), grid, block, 0, stream, p, ws);
    switch (p.width) {
        case 2: ZIGMA_CB(2) break;
        case 3: ZIGMA_CB(3) break;
        default: ZIGMA_CB(4) break;
    }
#undef ZIGMA_CB
}

}  // namespace zigma

using namespace zigma;

extern "C" int64_t zigma_causal_conv1d_bwd_workspace_bytes(const zigma_conv_bwd_params_t *p) {
    if (!p || p->batch <= 0 || p->dim <= 0 || p->seqlen <= 0) return 0;
    const int64_t n_seg = (p->seqlen + kCbSeg - 1) / kCbSeg;
    return static_cast<int64_t>(p->batch) * n_seg * p->dim * (p->width + 1) * static_cast<int64_t>(sizeof(float));
}

extern "C" int zigma_causal_conv1d_bwd(const zigma_conv_bwd_params_t *pp, void *stream_) {
    if (!pp) return ZIGMA_ERR_NULL;
    (void)hipGetLastError();
    const zigma_conv_bwd_params_t &p = *pp;
    if (p.width < 2 || p.width > 4) return ZIGMA_ERR_SHAPE;
    if (p.batch < 0 || p.dim < 0 || p.seqlen < 0 || p.batch > 65535) return ZIGMA_ERR_SHAPE;
    if (p.flags != 0) return ZIGMA_ERR_UNSUPPORTED;
    if (p.reset_period < 0 || p.reset_period % kCbLT != 0) return ZIGMA_ERR_SHAPE;
    if (p.batch == 0 || p.dim == 0 || p.seqlen == 0) return ZIGMA_OK;
    if (!p.x || !p.weight || !p.dout || !p.dx || !p.dweight) return ZIGMA_ERR_NULL;
    if (p.bias && !p.dbias) return ZIGMA_ERR_NULL;
    if (!p.workspace || p.workspace_bytes < zigma_causal_conv1d_bwd_workspace_bytes(pp)) return ZIGMA_ERR_NULL;
    const int es = p.io_dtype == ZIGMA_F32 ? 4 : 2;
    auto bad = [&](const void *q, int64_t ls, int64_t bs) {
        return reinterpret_cast<uintptr_t>(q) % (4 * es) != 0 || ls % 4 != 0 || bs % 4 != 0 || ls < 0 ||
               (ls * p.seqlen + p.dim) * static_cast<int64_t>(es) >= (int64_t(1) << 31);
    };
    if (p.dim % 4 != 0 || bad(p.x, p.x_l_stride, p.x_batch_stride) || bad(p.dout, p.dout_l_stride, p.dout_batch_stride) ||
        bad(p.dx, p.dx_l_stride, p.dx_batch_stride))
        return ZIGMA_ERR_STRIDE;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int n_seg = (p.seqlen + kCbSeg - 1) / kCbSeg;
    ZIGMA_DISPATCH_DTYPE(p.io_dtype, IO, {
        ZIGMA_DISPATCH_DTYPE(p.w_dtype, WT, { launch_conv_bwd<IO, WT>(p, n_seg, stream); })
    })
    const int n = p.dim * (p.width + 1);
    hipLaunchKernelGGL(conv_bwd_finish, dim3((n + 63) / 64), dim3(256), 0, stream, p, reinterpret_cast<const float *>(p.workspace),
                       n_seg * p.batch);
    set_last_kernel("conv_bwd_tok");
    return check_launch();
}
