// Depthwise causal conv1d (+bias, +SiLU) forward for gfx950.  C ABI: zigma_causal_conv1d_fwd.
//
// Replaces causal_conv1d_fwd_kernel / causal_conv1d_channellast_fwd_kernel of the reference
// (dis_causal_conv1d/csrc/causal_conv1d_fwd.cu:39-130,193-298).  HBM-bound (read x once, write out
// once), so the only things that matter are full-line coalesced accesses and enough waves.
//
//  conv_tok_kernel     — token-major (channel contiguous) operands: a lane owns VEC adjacent channels
//      (8-16 B per access), walks LT consecutive scan positions keeping the width-1 previous inputs
//      in registers; the zigzag gather (x_row_index) picks whole rows, so it costs nothing.
//  conv_generic_kernel — any strides (channel-first views of the reference, channel-last, ...): one
//      output element per thread, threads run along the contiguous dimension.
#include "conv_helpers.h"

namespace zigma {

// grid: (ceil(dim / (4*64)), ceil(L / LT), batch); block 64.  A lane owns 4 adjacent channels and LT consecutive
// scan positions.  All LT + W - 1 row loads of the tile are issued before the first FMA (memory-level
// parallelism is what an HBM-bound kernel needs); rows are addressed as buffer descriptor + fixed lane offset +
// scalar row offset, the row table arrives through one vector load and v_readlane.
template <typename IO, typename WT, int W, int LT, bool SILU>
__global__ __launch_bounds__(64) void conv_tok_kernel(const zigma_conv_params_t p) {
    using P = typename Pack<IO, 4>::type;
    constexpr int ES = static_cast<int>(sizeof(typename IO::raw)), NR = LT + W - 1;
    static_assert(NR <= 64, "row table of a tile must fit one wave");
    const int lane = threadIdx.x;
    const int c0 = (blockIdx.x * 64 + lane) * 4;
    const int b = blockIdx.z;
    const int k0 = blockIdx.y * LT;
    const int L = p.seqlen;
    const bool live = c0 < p.dim;          // dead lanes keep running (readlane needs the whole wave), re-read channel 0, never store
    const int cc = live ? c0 : 0;

    float w[4][W], bias[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int j = 0; j < W; ++j) w[i][j] = ld<WT>(p.weight, (cc + i) * p.weight_c_stride + j * p.weight_width_stride);
        bias[i] = p.bias ? ld<WT>(p.bias, cc + i) : 0.f;
    }
    const int64_t span = static_cast<int64_t>(L - 1);
    const int x_ls = static_cast<int>(p.x_l_stride) * ES, o_ls = static_cast<int>(p.out_l_stride) * ES;
    const rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<typename IO::raw *>(reinterpret_cast<const typename IO::raw *>(p.x) + b * p.x_batch_stride), 0,
        static_cast<int>(span * x_ls + static_cast<int64_t>(p.dim) * ES), 0x00020000);
    const rsrc_t o_rs = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<typename IO::raw *>(p.out) + b * p.out_batch_stride, 0,
        static_cast<int>(span * o_ls + static_cast<int64_t>(p.dim) * ES), 0x00020000);
    const unsigned lane_off = static_cast<unsigned>(cc) * ES;

    // row table entries of this tile: lane i <- input row of scan position k0 - (W-1) + i
    int rowv;
    {
        int k = k0 - (W - 1) + lane;
        k = k < 0 ? 0 : (k < L ? k : L - 1);
        rowv = p.x_row_index ? p.x_row_index[k] : k;
    }
    // first position the causal window may reach: 0, or the start of this tile's own sequence (LT divides reset_period)
    const int k_lo = p.reset_period > 0 ? (k0 / p.reset_period) * p.reset_period : 0;
    P raw[NR];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const int k = k0 - (W - 1) + i;                       // wave-uniform
        const int row = __builtin_amdgcn_readlane(rowv, i);
        raw[i] = P{};
        if (k >= k_lo && k < L) raw[i] = buf_ld4<IO>(x_rs, lane_off, row * x_ls);   // zero left padding otherwise
    }
#pragma unroll
    for (int j = 0; j < LT; ++j) {
        const int k = k0 + j;
        if (k < L) {
            float o[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = bias[i];
#pragma unroll
            for (int t = 0; t < W; ++t) {
                float xin[4];
                unpack4<IO>(raw[j + t], xin);
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] += w[i][t] * xin[i];
            }
            if (SILU) {
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = silu(o[i]);
            }
            if (live) buf_st4<IO>(pack4<IO>(o), o_rs, lane_off, k * o_ls);
        }
    }
}

// one output per thread; threads fastest along l when CONTIG_L, along c otherwise.
template <typename IO, typename WT, bool CONTIG_L>
__global__ __launch_bounds__(256) void conv_generic_kernel(const zigma_conv_params_t p) {
    const int64_t idx = static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x;
    const int64_t per_b = static_cast<int64_t>(p.dim) * p.seqlen;
    if (idx >= per_b * p.batch) return;
    const int b = static_cast<int>(idx / per_b);
    const int64_t rem = idx % per_b;
    const int c = CONTIG_L ? static_cast<int>(rem / p.seqlen) : static_cast<int>(rem % p.dim);
    const int l = CONTIG_L ? static_cast<int>(rem % p.seqlen) : static_cast<int>(rem / p.dim);
    float acc = p.bias ? ld<WT>(p.bias, c) : 0.f;
    const int64_t xo = b * p.x_batch_stride + c * p.x_c_stride;
    for (int j = 0; j < p.width; ++j) {
        const int k = l - (p.width - 1 - j);
        if (k >= 0) {
            const int64_t r = p.x_row_index ? p.x_row_index[k] : k;
            acc += ld<WT>(p.weight, c * p.weight_c_stride + j * p.weight_width_stride) * ld<IO>(p.x, xo + r * p.x_l_stride);
        }
    }
    if (p.silu_activation) acc = silu(acc);
    st<IO>(p.out, b * p.out_batch_stride + c * p.out_c_stride + l * p.out_l_stride, acc);
}

template <typename IO, typename WT>
static int launch_conv(const zigma_conv_params_t &p, hipStream_t stream) {
    constexpr size_t es = sizeof(typename IO::raw);
    if (p.batch > 65535) {              // batch rides in gridDim.z: larger batches (video: batch x tokens-per-frame rows) go in slices
        for (int b0 = 0; b0 < p.batch; b0 += 65535) {
            zigma_conv_params_t q = p;
            q.batch = p.batch - b0 < 65535 ? p.batch - b0 : 65535;
            q.x = reinterpret_cast<const char *>(p.x) + static_cast<int64_t>(b0) * p.x_batch_stride * es;
            q.out = reinterpret_cast<char *>(p.out) + static_cast<int64_t>(b0) * p.out_batch_stride * es;
            const int rc = launch_conv<IO, WT>(q, stream);
            if (rc != ZIGMA_OK) return rc;
        }
        return ZIGMA_OK;
    }
    const bool tok = p.x_c_stride == 1 && p.out_c_stride == 1 && p.dim % 4 == 0 &&
                     reinterpret_cast<uintptr_t>(p.x) % (4 * es) == 0 && reinterpret_cast<uintptr_t>(p.out) % (4 * es) == 0 &&
                     p.x_l_stride % 4 == 0 && p.out_l_stride % 4 == 0 && p.x_batch_stride % 4 == 0 && p.out_batch_stride % 4 == 0 &&
                     p.x_l_stride >= 0 && p.out_l_stride >= 0 &&
                     (p.x_l_stride * p.seqlen + p.dim) * static_cast<int64_t>(es) < (int64_t(1) << 31) &&
                     (p.out_l_stride * p.seqlen + p.dim) * static_cast<int64_t>(es) < (int64_t(1) << 31);
    if (p.reset_period < 0 || p.reset_period % 16 != 0) return ZIGMA_ERR_SHAPE;
    if (p.reset_period > 0 && !tok) return ZIGMA_ERR_STRIDE;   // only the token-major kernel restarts sequences
    if (tok) {
        constexpr int LT = 16;
        dim3 grid((p.dim / 4 + 63) / 64, (p.seqlen + LT - 1) / LT, p.batch), block(64);
#define ZIGMA_CONV_TOK(W_)                                                                                          \
    if (p.silu_activation) hipLaunchKernelGGL((conv_tok_kernel<IO, WT, W_, LT, true>), grid, block, 0, stream, p);  \
    else hipLaunchKernelGGL((conv_tok_kernel<IO, WT, W_, LT, false>), grid, block, 0, stream, p);
        switch (p.width) {
            case 2: ZIGMA_CONV_TOK(2) break;
            case 3: ZIGMA_CONV_TOK(3) break;
            default: ZIGMA_CONV_TOK(4) break;
        }
#undef ZIGMA_CONV_TOK
        set_last_kernel("conv_tok");
        return check_launch();
    }
    const int64_t total = static_cast<int64_t>(p.batch) * p.dim * p.seqlen;
    dim3 grid(static_cast<unsigned>((total + 255) / 256)), block(256);
    if (p.x_l_stride == 1) hipLaunchKernelGGL((conv_generic_kernel<IO, WT, true>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((conv_generic_kernel<IO, WT, false>), grid, block, 0, stream, p);
    set_last_kernel("conv_generic");
    return check_launch();
}

}  // namespace zigma

using namespace zigma;

extern "C" int zigma_causal_conv1d_fwd(const zigma_conv_params_t *pp, void *stream_) {
    if (!pp) return ZIGMA_ERR_NULL;
    (void)hipGetLastError();  // a stale error of an unrelated earlier call is not ours to report
    const zigma_conv_params_t &p = *pp;
    if (p.width < 2 || p.width > 4) return ZIGMA_ERR_SHAPE;  // causal_conv1d.cpp:157
    if (p.batch < 0 || p.dim < 0 || p.seqlen < 0) return ZIGMA_ERR_SHAPE;
    if (p.flags != 0) return ZIGMA_ERR_UNSUPPORTED;
    if (p.batch == 0 || p.dim == 0 || p.seqlen == 0) return ZIGMA_OK;  // empty (pointers may be NULL)
    if (!p.x || !p.weight || !p.out) return ZIGMA_ERR_NULL;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    ZIGMA_DISPATCH_DTYPE(p.io_dtype, IO, {
        ZIGMA_DISPATCH_DTYPE(p.w_dtype, WT, { return launch_conv<IO, WT>(p, stream); })
    })
    return ZIGMA_ERR_DTYPE;
}
