// dt_proj + bias + softplus on the matrix cores, gfx950.  C ABI: zigma_dt_proj_softplus_fwd.
//
// Replaces  delta = delta_proj_weight @ x_dbl[:, :dt_rank].T  (reference selective_scan_interface.py:323, a cuBLAS
// GEMM with K = dt_rank = 40) together with the per-element softplus(delta + delta_bias) the reference applies at
// the top of its scan kernel (selective_scan_fwd_kernel.cuh:153-156).  With K that small the product is a
// WRITE-bound streaming kernel (one pass over the (M, Di) output, 168 MB at B=64), so the transcendental work
// rides for free under the stores — and it no longer sits in the VALU-bound scan kernel.
//
//   out[m, d] = softplus20( sum_r x[m, r] * w[d, r] + bias[d] )        (bf16 in, fp32 accumulate, bf16 out)
//
// One wave = 32 tokens x 64 channels: v_mfma_f32_32x32x16_bf16, 3 k-steps (K padded to 48 with zero fragments),
// two accumulators.  The B fragments of the two accumulators hold the EVEN and the ODD channels of the 64-channel
// slab, so lane j ends up with channels 2j and 2j+1 of each token: one v_cvt_pk_bf16_f32 per token and pair; the tile is
// then transposed through a wave-private LDS tile and leaves as 16-byte stores (8 tokens x 128 B per instruction: 16 four-byte
// stores per lane made the store instruction rate the limit, 62-66 -> 59-60 us).  A / B fragments are 16-byte loads straight from global
// (x_dbl is 9 MB, the weight 100 KB: L2 resident), no LDS.
#include "zigma_common.h"

namespace zigma {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kDtTokPerWave = 32, kDtChPerBlock = 64, kDtWaves = 4;
#ifndef ZIGMA_DT_ITERS
#define ZIGMA_DT_ITERS 4
#endif
constexpr int kDtIters = ZIGMA_DT_ITERS;

__global__ __launch_bounds__(64 * kDtWaves, 5) void dt_proj_softplus_kernel(const zigma_dtproj_params_t p) {
    __shared__ __attribute__((aligned(16))) unsigned char s_tile[kDtWaves * kDtTokPerWave * 144];
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int d0 = blockIdx.x * kDtChPerBlock;
    const bool wide_ok = p.out_row_stride % 8 == 0 && reinterpret_cast<uintptr_t>(p.out) % 16 == 0 && !(p.flags & 1);   // 16-byte stores
    const int j = lane & 31, kh = lane >> 5;                   // fragment row / k-half of this lane
    const uint16_t *xw = reinterpret_cast<const uint16_t *>(p.x);
    const uint16_t *ww = reinterpret_cast<const uint16_t *>(p.w);
    uint16_t *ow = reinterpret_cast<uint16_t *>(p.out);
    const float *bias = reinterpret_cast<const float *>(p.bias);

    // 8 consecutive k of one row; k % 8 == 0 (dispatcher), so a fragment is either whole or beyond K (= zero): the
    // address is clamped and the value selected — no divergent branches around the loads
    auto frag = [&](const uint16_t *row, int k0) -> bf16x8 {
        const bool live = k0 < p.k;
        const uint4 v = *reinterpret_cast<const uint4 *>(row + (live ? k0 : 0));
        return __builtin_bit_cast(bf16x8, live ? v : make_uint4(0, 0, 0, 0));
    };
    // B fragments (weights) of this block's channels: even / odd channel per lane, 3 k-steps, loaded once
    bf16x8 be[3], bo[3];
    const uint16_t *we = ww + static_cast<int64_t>(d0 + 2 * j) * p.w_row_stride;
    const uint16_t *wo = we + p.w_row_stride;
#pragma unroll
    for (int s = 0; s < 3; ++s) {
        be[s] = frag(we, s * 16 + kh * 8);
        bo[s] = frag(wo, s * 16 + kh * 8);
    }
    const float b_e = bias ? bias[d0 + 2 * j] : 0.f, b_o = bias ? bias[d0 + 2 * j + 1] : 0.f;

    const int64_t m_blk = static_cast<int64_t>(blockIdx.y) * (kDtTokPerWave * kDtWaves * kDtIters);
    // A fragments of tile `it`: the next tile's are requested before this tile's softplus / stores (their L2 latency otherwise
    // sits between every two tiles of a wave)
    auto a_frags = [&](int it, bf16x8 (&a)[3]) {
        int64_t mr = m_blk + (static_cast<int64_t>(it) * kDtWaves + wave) * kDtTokPerWave + j;
        mr = mr < p.m ? mr : p.m - 1;                            // rows beyond m: clamped loads, never stored
        const uint16_t *xr = xw + mr * p.x_row_stride;
#pragma unroll
        for (int s = 0; s < 3; ++s) a[s] = frag(xr, s * 16 + kh * 8);
    };
    bf16x8 a_cur[3], a_nxt[3];
    a_frags(0, a_cur);
#pragma unroll 1
    for (int it = 0; it < kDtIters; ++it) {
        const int64_t m0 = m_blk + (static_cast<int64_t>(it) * kDtWaves + wave) * kDtTokPerWave;
        if (m0 >= p.m) break;
        if (it + 1 < kDtIters) a_frags(it + 1, a_nxt);
        f32x16 ce = {}, co = {};
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            ce = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[s], be[s], ce, 0, 0, 0);
            co = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a_cur[s], bo[s], co, 0, 0, 0);
        }
        // C/D layout: column = lane & 31 (channel pair j), row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5) (token)
        const bool full = m0 + kDtTokPerWave <= p.m;            // wave-uniform: whole tile inside -> no per-store predicate
        if (full && wide_ok) {
            // 16 four-byte stores per lane make the store INSTRUCTION rate the limit (2560 per CU at the headline shape); the tile goes
            // through a wave-private LDS tile (32 tokens x 128 B, pitch 144) and leaves as 4 sixteen-byte stores per lane
            unsigned char *tile = s_tile + wave * (kDtTokPerWave * 144);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dm = (r & 3) + 8 * (r >> 2) + 4 * kh;
                float ve = ce[r] + b_e, vo = co[r] + b_o;
                if (p.softplus) { ve = softplus20_r16(ve); vo = softplus20_r16(vo); }
                *reinterpret_cast<uint32_t *>(tile + dm * 144 + j * 4) =
                    static_cast<uint32_t>(from_float<BF16>(ve)) | (static_cast<uint32_t>(from_float<BF16>(vo)) << 16);
            }
            uint16_t *orow8 = ow + (m0 + (lane >> 3)) * p.out_row_stride + d0 + (lane & 7) * 8;
#pragma unroll
            for (int i = 0; i < 4; ++i)
                *reinterpret_cast<uint4 *>(orow8 + static_cast<int64_t>(i * 8) * p.out_row_stride) =
                    *reinterpret_cast<const uint4 *>(tile + (i * 8 + (lane >> 3)) * 144 + (lane & 7) * 16);
        } else {
        uint16_t *orow = ow + (m0 + 4 * kh) * p.out_row_stride + d0 + 2 * j;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dm = (r & 3) + 8 * (r >> 2);
            float ve = ce[r] + b_e, vo = co[r] + b_o;
            if (p.softplus) { ve = softplus20_r16(ve); vo = softplus20_r16(vo); }
            const uint32_t pk = static_cast<uint32_t>(from_float<BF16>(ve)) | (static_cast<uint32_t>(from_float<BF16>(vo)) << 16);
            if (full || m0 + 4 * kh + dm < p.m) *reinterpret_cast<uint32_t *>(orow + dm * p.out_row_stride) = pk;
        }
        }
#pragma unroll
        for (int s = 0; s < 3; ++s) a_cur[s] = a_nxt[s];
    }
}

}  // namespace zigma

using namespace zigma;

extern "C" int zigma_dt_proj_softplus_fwd(const zigma_dtproj_params_t *pp, void *stream_) {
    if (!pp) return ZIGMA_ERR_NULL;
    (void)hipGetLastError();
    const zigma_dtproj_params_t &p = *pp;
    if (p.m < 0 || p.n < 0 || p.k < 1) return ZIGMA_ERR_SHAPE;
    if (p.flags & ~1) return ZIGMA_ERR_UNSUPPORTED;         // 1: four-byte stores as the accumulators lie (A/B probe)
    if (p.m == 0 || p.n == 0) return ZIGMA_OK;
    if (!p.x || !p.w || !p.out) return ZIGMA_ERR_NULL;
    if (p.dtype != ZIGMA_BF16) return ZIGMA_ERR_DTYPE;
    if (p.k > 48 || p.k % 8 != 0 || p.n % kDtChPerBlock != 0) return ZIGMA_ERR_SHAPE;
    // 16-byte fragment loads, 4-byte packed stores
    if (p.x_row_stride % 8 != 0 || p.w_row_stride % 8 != 0 || p.out_row_stride % 2 != 0 ||
        reinterpret_cast<uintptr_t>(p.x) % 16 != 0 || reinterpret_cast<uintptr_t>(p.w) % 16 != 0 ||
        reinterpret_cast<uintptr_t>(p.out) % 4 != 0)
        return ZIGMA_ERR_STRIDE;
    const int64_t tok_per_block = kDtTokPerWave * kDtWaves * kDtIters;
    dim3 grid(p.n / kDtChPerBlock, static_cast<unsigned>((p.m + tok_per_block - 1) / tok_per_block)), block(64 * kDtWaves);
    hipLaunchKernelGGL(dt_proj_softplus_kernel, grid, block, 0, static_cast<hipStream_t>(stream_), p);
    set_last_kernel("dt_proj_softplus_mfma");
    return check_launch();
}
