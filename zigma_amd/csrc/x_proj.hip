// x_proj: x_dbl = u @ W_x^T for the SKINNY output of the Mamba block (N = dt_rank + 2 d_state = 72, K = d_inner = 1280),
// gfx950.  C ABI: zigma_x_proj_fwd.
//
// Replaces F.linear(conv1d_out, x_proj_weight) of MambaInnerFn.forward (reference selective_scan_interface.py:318-322).
// With N = 72 the product is a READ-bound streaming pass over u (168 MB at B=64; 12 GFLOP): the library's tiled GEMM
// (MT128x256) computes a tile 3.5x wider than the output.  Here:
//   workgroup = 256 tokens = 8 waves x 32 tokens; W_x (zero-padded to 96 rows) is staged through LDS in K-chunks of 256
//   (shared by the 8 waves); every wave streams ITS 32 token rows of u straight from HBM as MFMA A fragments
//   (v_mfma_f32_32x32x16_bf16: lane -> token row, 8 consecutive channels; 4 k-steps deep in flight) and keeps the
//   32 x 96 accumulator in registers; output rows leave as bf16.
// bf16 only; n <= 96; k % 256 == 0; rows 16-byte aligned.
#include "zigma_common.h"

namespace zigma {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int kXpWaves = 8, kXpTok = 32, kXpChunk = 256, kXpRows = 96, kXpDepth = 4;
constexpr int kXpPitch = kXpChunk * 2 + 16;          // bytes per staged weight row (16 B skew)

__global__ __launch_bounds__(64 * kXpWaves) void x_proj_kernel(const zigma_xproj_params_t p) {
    __shared__ __attribute__((aligned(16))) unsigned char s_w[kXpRows * kXpPitch];   // 50.7 KB
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, kh = lane >> 5;
    const uint16_t *xw = reinterpret_cast<const uint16_t *>(p.x);
    const uint16_t *ww = reinterpret_cast<const uint16_t *>(p.w);
    uint16_t *ow = reinterpret_cast<uint16_t *>(p.out);
    const int64_t m0 = (static_cast<int64_t>(blockIdx.x) * kXpWaves + wave) * kXpTok;
    int64_t mr = m0 + j;
    mr = mr < p.m ? mr : p.m - 1;                                 // rows beyond m: clamped loads, no stores
    const uint16_t *xrow = xw + mr * p.x_row_stride + kh * 8;

    f32x16 acc[3];
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) acc[nb] = f32x16{};
    const int n_steps = p.k / 16;                                  // MFMA k-steps over the whole K
    uint4 aq[kXpDepth];                                            // A fragments in flight (k-steps s .. s + depth - 1)
#pragma unroll
    for (int d = 0; d < kXpDepth; ++d) aq[d] = *reinterpret_cast<const uint4 *>(xrow + d * 16);

#pragma unroll 1
    for (int c0 = 0; c0 < p.k; c0 += kXpChunk) {
        __syncthreads();                                           // previous chunk's fragments are consumed
        for (int piece = tid; piece < kXpRows * (kXpChunk / 8); piece += 64 * kXpWaves) {
            const int row = piece >> 5, pc = piece & 31;           // 32 16-byte pieces per row of the chunk
            uint4 v = make_uint4(0, 0, 0, 0);
            if (row < p.n) v = *reinterpret_cast<const uint4 *>(ww + static_cast<int64_t>(row) * p.w_row_stride + c0 + pc * 8);
            *reinterpret_cast<uint4 *>(s_w + row * kXpPitch + pc * 16) = v;
        }
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < kXpChunk / 16; ++ks) {
            const int s = c0 / 16 + ks;
            const bf16x8 a = __builtin_bit_cast(bf16x8, aq[ks % kXpDepth]);
            if (s + kXpDepth < n_steps) aq[ks % kXpDepth] = *reinterpret_cast<const uint4 *>(xrow + (s + kXpDepth) * 16);
#pragma unroll
            for (int nb = 0; nb < 3; ++nb) {
                const bf16x8 b = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(s_w + (nb * 32 + j) * kXpPitch + ks * 32 + kh * 16));
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[nb], 0, 0, 0);
            }
        }
    }
    // C/D layout: column = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) {
        const int n = nb * 32 + j;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t m = m0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (n < p.n && m < p.m) ow[m * p.out_row_stride + n] = from_float<BF16>(acc[nb][r]);
        }
    }
}

// Few tokens (serving-size batches, the video model at B = 2: 8192 tokens): the streaming kernel above is 32 workgroups on 256 CUs and
// every one of them walks the whole K alone (31 us at 8192 x 1536, the library's tiled GEMM 21).  Here K is split over the 8 waves of a
// workgroup of 32 tokens — m / 32 workgroups, each wave k / 8 columns (12 k-steps at k = 1536) — with BOTH operands straight from memory
// as MFMA fragments (W_x is 245 KB: L2-resident, every workgroup reads all of it once), all of a wave's token fragments in flight at once;
// the eight partial 32 x 96 tiles are added through LDS in wave order (a fixed summation order: results do not depend on timing).
constexpr int kXsWaves = 8, kXsMaxSteps = 12;

__global__ __launch_bounds__(64 * kXsWaves) void x_proj_splitk_kernel(const zigma_xproj_params_t p) {
    __shared__ __attribute__((aligned(16))) float s_part[kXsWaves][3][16][64];     // 96 KB: [wave][feature block][accumulator register][lane]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, kh = lane >> 5;
    const uint16_t *xw = reinterpret_cast<const uint16_t *>(p.x);
    const uint16_t *ww = reinterpret_cast<const uint16_t *>(p.w);
    uint16_t *ow = reinterpret_cast<uint16_t *>(p.out);
    const int64_t m0 = static_cast<int64_t>(blockIdx.x) * kXpTok;
    int64_t mr = m0 + j;
    mr = mr < p.m ? mr : p.m - 1;                                  // rows beyond m: clamped loads, no stores
    const int kw = p.k / kXsWaves, steps = kw / 16;                // this wave's k range [wave * kw, + kw): steps <= kXsMaxSteps k-steps
    const uint16_t *xrow = xw + mr * p.x_row_stride + wave * kw + kh * 8;
    const uint16_t *wrow[3];
    bool live[3];
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) {
        const int n = nb * 32 + j;
        live[nb] = n < p.n;
        wrow[nb] = ww + static_cast<int64_t>(live[nb] ? n : 0) * p.w_row_stride + wave * kw + kh * 8;
    }
    uint4 aq[kXsMaxSteps];
#pragma unroll
    for (int s = 0; s < kXsMaxSteps; ++s) aq[s] = s < steps ? *reinterpret_cast<const uint4 *>(xrow + s * 16) : make_uint4(0, 0, 0, 0);
    f32x16 acc[3];
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) acc[nb] = f32x16{};
    uint4 bq[2][3];
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) bq[0][nb] = live[nb] ? *reinterpret_cast<const uint4 *>(wrow[nb]) : make_uint4(0, 0, 0, 0);
#pragma unroll
    for (int s = 0; s < kXsMaxSteps; ++s) {
        if (s < steps) {                                           // (wave-uniform)
            if (s + 1 < steps) {
#pragma unroll
                for (int nb = 0; nb < 3; ++nb)
                    bq[(s + 1) & 1][nb] = live[nb] ? *reinterpret_cast<const uint4 *>(wrow[nb] + (s + 1) * 16) : make_uint4(0, 0, 0, 0);
            }
            const bf16x8 a = __builtin_bit_cast(bf16x8, aq[s]);
#pragma unroll
            for (int nb = 0; nb < 3; ++nb)
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, __builtin_bit_cast(bf16x8, bq[s & 1][nb]), acc[nb], 0, 0, 0);
        }
    }
#pragma unroll
    for (int nb = 0; nb < 3; ++nb)
#pragma unroll
        for (int r = 0; r < 16; ++r) s_part[wave][nb][r][lane] = acc[nb][r];
    __syncthreads();
    // C/D layout: column (feature) = lane & 31, row (token) = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int q = 0; q < 3 * 16 * 64 / (64 * kXsWaves); ++q) {
        const int idx = q * 64 * kXsWaves + tid, l = idx & 63, r = (idx >> 6) & 15, nb = idx >> 10;
        float v = 0.f;
#pragma unroll
        for (int w = 0; w < kXsWaves; ++w) v += s_part[w][nb][r][l];
        const int n = nb * 32 + (l & 31);
        const int64_t m = m0 + (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        if (n < p.n && m < p.m) ow[m * p.out_row_stride + n] = from_float<BF16>(v);
    }
}

}  // namespace zigma

using namespace zigma;

extern "C" int zigma_x_proj_fwd(const zigma_xproj_params_t *pp, void *stream_) {
    if (!pp) return ZIGMA_ERR_NULL;
    (void)hipGetLastError();
    const zigma_xproj_params_t &p = *pp;
    if (p.m < 0 || p.n < 1 || p.k < 1) return ZIGMA_ERR_SHAPE;
    if (p.flags != 0) return ZIGMA_ERR_UNSUPPORTED;
    if (p.m == 0) return ZIGMA_OK;
    if (!p.x || !p.w || !p.out) return ZIGMA_ERR_NULL;
    if (p.dtype != ZIGMA_BF16) return ZIGMA_ERR_DTYPE;
    if (p.n > kXpRows || p.k % kXpChunk != 0 || p.k / 16 < kXpDepth) return ZIGMA_ERR_SHAPE;
    if (p.x_row_stride % 8 != 0 || p.w_row_stride % 8 != 0 || reinterpret_cast<uintptr_t>(p.x) % 16 != 0 ||
        reinterpret_cast<uintptr_t>(p.w) % 16 != 0)
        return ZIGMA_ERR_STRIDE;
    // few tokens: K split over the waves of 32-token workgroups (the streaming form would leave most CUs idle)
    if (p.m < 16384 && p.k % (16 * kXsWaves) == 0 && p.k / (16 * kXsWaves) <= kXsMaxSteps) {
        hipLaunchKernelGGL(x_proj_splitk_kernel, dim3(static_cast<unsigned>((p.m + kXpTok - 1) / kXpTok)), dim3(64 * kXsWaves), 0,
                           static_cast<hipStream_t>(stream_), p);
        set_last_kernel("x_proj_splitk");
        return check_launch();
    }
    const int64_t tok_per_wg = kXpTok * kXpWaves;
    hipLaunchKernelGGL(x_proj_kernel, dim3(static_cast<unsigned>((p.m + tok_per_wg - 1) / tok_per_wg)), dim3(64 * kXpWaves), 0,
                       static_cast<hipStream_t>(stream_), p);
    set_last_kernel("x_proj_mfma");
    return check_launch();
}
