// Skinny projections: out = act(x) @ W^T + bias for at most 64 rows of x (one per sample), on the matrix cores, gfx950.
// C ABI: zigma_skinny_linear_fwd.
//
// The conditioning path of the ZigMa forward: the timestep MLP (model_zigma.py:232-275: Linear, SiLU, Linear on (B, 256) / (B, E)) and the
// adaLN modulation of every block, SiLU + Linear(E, 6E) on c (B, E) (:441, batched over the 18 blocks: n = 69 120).  With m <= 64 these are
// weight-streaming problems (88 MB of weights against 80 KB of activations at the headline shape): a tile GEMM spends its time filling
// tiles, the library takes 35 us for the adaLN product.  Here
//   workgroup = 8 waves; act(x) (SiLU evaluated in fp32, rounded to bf16 like the reference's bf16 module) sits ONCE in LDS, rows padded to 64;
//   wave -> 16 rows of W (16 output features), streamed straight from HBM as MFMA A fragments (v_mfma_f32_16x16x32_bf16: lane = weight row,
//   8 consecutive k), the whole strip (k / 32 loads of 16 bytes per lane) in flight per wave; B fragments = x rows from LDS (lane = sample, 8 consecutive k); 4 accumulator blocks
//   (64 samples); bias added from a 8-byte read per lane; out[m][n0 + 4 g + r] leaves as 8-byte stores.
// HBM-bound on the weight read.
#include "zigma_common.h"

namespace zigma {

typedef __bf16 sk_bf16x8 __attribute__((ext_vector_type(8)));
typedef float sk_f32x4 __attribute__((ext_vector_type(4)));

constexpr int kSkWaves = 8;

// KS: k / 32 as a compile-time constant — the whole strip of weights (KS 16-byte fragments per lane) is requested up front, no branch sits
// between the loads and the products (with a run-time k the refill conditions made hipcc wait for ALL loads at every step: 0.5 us per k-step)
template <bool SILU, int KS>
__global__ __launch_bounds__(64 * kSkWaves) void skinny_linear_kernel(const zigma_skinny_params_t p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_x[];      // [64][K + 8] bf16
    constexpr int K = KS * 32, pitch = (K + 8) * 2;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // ---- stage act(x): 16-byte pieces, rows >= m are zero ----------------------------------------------------------------------
    constexpr int PPR = KS * 4, NB = (64 * PPR) / (64 * kSkWaves);      // 16-byte pieces per row; pieces per thread: ALL requested before the first use
    {
        uint4 v[NB];
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int piece = q * 64 * kSkWaves + tid, row = piece / PPR, pc = piece % PPR;
            v[q] = make_uint4(0, 0, 0, 0);
            if (row < p.m) v[q] = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint16_t *>(p.x) + static_cast<int64_t>(row) * p.x_row_stride + pc * 8);
        }
#pragma unroll
        for (int q = 0; q < NB; ++q) {
            const int piece = q * 64 * kSkWaves + tid, row = piece / PPR, pc = piece % PPR;
            if (SILU) {
                uint32_t w[4] = {v[q].x, v[q].y, v[q].z, v[q].w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float lo = silu(to_float<BF16>(static_cast<uint16_t>(w[i]))), hi = silu(to_float<BF16>(static_cast<uint16_t>(w[i] >> 16)));
                    w[i] = static_cast<uint32_t>(from_float<BF16>(lo)) | (static_cast<uint32_t>(from_float<BF16>(hi)) << 16);
                }
                v[q] = make_uint4(w[0], w[1], w[2], w[3]);
            }
            *reinterpret_cast<uint4 *>(s_x + row * pitch + pc * 16) = v[q];
        }
    }
    __syncthreads();
    const int i16 = lane & 15, g = lane >> 4;
    const int n_strips = p.n / 16;
    const int stride = gridDim.x * kSkWaves;
    auto wrow_of = [&](int strip) { return reinterpret_cast<const uint16_t *>(p.w) + static_cast<int64_t>(strip * 16 + i16) * p.w_row_stride + g * 8; };
    auto fetch = [&](int strip, uint4 (&wq)[KS]) {
        const uint16_t *wrow = wrow_of(strip);
#pragma unroll
        for (int i = 0; i < KS; ++i) wq[i] = *reinterpret_cast<const uint4 *>(wrow + i * 32);
    };
    auto compute = [&](int strip, const uint4 (&wq)[KS]) {
        const int n0 = strip * 16;
        sk_f32x4 acc[4];
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) acc[mb] = sk_f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            // (the x fragments do not depend on the strip: without this fence hipcc hoists all 4 KS of them out of the strip loop — 320
            // registers for k = 640 — and spills)
            asm volatile("" ::: "memory");
            const sk_bf16x8 wf = __builtin_bit_cast(sk_bf16x8, wq[ks]);
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) {
                const sk_bf16x8 xf = __builtin_bit_cast(sk_bf16x8, *reinterpret_cast<const uint4 *>(s_x + (mb * 16 + i16) * pitch + ks * 64 + g * 16));
                acc[mb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wf, xf, acc[mb], 0, 0, 0);      // D[i = feature 4 g + r][j = sample i16]
            }
        }
        float bv[4] = {0.f, 0.f, 0.f, 0.f};
        if (p.bias) {
            const uint2 b2 = *reinterpret_cast<const uint2 *>(reinterpret_cast<const uint16_t *>(p.bias) + n0 + 4 * g);
            bv[0] = to_float<BF16>(static_cast<uint16_t>(b2.x)); bv[1] = to_float<BF16>(static_cast<uint16_t>(b2.x >> 16));
            bv[2] = to_float<BF16>(static_cast<uint16_t>(b2.y)); bv[3] = to_float<BF16>(static_cast<uint16_t>(b2.y >> 16));
        }
#pragma unroll
        for (int mb = 0; mb < 4; ++mb) {
            const int m = mb * 16 + i16;
            if (m < p.m) {
                uint2 o;
                o.x = static_cast<uint32_t>(from_float<BF16>(acc[mb][0] + bv[0])) | (static_cast<uint32_t>(from_float<BF16>(acc[mb][1] + bv[1])) << 16);
                o.y = static_cast<uint32_t>(from_float<BF16>(acc[mb][2] + bv[2])) | (static_cast<uint32_t>(from_float<BF16>(acc[mb][3] + bv[3])) << 16);
                *reinterpret_cast<uint2 *>(reinterpret_cast<uint16_t *>(p.out) + static_cast<int64_t>(m) * p.out_row_stride + n0 + 4 * g) = o;
            }
        }
    };
    int strip = blockIdx.x * kSkWaves + wave;
    if constexpr (KS <= 20) {        // two strips of weights in registers: the next strip is on its way while this one is multiplied
        uint4 wa[KS], wb[KS];
        if (strip < n_strips) fetch(strip, wa);
        while (strip < n_strips) {
            if (strip + stride < n_strips) fetch(strip + stride, wb);
            compute(strip, wa);
            strip += stride;
            if (strip >= n_strips) break;
            if (strip + stride < n_strips) fetch(strip + stride, wa);
            compute(strip, wb);
            strip += stride;
        }
    } else {
        uint4 wa[KS];
        for (; strip < n_strips; strip += stride) {
            fetch(strip, wa);
            compute(strip, wa);
        }
    }
}

}  // namespace zigma

using namespace zigma;

extern "C" int zigma_skinny_linear_fwd(const zigma_skinny_params_t *pp, void *stream_) {
    if (!pp) return ZIGMA_ERR_NULL;
    (void)hipGetLastError();
    const zigma_skinny_params_t &p = *pp;
    if (p.m < 0 || p.n < 0 || p.k < 0) return ZIGMA_ERR_SHAPE;
    if (p.flags & ~1) return ZIGMA_ERR_UNSUPPORTED;
    if (p.dtype != ZIGMA_BF16) return ZIGMA_ERR_DTYPE;
    if (p.m > 64 || p.n % 16 != 0 || p.k % 128 != 0 || p.k > 1024) return ZIGMA_ERR_SHAPE;       // (64 x (k + 8) bf16 of LDS <= 129 KB; k / 128 instantiations)
    if (p.m == 0 || p.n == 0) return ZIGMA_OK;
    if (p.k == 0 || !p.x || !p.w || !p.out) return ZIGMA_ERR_NULL;
    auto mis = [](const void *q, int64_t rs, int al) { return reinterpret_cast<uintptr_t>(q) % al != 0 || rs % (al / 2) != 0; };
    if (mis(p.x, p.x_row_stride, 16) || mis(p.w, p.w_row_stride, 16) || mis(p.out, p.out_row_stride, 8)) return ZIGMA_ERR_STRIDE;
    if (p.bias && reinterpret_cast<uintptr_t>(p.bias) % 8 != 0) return ZIGMA_ERR_STRIDE;
    const int n_strips = p.n / 16;
    const int want = (n_strips + kSkWaves - 1) / kSkWaves;
    const dim3 grid(want < 256 ? want : 256), block(64 * kSkWaves);      // one workgroup per CU (LDS), persistent over the strips: x is staged once
    const size_t lds = static_cast<size_t>(64) * (p.k + 8) * 2;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    // (more than 64 KB of dynamic LDS needs a function attribute; it applies to the CURRENT device only, so it is set per call —
    // a host-side table update, no device work — instead of once per process: a second GPU driven by the same process would
    // otherwise launch without it and fail)
#define ZIGMA_SK_CASE(KS_)                                                                                                     \
    case KS_: {                                                                                                                \
        const int v = p.flags & 1;                                                                                             \
        const void *fn = v ? reinterpret_cast<const void *>(skinny_linear_kernel<true, KS_>)                                   \
                           : reinterpret_cast<const void *>(skinny_linear_kernel<false, KS_>);                                 \
        if (lds > 65536 &&                                                                                                     \
            hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds)) != hipSuccess)         \
            return ZIGMA_ERR_LAUNCH;                                                                                                                      \
        if (v) hipLaunchKernelGGL((skinny_linear_kernel<true, KS_>), grid, block, lds, stream, p);                             \
        else hipLaunchKernelGGL((skinny_linear_kernel<false, KS_>), grid, block, lds, stream, p);                              \
        break;                                                                                                                 \
    }
    switch (p.k / 32) {
        ZIGMA_SK_CASE(4) ZIGMA_SK_CASE(8) ZIGMA_SK_CASE(12) ZIGMA_SK_CASE(16) ZIGMA_SK_CASE(20) ZIGMA_SK_CASE(24) ZIGMA_SK_CASE(28) ZIGMA_SK_CASE(32)
        default: return ZIGMA_ERR_SHAPE;
    }
#undef ZIGMA_SK_CASE
    set_last_kernel("skinny_linear_mfma");
    return check_launch();
}
