// Cross-attention core (softmax(Q K^T / sqrt(d)) V over a SHORT context) on the matrix cores, gfx950.
// C ABI: zigma_cross_attn_fwd.
//
// Replaces the scaled_dot_product_attention call of CrossAttention.forward (reference model_zigma.py:113-127: 8 heads x 64,
// 77 text tokens, no mask, no dropout at inference).  With n_ctx <= 128 the whole K_h / V_h of a (sample, head) is
// 2 x 10 KB: there is no "flash" loop, no online softmax — one pass:
//   workgroup = (256 query tokens, head, sample): K_h (row-major) and V_h^T staged ONCE in LDS, then 4 waves x 4 tiles of 16
//   S^T = K Q^T : v_mfma_f32_16x16x32_bf16, A = K rows from LDS, B = Q rows straight from HBM (16 B per lane, next tile
//                 prefetched).  The TRANSPOSED product puts a token in a lane column and 4 consecutive keys in its 4
//                 accumulator registers: softmax needs 2 cross-lane steps, P is written as packed 8-byte pieces
//   softmax     : exp2 with the scale folded into the exponent, padded keys masked to -inf
//   O^T = V^T P^T : A = V^T rows from LDS, B = P rows (bf16) from the per-wave LDS tile
//   O^T / rowsum -> per-wave LDS tile (8-byte pieces) -> 16-byte stores, 128 contiguous bytes per token
// HBM-bound by construction: reads Q once, writes O once (2 x 67 MB at B=64, L=1024); K/V come from L2.
// bf16 only (MFMA operand type), head_dim 64, n_ctx <= 128.
#include "zigma_common.h"

namespace zigma {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kXaD = 64;                 // head dim
constexpr int kXaWaves = 4, kXaTok = 16; // tokens per wave (MFMA M)
constexpr int kXaKPitch = (kXaD + 8) * 2;   // bytes per K row in LDS (16 B skew)

// 16-token tiles per wave = how many query tokens share one staging of K_h / V_h^T (20 KB): 8 (512 tokens per workgroup) where the
// sequence has them — 35.5 us against 38.0 with 4 and 40.2 with 16 at the headline shape (tools/attn_probe.py) —, 4 for short ones

typedef float xa_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 xa_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ uint32_t pack_bf16(float lo, float hi) {        // one v_cvt_pk_bf16_f32 (round to nearest even, like from_float<BF16>)
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(xa_f32x2{lo, hi}, xa_bf16x2));
}

// NKB 16-key blocks: n_ctx <= 16 * NKB; MASK_ALL = false: n_ctx > 16 (NKB - 1), padded keys only in the last block
template <int NKB, bool MASK_ALL>
__global__ __launch_bounds__(64 * kXaWaves) void cross_attn_kernel(const zigma_xattn_params_t p, const int kXaTiles) {
    constexpr int KP = 16 * NKB;                         // keys covered by S
    constexpr int KS = (KP + 31) / 32, KP2 = 32 * KS;    // k-steps / padded keys of the P V product
    constexpr int VPitch = (KP2 + 8) * 2;                // bytes per row of V^T and of P (16 B skew)
    constexpr int OPitch = kXaKPitch;
    static_assert(VPitch >= OPitch, "the O tile reuses the P tile");
    __shared__ __attribute__((aligned(16))) unsigned char s_k[KP * kXaKPitch];
    __shared__ __attribute__((aligned(16))) unsigned char s_vt[kXaD * VPitch];
    __shared__ __attribute__((aligned(16))) unsigned char s_p[kXaWaves][kXaTok * VPitch];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = blockIdx.y, b = blockIdx.z, L = p.seqlen, NC = p.n_ctx;
    const int i16 = lane & 15, g = lane >> 4;
    const uint16_t *qb = reinterpret_cast<const uint16_t *>(p.q) + b * p.q_batch_stride + h * kXaD;
    const uint16_t *kb = reinterpret_cast<const uint16_t *>(p.k) + b * p.k_batch_stride + h * kXaD;
    const uint16_t *vb = reinterpret_cast<const uint16_t *>(p.v) + b * p.v_batch_stride + h * kXaD;
    uint16_t *ob = reinterpret_cast<uint16_t *>(p.out) + b * p.o_batch_stride + h * kXaD;

    // ---- stage K_h (row-major, 16-byte pieces) and V_h^T (8 keys x 2 dims per unit -> two 16-byte rows); zero padding ----
    for (int piece = tid; piece < KP * 8; piece += 64 * kXaWaves) {
        const int row = piece >> 3, pc = piece & 7;
        uint4 kv = make_uint4(0, 0, 0, 0);
        if (row < NC) kv = *reinterpret_cast<const uint4 *>(kb + row * p.k_row_stride + pc * 8);
        *reinterpret_cast<uint4 *>(s_k + row * kXaKPitch + pc * 16) = kv;
    }
    for (int unit = tid; unit < (KP2 / 8) * (kXaD / 2); unit += 64 * kXaWaves) {
        const int kg = unit / (kXaD / 2), dp = unit % (kXaD / 2);
        uint32_t w[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int key = kg * 8 + e;
            w[e] = key < NC ? *reinterpret_cast<const uint32_t *>(vb + key * p.v_row_stride + dp * 2) : 0u;
        }
        uint4 lo, hi;       // dims 2dp / 2dp+1 of keys 8kg .. 8kg+7
        lo.x = (w[0] & 0xffffu) | (w[1] << 16); lo.y = (w[2] & 0xffffu) | (w[3] << 16);
        lo.z = (w[4] & 0xffffu) | (w[5] << 16); lo.w = (w[6] & 0xffffu) | (w[7] << 16);
        hi.x = (w[0] >> 16) | (w[1] & 0xffff0000u); hi.y = (w[2] >> 16) | (w[3] & 0xffff0000u);
        hi.z = (w[4] >> 16) | (w[5] & 0xffff0000u); hi.w = (w[6] >> 16) | (w[7] & 0xffff0000u);
        *reinterpret_cast<uint4 *>(s_vt + (2 * dp) * VPitch + kg * 16) = lo;
        *reinterpret_cast<uint4 *>(s_vt + (2 * dp + 1) * VPitch + kg * 16) = hi;
    }
    __syncthreads();

    const float sc = p.scale * kLog2e;
    unsigned char *pt = s_p[wave];
    const int tile0 = blockIdx.x * kXaTiles * kXaWaves + wave;      // this wave's tiles: tile0, tile0 + 4, ...
    auto load_q = [&](int t0, bf16x8 (&qa)[2]) {                    // lane -> token t0 + i16, dims 32 ks + 8 g ..
        int tq = t0 + i16;
        tq = tq < L ? tq : L - 1;
        const uint16_t *qrow = qb + static_cast<int64_t>(tq) * p.q_row_stride;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qa[ks] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(qrow + ks * 32 + g * 8));
    };
    bf16x8 qn[2];
    if (tile0 * kXaTok < L) load_q(tile0 * kXaTok, qn);
#pragma unroll 1
    for (int it = 0; it < kXaTiles; ++it) {
        const int t0 = (tile0 + it * kXaWaves) * kXaTok;
        if (t0 >= L) break;                                         // wave-uniform; no workgroup barriers below
        const bf16x8 qa[2] = {qn[0], qn[1]};
        if (it + 1 < kXaTiles && t0 + kXaWaves * kXaTok < L) load_q(t0 + kXaWaves * kXaTok, qn);
        // ---- S^T = K Q^T : lane -> token column i16, key rows 16 nb + 4 g + r -----------------------------------------
        f32x4 s[NKB];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {                 // (k-step outermost: consecutive MFMAs on different accumulators; the first on a literal zero)
#pragma unroll
            for (int nb = 0; nb < NKB; ++nb) {
                const bf16x8 kf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(s_k + (nb * 16 + i16) * kXaKPitch + ks * 64 + g * 16));
                s[nb] = ks == 0 ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qa[ks], f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0)
                                : __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qa[ks], s[nb], 0, 0, 0);
            }
        }
        // ---- softmax over the keys of this lane's token: 4 NKB values in the lane, the rest in lanes i16 + 16 g' ----------
        // (raw scores: the positive scale commutes with the maximum and enters the exponent as one fma per key)
        float m = -INFINITY;
#pragma unroll
        for (int nb = 0; nb < NKB; ++nb) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (MASK_ALL || nb == NKB - 1) s[nb][r] = (nb * 16 + 4 * g + r < NC) ? s[nb][r] : -INFINITY;     // padded keys
                m = fmaxf(m, s[nb][r]);
            }
        }
        m = fmaxf(m, __shfl_xor(m, 16, 64));
        m = fmaxf(m, __shfl_xor(m, 32, 64));
        const float msc = -m * sc;
        float sum = 0.f;
#pragma unroll
        for (int nb = 0; nb < NKB; ++nb) {
            float e[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) { e[r] = fast_exp2(__builtin_fmaf(s[nb][r], sc, msc)); sum += e[r]; }   // exp2(-inf) = 0 for padded keys
            *reinterpret_cast<uint2 *>(pt + i16 * VPitch + (nb * 16 + 4 * g) * 2) = make_uint2(pack_bf16(e[0], e[1]), pack_bf16(e[2], e[3]));
        }
        if (KP2 > KP) *reinterpret_cast<uint2 *>(pt + i16 * VPitch + (KP + 4 * g) * 2) = make_uint2(0u, 0u);
        sum += __shfl_xor(sum, 16, 64);
        sum += __shfl_xor(sum, 32, 64);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- O^T = V^T P^T : lane -> token column i16, dims 16 db + 4 g + r ---------------------------------------------------
        f32x4 o[4];
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
            const bf16x8 pf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(pt + i16 * VPitch + ks * 64 + g * 16));
#pragma unroll
            for (int db = 0; db < 4; ++db) {
                const bf16x8 vf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(s_vt + (db * 16 + i16) * VPitch + ks * 64 + g * 16));
                o[db] = ks == 0 ? __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, f32x4{0.f, 0.f, 0.f, 0.f}, 0, 0, 0)
                                : __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pf, o[db], 0, 0, 0);
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                  // P tile fully consumed: reuse it for O
        const float inv = fast_rcp(sum);
#pragma unroll
        for (int db = 0; db < 4; ++db)
            *reinterpret_cast<uint2 *>(pt + i16 * OPitch + (db * 16 + 4 * g) * 2) =
                make_uint2(pack_bf16(o[db][0] * inv, o[db][1] * inv), pack_bf16(o[db][2] * inv, o[db][3] * inv));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int idx = lane + 64 * q, tok = idx >> 3, pc = idx & 7;
            if (t0 + tok < L)
                *reinterpret_cast<uint4 *>(ob + static_cast<int64_t>(t0 + tok) * p.o_row_stride + pc * 8) =
                    *reinterpret_cast<const uint4 *>(pt + tok * OPitch + pc * 16);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                  // the next tile overwrites the P / O tile
    }
}

}  // namespace zigma

using namespace zigma;

extern "C" int zigma_cross_attn_fwd(const zigma_xattn_params_t *pp, void *stream_) {
    if (!pp) return ZIGMA_ERR_NULL;
    (void)hipGetLastError();
    const zigma_xattn_params_t &p = *pp;
    if (p.batch < 0 || p.seqlen < 0 || p.heads < 1 || p.n_ctx < 1) return ZIGMA_ERR_SHAPE;
    if (p.flags != 0) return ZIGMA_ERR_UNSUPPORTED;
    // the softmax takes its maximum over the RAW scores and folds the scale into the exponent (fma(s, sc, -m * sc)): that is the
    // maximum of the scaled scores only for a positive scale (the reference's is head_dim^-0.5)
    if (!(p.scale > 0.f) || !(p.scale < 3.0e38f)) return ZIGMA_ERR_UNSUPPORTED;
    if (p.batch == 0 || p.seqlen == 0) return ZIGMA_OK;
    if (!p.q || !p.k || !p.v || !p.out) return ZIGMA_ERR_NULL;
    if (p.dtype != ZIGMA_BF16) return ZIGMA_ERR_DTYPE;
    if (p.head_dim != kXaD || p.n_ctx > 128 || p.batch > 65535 || p.heads > 65535) return ZIGMA_ERR_SHAPE;
    auto mis = [](const void *q, int64_t rs, int64_t bs) { return reinterpret_cast<uintptr_t>(q) % 16 != 0 || rs % 8 != 0 || bs % 8 != 0; };
    if (mis(p.q, p.q_row_stride, p.q_batch_stride) || mis(p.k, p.k_row_stride, p.k_batch_stride) ||
        mis(p.v, p.v_row_stride, p.v_batch_stride) || mis(p.out, p.o_row_stride, p.o_batch_stride))
        return ZIGMA_ERR_STRIDE;
    const int tiles = p.seqlen >= 512 ? 8 : 4;
    const int tok_per_wg = kXaTok * kXaWaves * tiles;
    dim3 grid((p.seqlen + tok_per_wg - 1) / tok_per_wg, p.heads, p.batch), block(64 * kXaWaves);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (p.n_ctx <= 80) {
        if (p.n_ctx > 64) hipLaunchKernelGGL((cross_attn_kernel<5, false>), grid, block, 0, stream, p, tiles);
        else hipLaunchKernelGGL((cross_attn_kernel<5, true>), grid, block, 0, stream, p, tiles);
    } else {
        if (p.n_ctx > 112) hipLaunchKernelGGL((cross_attn_kernel<8, false>), grid, block, 0, stream, p, tiles);
        else hipLaunchKernelGGL((cross_attn_kernel<8, true>), grid, block, 0, stream, p, tiles);
    }
    set_last_kernel("cross_attn_mfma");
    return check_launch();
}
