// Backward of the cross-attention core (zigma_cross_attn_fwd) on the matrix cores, gfx950.   C ABI: zigma_cross_attn_bwd.
//
// What autograd derives for the scaled_dot_product_attention call of CrossAttention.forward (reference model_zigma.py:113-127) when
// the reference trains: with P = softmax(scale Q K^T), dP = dO V^T, delta = rowsum(P o dP), dS = scale P o (dP - delta):
//   dQ = dS K,   dK = dS^T Q,   dV = P^T dO.
// The context is short (n_ctx <= 128), so nothing of the forward is saved: a workgroup stages K_h, V_h (row-major) and K_h^T once and
// RECOMPUTES the probabilities of each 16-token tile — in BOTH orientations, because the two kinds of product want the tile in the two
// transposed register layouts and a second 16x80x64 MFMA product is cheaper than a transpose through LDS:
//   "token in the lane"  S^T = K Q^T, dP^T = V dO^T  (16x16x32, operands: K / V rows from LDS, Q / dO rows straight from HBM)
//        -> row max, 1/rowsum, delta (2 cross-lane steps each), dS^T; the accumulator layout of a 16x16 block IS the B-operand layout of
//           v_mfma_f32_16x16x16_bf16, so dQ^T = K^T dS^T consumes dS^T straight from the registers (A = K^T rows from LDS)
//   "key in the lane"    S = Q K^T, dP = dO V^T      (the SAME operand registers, swapped); the row statistics come from the first
//        orientation by 12 lane reads.  P and dS are then B operands of the products that contract over the TOKENS,
//        dV^T += dO^T P,  dK^T += Q^T dS,  whose A operands dO^T / Q^T (feature in the lane, 4 tokens per lane) are obtained from the
//        row-major operand registers by a product with the identity (X I = X, accumulator layout = A-operand layout of X^T; exact)
// dK^T / dV^T stay in 2 x 4 x NKB accumulator blocks per wave over all of the wave's tiles, are summed over the 4 waves in a fixed
// order through LDS and written as fp32 partials per (chunk of tokens, sample): the caller adds the chunks (deterministic, no atomics).
// HBM traffic: reads Q and dO once, writes dQ once (3 x 67 MB at B=64, L=1024, 8 heads).
// bf16 only, head_dim 64, n_ctx <= 128.
#include "zigma_common.h"

namespace zigma {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kXbD = 64, kXbWaves = 4, kXbTok = 16;
constexpr int kXbRowPitch = (kXbD + 8) * 2;      // bytes per K / V row and per dQ-tile row in LDS (16 B skew)
constexpr int kXbRedPitch = (kXbD + 4) * 4;      // bytes per key row of the fp32 dK / dV reduction tiles

__device__ __forceinline__ uint32_t xb_pack(float lo, float hi) {
    return static_cast<uint32_t>(from_float<BF16>(lo)) | (static_cast<uint32_t>(from_float<BF16>(hi)) << 16);
}
__device__ __forceinline__ s16x4 xb_pack4(const f32x4 v) {
    return __builtin_bit_cast(s16x4, make_uint2(xb_pack(v[0], v[1]), xb_pack(v[2], v[3])));
}
__device__ __forceinline__ f32x4 mfma16(s16x4 a, s16x4 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(a, b, c, 0, 0, 0); }

constexpr int xb_max(int a, int b) { return a > b ? a : b; }

template <int NKB, int WPE, bool PF>     // 16-key blocks: n_ctx <= 16 * NKB; waves per SIMD the register budget is set for; operand prefetch
__global__ __launch_bounds__(64 * kXbWaves) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void cross_attn_bwd_kernel(const zigma_xattn_bwd_params_t p, const int tiles) {
    constexpr int KP = 16 * NKB;
    constexpr int KTPitch = (KP + 8) * 2;                 // bytes per row of K^T
    constexpr int kOffV = KP * kXbRowPitch, kOffKT = 2 * KP * kXbRowPitch, kOffTile = kOffKT + kXbD * KTPitch;
    constexpr int kStage = kOffTile + kXbWaves * kXbTok * kXbRowPitch, kRed = 2 * KP * kXbRedPitch;
    __shared__ __attribute__((aligned(16))) unsigned char smem[xb_max(kStage, kRed)];
    unsigned char *s_k = smem, *s_v = smem + kOffV, *s_kt = smem + kOffKT;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int h = blockIdx.y, b = blockIdx.z, L = p.seqlen, NC = p.n_ctx;
    const int i16 = lane & 15, g = lane >> 4;
    const uint16_t *qb = reinterpret_cast<const uint16_t *>(p.q) + b * p.q_batch_stride + h * kXbD;
    const uint16_t *gb = reinterpret_cast<const uint16_t *>(p.dout) + b * p.do_batch_stride + h * kXbD;
    const uint16_t *kb = reinterpret_cast<const uint16_t *>(p.k) + b * p.k_batch_stride + h * kXbD;
    const uint16_t *vb = reinterpret_cast<const uint16_t *>(p.v) + b * p.v_batch_stride + h * kXbD;
    uint16_t *dqb = reinterpret_cast<uint16_t *>(p.dq) + b * p.dq_batch_stride + h * kXbD;

    // ---- stage K_h, V_h (row-major, zero rows past n_ctx) and K_h^T (8 keys x 2 features per unit -> two 16-byte row pieces) ----
    for (int piece = tid; piece < KP * 8; piece += 64 * kXbWaves) {
        const int row = piece >> 3, pc = piece & 7;
        uint4 kv = make_uint4(0, 0, 0, 0), vv = make_uint4(0, 0, 0, 0);
        if (row < NC) {
            kv = *reinterpret_cast<const uint4 *>(kb + row * p.k_row_stride + pc * 8);
            vv = *reinterpret_cast<const uint4 *>(vb + row * p.v_row_stride + pc * 8);
        }
        *reinterpret_cast<uint4 *>(s_k + row * kXbRowPitch + pc * 16) = kv;
        *reinterpret_cast<uint4 *>(s_v + row * kXbRowPitch + pc * 16) = vv;
    }
    for (int unit = tid; unit < (KP / 8) * (kXbD / 2); unit += 64 * kXbWaves) {
        const int kg = unit / (kXbD / 2), dp = unit % (kXbD / 2);
        uint32_t w[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int key = kg * 8 + e;
            w[e] = key < NC ? *reinterpret_cast<const uint32_t *>(kb + key * p.k_row_stride + dp * 2) : 0u;
        }
        uint4 lo, hi;
        lo.x = (w[0] & 0xffffu) | (w[1] << 16); lo.y = (w[2] & 0xffffu) | (w[3] << 16);
        lo.z = (w[4] & 0xffffu) | (w[5] << 16); lo.w = (w[6] & 0xffffu) | (w[7] << 16);
        hi.x = (w[0] >> 16) | (w[1] & 0xffff0000u); hi.y = (w[2] >> 16) | (w[3] & 0xffff0000u);
        hi.z = (w[4] >> 16) | (w[5] & 0xffff0000u); hi.w = (w[6] >> 16) | (w[7] & 0xffff0000u);
        *reinterpret_cast<uint4 *>(s_kt + (2 * dp) * KTPitch + kg * 16) = lo;
        *reinterpret_cast<uint4 *>(s_kt + (2 * dp + 1) * KTPitch + kg * 16) = hi;
    }
    __syncthreads();

    const float sc2 = p.scale * kLog2e, scale = p.scale;
    unsigned char *tile = smem + kOffTile + wave * (kXbTok * kXbRowPitch);
    // identity piece of the register transposes: lane (column i16, rows 4 g + e) holds 1.0 where 4 g + e == i16
    s16x4 ident;
#pragma unroll
    for (int e = 0; e < 4; ++e) ident[e] = (4 * g + e == i16) ? static_cast<short>(0x3f80) : static_cast<short>(0);

    f32x4 dvt[4][NKB], dkt[4][NKB];      // [feature block (ks, half)][key block]: lane -> key 16 nb + i16, features 32 ks + 8 g + 4 half + r
#pragma unroll
    for (int blk = 0; blk < 4; ++blk)
#pragma unroll
        for (int nb = 0; nb < NKB; ++nb) dvt[blk][nb] = dkt[blk][nb] = f32x4{0.f, 0.f, 0.f, 0.f};

    const int tile0 = blockIdx.x * tiles * kXbWaves + wave;          // this wave's tiles: tile0, tile0 + 4, ...
    auto load_rows = [&](const uint16_t *base, int64_t row_stride, int t0, uint4 (&xa)[2]) {
        const int tq = t0 + i16;                                     // lane -> token t0 + i16, features 32 ks + 8 g ..
        const uint16_t *row = base + static_cast<int64_t>(tq < L ? tq : L - 1) * row_stride;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) xa[ks] = *reinterpret_cast<const uint4 *>(row + ks * 32 + g * 8);
    };
    uint4 qn[2], dn[2];
    if (PF && tile0 * kXbTok < L) { load_rows(qb, p.q_row_stride, tile0 * kXbTok, qn); load_rows(gb, p.do_row_stride, tile0 * kXbTok, dn); }
#pragma unroll 1
    for (int it = 0; it < tiles; ++it) {
        const int t0 = (tile0 + it * kXbWaves) * kXbTok;
        if (t0 >= L) break;                                          // wave-uniform; no workgroup barriers inside the loop
        if (!PF) { load_rows(qb, p.q_row_stride, t0, qn); load_rows(gb, p.do_row_stride, t0, dn); }
        const bool tail = t0 + i16 >= L;                             // dO = 0 for tokens past the end: no contribution to dK / dV
        const bf16x8 qa[2] = {__builtin_bit_cast(bf16x8, qn[0]), __builtin_bit_cast(bf16x8, qn[1])};
        const bf16x8 da[2] = {__builtin_bit_cast(bf16x8, tail ? make_uint4(0, 0, 0, 0) : dn[0]),
                              __builtin_bit_cast(bf16x8, tail ? make_uint4(0, 0, 0, 0) : dn[1])};
        if (PF && it + 1 < tiles && t0 + kXbWaves * kXbTok < L) {
            load_rows(qb, p.q_row_stride, t0 + kXbWaves * kXbTok, qn);
            load_rows(gb, p.do_row_stride, t0 + kXbWaves * kXbTok, dn);
        }
        // ======== token in the lane: S^T, dP^T -> statistics, dS^T, dQ^T ========
        float m = -INFINITY, inv, delta = 0.f;
        f32x4 dq[4];
        {
            f32x4 s[NKB], dpt[NKB];
#pragma unroll
            for (int nb = 0; nb < NKB; ++nb) {
                s[nb] = dpt[nb] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
                    const bf16x8 kf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(s_k + (nb * 16 + i16) * kXbRowPitch + ks * 64 + g * 16));
                    const bf16x8 vf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(s_v + (nb * 16 + i16) * kXbRowPitch + ks * 64 + g * 16));
                    s[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qa[ks], s[nb], 0, 0, 0);
                    dpt[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, da[ks], dpt[nb], 0, 0, 0);
                }
            }
#pragma unroll
            for (int nb = 0; nb < NKB; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float v = (nb * 16 + 4 * g + r < NC) ? s[nb][r] * sc2 : -INFINITY;
                    s[nb][r] = v;
                    m = fmaxf(m, v);
                }
            m = fmaxf(m, __shfl_xor(m, 16, 64));
            m = fmaxf(m, __shfl_xor(m, 32, 64));
            float sum = 0.f;
#pragma unroll
            for (int nb = 0; nb < NKB; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) { s[nb][r] = fast_exp2(s[nb][r] - m); sum += s[nb][r]; }   // exp2(-inf) = 0 for padded keys
            sum += __shfl_xor(sum, 16, 64);
            sum += __shfl_xor(sum, 32, 64);
            inv = fast_rcp(sum);
#pragma unroll
            for (int nb = 0; nb < NKB; ++nb)
#pragma unroll
                for (int r = 0; r < 4; ++r) { s[nb][r] *= inv; delta += s[nb][r] * dpt[nb][r]; }
            delta += __shfl_xor(delta, 16, 64);
            delta += __shfl_xor(delta, 32, 64);
            // dQ^T = K^T dS^T : lane -> token column i16, features 16 db + 4 g + r
#pragma unroll
            for (int db = 0; db < 4; ++db) dq[db] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int nb = 0; nb < NKB; ++nb) {
                f32x4 ds;
#pragma unroll
                for (int r = 0; r < 4; ++r) ds[r] = s[nb][r] * (dpt[nb][r] - delta);      // dS / scale
                const s16x4 dsb = xb_pack4(ds);
#pragma unroll
                for (int db = 0; db < 4; ++db) {
                    const s16x4 ktf = __builtin_bit_cast(s16x4, *reinterpret_cast<const uint2 *>(s_kt + (db * 16 + i16) * KTPitch + (nb * 16 + 4 * g) * 2));
                    dq[db] = mfma16(ktf, dsb, dq[db]);
                }
            }
        }
#pragma unroll
        for (int db = 0; db < 4; ++db)
            *reinterpret_cast<uint2 *>(tile + i16 * kXbRowPitch + (db * 16 + 4 * g) * 2) =
                make_uint2(xb_pack(dq[db][0] * scale, dq[db][1] * scale), xb_pack(dq[db][2] * scale, dq[db][3] * scale));
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int idx = lane + 64 * q, tok = idx >> 3, pc = idx & 7;
            if (t0 + tok < L)
                *reinterpret_cast<uint4 *>(dqb + static_cast<int64_t>(t0 + tok) * p.dq_row_stride + pc * 8) =
                    *reinterpret_cast<const uint4 *>(tile + tok * kXbRowPitch + pc * 16);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();                  // the next tile overwrites the dQ tile

        // ======== key in the lane: S, dP -> P, dS as B operands of the products over the tokens ========
        float mB[4], invB[4], deltaB[4];                  // statistics of tokens 4 g + r (held by lanes with i16 = 4 g + r)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            mB[r] = __shfl(m, 4 * g + r, 64);
            invB[r] = __shfl(inv, 4 * g + r, 64);
            deltaB[r] = __shfl(delta, 4 * g + r, 64);
        }
        // Q^T, dO^T pieces: block (ks, half): lane -> feature 32 ks + 4 half + 8 (i16 >> 2) + (i16 & 3), tokens 4 g + r
        s16x4 qT[4], dT[4];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
                const uint4 qw = __builtin_bit_cast(uint4, qa[ks]), dw = __builtin_bit_cast(uint4, da[ks]);
                const s16x4 qp = __builtin_bit_cast(s16x4, hf ? make_uint2(qw.z, qw.w) : make_uint2(qw.x, qw.y));
                const s16x4 dp = __builtin_bit_cast(s16x4, hf ? make_uint2(dw.z, dw.w) : make_uint2(dw.x, dw.y));
                qT[ks * 2 + hf] = xb_pack4(mfma16(qp, ident, f32x4{0.f, 0.f, 0.f, 0.f}));
                dT[ks * 2 + hf] = xb_pack4(mfma16(dp, ident, f32x4{0.f, 0.f, 0.f, 0.f}));
            }
#pragma unroll
        for (int nb = 0; nb < NKB; ++nb) {
            f32x4 s = f32x4{0.f, 0.f, 0.f, 0.f}, dp = s;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                const bf16x8 kf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(s_k + (nb * 16 + i16) * kXbRowPitch + ks * 64 + g * 16));
                const bf16x8 vf = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(s_v + (nb * 16 + i16) * kXbRowPitch + ks * 64 + g * 16));
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(qa[ks], kf, s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(da[ks], vf, dp, 0, 0, 0);
            }
            const bool live = nb * 16 + i16 < NC;
            f32x4 pr, ds;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                pr[r] = live ? fast_exp2(s[r] * sc2 - mB[r]) * invB[r] : 0.f;
                ds[r] = pr[r] * (dp[r] - deltaB[r]);                  // dS / scale: dK^T is scaled once, in the reduction
            }
            const s16x4 pb = xb_pack4(pr), dsb = xb_pack4(ds);
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) {
                dvt[blk][nb] = mfma16(dT[blk], pb, dvt[blk][nb]);
                dkt[blk][nb] = mfma16(qT[blk], dsb, dkt[blk][nb]);
            }
        }
    }

    // ---- sum dK^T / dV^T over the 4 waves in a fixed order (fp32 tiles [key][feature] over the staging area), write the partials ----
    __syncthreads();
    float *red_k = reinterpret_cast<float *>(smem), *red_v = reinterpret_cast<float *>(smem + KP * kXbRedPitch);
#pragma unroll 1
    for (int w = 0; w < kXbWaves; ++w) {
        if (wave == w) {
#pragma unroll
            for (int blk = 0; blk < 4; ++blk)
#pragma unroll
                for (int nb = 0; nb < NKB; ++nb) {
                    const int off = (nb * 16 + i16) * (kXbRedPitch / 4) + 32 * (blk >> 1) + 8 * g + 4 * (blk & 1);
                    f32x4 ak = dkt[blk][nb] * scale, av = dvt[blk][nb];
                    if (w > 0) {
                        ak += *reinterpret_cast<const f32x4 *>(red_k + off);
                        av += *reinterpret_cast<const f32x4 *>(red_v + off);
                    }
                    *reinterpret_cast<f32x4 *>(red_k + off) = ak;
                    *reinterpret_cast<f32x4 *>(red_v + off) = av;
                }
        }
        __syncthreads();
    }
    const int64_t C = static_cast<int64_t>(p.heads) * kXbD;
    const int64_t part = (static_cast<int64_t>(blockIdx.x) * p.batch + b) * NC * C + h * kXbD;
    float *dkp = reinterpret_cast<float *>(p.dk_part) + part, *dvp = reinterpret_cast<float *>(p.dv_part) + part;
    for (int piece = tid; piece < NC * 16; piece += 64 * kXbWaves) {
        const int key = piece >> 4, pc = piece & 15;
        *reinterpret_cast<f32x4 *>(dkp + key * C + pc * 4) = *reinterpret_cast<const f32x4 *>(red_k + key * (kXbRedPitch / 4) + pc * 4);
        *reinterpret_cast<f32x4 *>(dvp + key * C + pc * 4) = *reinterpret_cast<const f32x4 *>(red_v + key * (kXbRedPitch / 4) + pc * 4);
    }
}

static int xb_tiles(int seqlen) { return seqlen >= 512 ? 8 : 4; }

}  // namespace zigma

using namespace zigma;

extern "C" int zigma_cross_attn_bwd_chunks(int seqlen) {
    if (seqlen <= 0) return 0;
    const int tok_per_wg = kXbTok * kXbWaves * xb_tiles(seqlen);
    return (seqlen + tok_per_wg - 1) / tok_per_wg;
}

extern "C" int zigma_cross_attn_bwd(const zigma_xattn_bwd_params_t *pp, void *stream_) {
    if (!pp) return ZIGMA_ERR_NULL;
    (void)hipGetLastError();
    const zigma_xattn_bwd_params_t &p = *pp;
    if (p.batch < 0 || p.seqlen < 0 || p.heads < 1 || p.n_ctx < 1) return ZIGMA_ERR_SHAPE;
    if (p.flags != 0) return ZIGMA_ERR_UNSUPPORTED;
    if (p.batch == 0 || p.seqlen == 0) return ZIGMA_OK;
    if (!p.q || !p.k || !p.v || !p.dout || !p.dq || !p.dk_part || !p.dv_part) return ZIGMA_ERR_NULL;
    if (p.dtype != ZIGMA_BF16) return ZIGMA_ERR_DTYPE;
    if (p.head_dim != kXbD || p.n_ctx > 128 || p.batch > 65535 || p.heads > 65535) return ZIGMA_ERR_SHAPE;
    if (p.chunks != zigma_cross_attn_bwd_chunks(p.seqlen)) return ZIGMA_ERR_SHAPE;
    auto mis = [](const void *q, int64_t rs, int64_t bs) { return reinterpret_cast<uintptr_t>(q) % 16 != 0 || rs % 8 != 0 || bs % 8 != 0; };
    if (mis(p.q, p.q_row_stride, p.q_batch_stride) || mis(p.k, p.k_row_stride, p.k_batch_stride) ||
        mis(p.v, p.v_row_stride, p.v_batch_stride) || mis(p.dout, p.do_row_stride, p.do_batch_stride) ||
        mis(p.dq, p.dq_row_stride, p.dq_batch_stride) || reinterpret_cast<uintptr_t>(p.dk_part) % 16 != 0 ||
        reinterpret_cast<uintptr_t>(p.dv_part) % 16 != 0)
        return ZIGMA_ERR_STRIDE;
    const int tiles = xb_tiles(p.seqlen);
    dim3 grid(p.chunks, p.heads, p.batch), block(64 * kXbWaves);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    // measured at B=64, L=1024, 8 heads, 77 keys (tools/attn_probe.py): 2 waves per SIMD without operand prefetch 147 us, with 194 us
    // (45 spilled registers); 1 wave per SIMD 171 us without / 188 us with prefetch
    if (p.n_ctx <= 80) hipLaunchKernelGGL((cross_attn_bwd_kernel<5, 2, false>), grid, block, 0, stream, p, tiles);
    else hipLaunchKernelGGL((cross_attn_bwd_kernel<8, 1, false>), grid, block, 0, stream, p, tiles);
    set_last_kernel("cross_attn_bwd_mfma");
    return check_launch();
}
