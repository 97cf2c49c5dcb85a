// Everything between the in_proj GEMM and the selective scan in ONE kernel, gfx950.  C ABI: zigma_mamba_pre_fwd.
//
// Replaces, for the token-major fused block, the three launches of MambaInnerFn.forward
// (reference dis_mamba/mamba_ssm/ops/selective_scan_interface.py:311-335):
//     conv1d_out = causal_conv1d_fwd(x, w, b, silu)          (+ the zigzag gather xz[:, :, perm], mamba_simple.py:362-370)
//     x_dbl      = F.linear(conv1d_out^T, x_proj_weight)     K = d_inner, N = dt_rank + 2 d_state  (72)
//     delta      = delta_proj_weight @ x_dbl[:, :R]^T        K = dt_rank (40), + softplus(delta + bias) of the scan prologue
// As separate launches that segment moves 168 MB four times and 9 MB twice (read x, write u, re-read u, write delta)
// through three kernels with skinny GEMM shapes (81 + 54 + 68 us on MI355X at B=64).  Fused, u is consumed from LDS
// by the matrix cores while it is on its way to HBM: read x once, write u once, write delta once.
//
// One WAVE owns 16 consecutive scan positions and ALL d_inner channels; a workgroup = 4 waves = 64 positions that
// share the weight slabs through LDS (streaming them per wave from L2 costs 1.5 GB of L2->L1 traffic per launch and
// was measured to dominate):
//   for each 256-channel slab:  conv (lane = 4 adjacent channels, the 19 gathered rows of the tile in flight since
//       the previous slab, exactly conv_tok_kernel) -> u to HBM (8 B per lane, 512 B per row) and, as bf16, to the
//       wave's LDS tile; Wx[:, slab] (72 x 256) staged to LDS by the whole workgroup
//       -> 8 x 5 v_mfma_f32_16x16x32_bf16: acc[16 tokens x 80] += u_tile[16 x 256] * Wx_slab^T
//   x_dbl = bf16(acc) -> HBM (the scan reads its B_l / C_l columns from there) and, dt columns only, to LDS
//   for each 256-channel chunk of Wdt (staged to LDS in fragment order), for each 64-channel group: 4 accumulators
//       (channels 4c..4c+3 of lane column c, initialised with the bias) += x_dt[16 x R] * Wdt^T, softplus, one 8-byte
//       store per (lane, token): 128 contiguous bytes per row.
// bf16 only (MFMA operand type); fp32 accumulate everywhere; u and x_dbl are rounded to bf16 exactly where the
// reference's tensors are (conv output, GEMM output).
#include "zigma_common.h"

namespace zigma {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
using rsrc_t = __amdgpu_buffer_rsrc_t;

constexpr int kPreTok = 16;                       // scan positions per wave (MFMA M)
constexpr int kPreWaves = 4;
constexpr int kPreSlab = 256;                     // channels per conv pass of a wave (64 lanes x 4)
constexpr int kPrePitch = kPreSlab * 2 + 16;      // bytes per token row of the LDS tile (16 B skew)
constexpr int kPreXdPitch = 64 * 2 + 16;          // bytes per token row of the x_dt tile (reuses the same slice)
constexpr int kPreNB = 5;                         // 16-column blocks of x_dbl (<= 80 columns)

__device__ __forceinline__ bf16x8 ld_frag(const uint16_t *p) {
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(p));
}
__device__ __forceinline__ bf16x8 zero_frag() { return __builtin_bit_cast(bf16x8, make_uint4(0, 0, 0, 0)); }

constexpr int kPreWRows = 16 * kPreNB;            // rows of the staged Wx slab (rows >= dt_rank + 2 d_state stay zero)
constexpr int kPreWBytes = kPreWRows * kPrePitch;  // 42 240 B; the Wdt chunks (256 rows x <= 144 B) reuse it

template <typename WT, int W>
__global__ __launch_bounds__(64 * kPreWaves, 2) void mamba_pre_kernel(const zigma_pre_params_t p) {
    constexpr int NR = kPreTok + W - 1;
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    __shared__ __attribute__((aligned(16))) unsigned char s_tile[kPreWaves][kPreTok * kPrePitch];
    __shared__ __attribute__((aligned(16))) unsigned char s_w[kPreWBytes];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int b = blockIdx.y, L = p.seqlen, Di = p.dim, R = p.dt_rank, XC = p.dt_rank + 2 * p.dstate;
    const int k0r = (blockIdx.x * kPreWaves + wave) * kPreTok;
    const bool active = k0r < L;                  // wave-uniform; idle waves still stage weights and hit the barriers
    const int k0 = active ? k0r : 0;
    unsigned char *tile = s_tile[wave];
    const int i16 = lane & 15, g = lane >> 4;     // MFMA fragment row / column index, k-group (A, B) or row group (D)

    const int x_ls = static_cast<int>(p.x_l_stride) * 2, u_ls = static_cast<int>(p.u_l_stride) * 2;
    const int64_t Lm1 = L - 1;
    const rsrc_t x_rs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<uint16_t *>(reinterpret_cast<const uint16_t *>(p.x) + b * p.x_batch_stride), 0,
        static_cast<int>(Lm1 * x_ls + static_cast<int64_t>(Di) * 2), 0x00020000);
    const rsrc_t u_rs = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<uint16_t *>(p.u) + b * p.u_batch_stride, 0,
        active ? static_cast<int>(Lm1 * u_ls + static_cast<int64_t>(Di) * 2) : 0, 0x00020000);   // idle wave: stores dropped

    int rowv;                                     // lane i <- input row of scan position k0 - (W-1) + i
    {
        int k = k0 - (W - 1) + lane;
        k = k < 0 ? 0 : (k < L ? k : L - 1);
        rowv = p.x_row_index ? p.x_row_index[k] : k;
    }
    const uint16_t *wx = reinterpret_cast<const uint16_t *>(p.xproj_weight);

    f32x4 acc[kPreNB];
#pragma unroll
    for (int nb = 0; nb < kPreNB; ++nb) acc[nb] = f32x4{0.f, 0.f, 0.f, 0.f};

    u2 raw[NR];
    float w[4][W], cb[4];
    // rows + conv weights of one slab; issued one slab AHEAD (right after the previous slab's conv consumed the
    // registers) so that the HBM latency runs under the previous slab's MFMA phase
    auto issue_slab = [&](int c_slab) {
        const int c0 = c_slab + lane * 4;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int row = __builtin_amdgcn_readlane(rowv, i);
            raw[i] = u2{0u, 0u};
            if (k0 - (W - 1) + i >= 0) raw[i] = __builtin_amdgcn_raw_buffer_load_b64(x_rs, static_cast<unsigned>(c0) * 2, row * x_ls, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int j = 0; j < W; ++j) w[i][j] = ld<WT>(p.conv_weight, (c0 + i) * p.conv_w_c_stride + j * p.conv_w_width_stride);
            cb[i] = p.conv_bias ? ld<WT>(p.conv_bias, c0 + i) : 0.f;
        }
    };
    // Wx[:, slab] -> registers (9 x 16 B per thread: piece = tid + 256 i -> row piece / 32, 16-byte column piece % 32)
    constexpr int kStage = kPreWRows * (kPreSlab / 8) / (64 * kPreWaves);   // 10
    uint4 stg[kStage];
    auto stage_load = [&](int c_slab) {
#pragma unroll
        for (int i = 0; i < kStage; ++i) {
            const int piece = tid + 64 * kPreWaves * i, row = piece >> 5, col = piece & 31;
            stg[i] = make_uint4(0, 0, 0, 0);
            if (row < XC) stg[i] = *reinterpret_cast<const uint4 *>(wx + static_cast<int64_t>(row) * p.xproj_w_row_stride + c_slab + col * 8);
        }
    };
    auto stage_store = [&]() {
#pragma unroll
        for (int i = 0; i < kStage; ++i) {
            const int piece = tid + 64 * kPreWaves * i, row = piece >> 5, col = piece & 31;
            *reinterpret_cast<uint4 *>(s_w + row * kPrePitch + col * 16) = stg[i];
        }
    };
    issue_slab(0);
    stage_load(0);

#pragma unroll 1
    for (int c_slab = 0; c_slab < Di; c_slab += kPreSlab) {
        const int c0 = c_slab + lane * 4;
        stage_store();                            // previous slab's MFMA phase ended with a barrier
        // ---- conv + SiLU of 16 positions x 4 channels per lane -------------------------------------------------
#pragma unroll
        for (int j = 0; j < kPreTok; ++j) {
            float o[4] = {cb[0], cb[1], cb[2], cb[3]};
#pragma unroll
            for (int t = 0; t < W; ++t) {
                const u2 r = raw[j + t];
                o[0] += w[0][t] * __uint_as_float(r[0] << 16);
                o[1] += w[1][t] * __uint_as_float(r[0] & 0xffff0000u);
                o[2] += w[2][t] * __uint_as_float(r[1] << 16);
                o[3] += w[3][t] * __uint_as_float(r[1] & 0xffff0000u);
            }
            if (p.silu_activation) {
#pragma unroll
                for (int i = 0; i < 4; ++i) o[i] = silu(o[i]);
            }
            const u2 pk = {static_cast<unsigned>(from_float<BF16>(o[0])) | (static_cast<unsigned>(from_float<BF16>(o[1])) << 16),
                           static_cast<unsigned>(from_float<BF16>(o[2])) | (static_cast<unsigned>(from_float<BF16>(o[3])) << 16)};
            if (!(p.flags & 4)) __builtin_amdgcn_raw_buffer_store_b64(pk, u_rs, static_cast<unsigned>(c0) * 2, (k0 + j) * u_ls, 0);
            *reinterpret_cast<u2 *>(tile + j * kPrePitch + lane * 8) = pk;
        }
        __syncthreads();                          // u tile + Wx slab visible
        if (c_slab + kPreSlab < Di) {
            issue_slab(c_slab + kPreSlab);
            stage_load(c_slab + kPreSlab);
        }
        // ---- x_proj partial: acc += u_tile[16 x 256] * Wx[:, slab]^T -------------------------------------------------
        if (!(p.flags & 1)) {
#pragma unroll
            for (int s = 0; s < kPreSlab / 32; ++s) {
                const bf16x8 a = ld_frag(reinterpret_cast<const uint16_t *>(tile + i16 * kPrePitch + s * 64 + g * 16));
#pragma unroll
                for (int nb = 0; nb < kPreNB; ++nb) {
                    const bf16x8 bf = ld_frag(reinterpret_cast<const uint16_t *>(s_w + (nb * 16 + i16) * kPrePitch + s * 64 + g * 16));
                    acc[nb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bf, acc[nb], 0, 0, 0);
                }
            }
        }
        __syncthreads();                          // the next slab overwrites the tile and s_w
    }

    // ---- x_dbl: D layout = column i16 (of block nb), rows 4g + r -----------------------------------------------------
    uint16_t *xd = reinterpret_cast<uint16_t *>(p.x_dbl) + b * p.xdbl_batch_stride;
#pragma unroll
    for (int nb = 0; nb < kPreNB; ++nb) {
        const int n = nb * 16 + i16;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int tok = 4 * g + r;
            const uint16_t v = from_float<BF16>(acc[nb][r]);
            if (active && n < XC) xd[static_cast<int64_t>(k0 + tok) * p.xdbl_l_stride + n] = v;
            if (nb < 4) *reinterpret_cast<uint16_t *>(tile + tok * kPreXdPitch + n * 2) = n < R ? v : static_cast<uint16_t>(0);
        }
    }
    if (p.flags & 2) return;

    // ---- dt_proj + bias + softplus -------------------------------------------------------------------------------------
    // Wdt chunk (256 channels x R) staged in FRAGMENT order: channel cg + 4 i + q -> slot (cg/64)*64 + q*16 + i, so the
    // 16 lanes of a fragment read 16 consecutive slots; slot pitch R*2 (+16) bytes with (pitch/16) odd: conflict-free.
    const uint16_t *wdt = reinterpret_cast<const uint16_t *>(p.dtproj_weight);
    const float *dtb = reinterpret_cast<const float *>(p.dt_bias);
    const int d_ls = static_cast<int>(p.delta_l_stride) * 2;
    const rsrc_t d_rs = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<uint16_t *>(p.delta) + b * p.delta_batch_stride, 0,
        active ? static_cast<int>(Lm1 * d_ls + static_cast<int64_t>(Di) * 2) : 0, 0x00020000);
    const int r16 = R / 8;                                            // 16-byte pieces per weight row
    const int wpitch = (r16 | 1) * 16;                                // odd number of 16-byte units
    const bool k0_live = 8 * g < R, k1_live = 32 + 8 * g < R;
    bf16x8 a0, a1;
#pragma unroll 1
    for (int cc = 0; cc < Di; cc += kPreSlab) {
        for (int piece = tid; piece < kPreSlab * r16; piece += 64 * kPreWaves) {
            const int c = piece / r16, kp = piece - c * r16;          // channel within the chunk, 16-byte piece of its row
            const int slot = (c & ~63) + (c & 3) * 16 + ((c & 63) >> 2);
            *reinterpret_cast<uint4 *>(s_w + slot * wpitch + kp * 16) =
                *reinterpret_cast<const uint4 *>(wdt + static_cast<int64_t>(cc + c) * p.dtproj_w_row_stride + kp * 8);
        }
        __syncthreads();                          // also orders the x_dt tile writes above before the reads below
        if (cc == 0) {
            a0 = ld_frag(reinterpret_cast<const uint16_t *>(tile + i16 * kPreXdPitch + g * 16));
            a1 = ld_frag(reinterpret_cast<const uint16_t *>(tile + i16 * kPreXdPitch + 64 + g * 16));
        }
#pragma unroll 2
        for (int cgl = 0; cgl < kPreSlab; cgl += 64) {
            const int cg = cc + cgl;
            f32x4 d[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned char *slot = s_w + (cgl + q * 16 + i16) * wpitch;
                const float bv = dtb ? dtb[cg + 4 * i16 + q] : 0.f;
                bf16x8 b0 = zero_frag(), b1 = zero_frag();
                if (k0_live) b0 = ld_frag(reinterpret_cast<const uint16_t *>(slot + g * 16));
                if (k1_live) b1 = ld_frag(reinterpret_cast<const uint16_t *>(slot + 64 + g * 16));
                d[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a0, b0, f32x4{bv, bv, bv, bv}, 0, 0, 0);
                if (R > 32) d[q] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a1, b1, d[q], 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                float v[4] = {d[0][r], d[1][r], d[2][r], d[3][r]};
                if (p.delta_softplus) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) v[q] = softplus20(v[q]);
                }
                const u2 pk = {static_cast<unsigned>(from_float<BF16>(v[0])) | (static_cast<unsigned>(from_float<BF16>(v[1])) << 16),
                               static_cast<unsigned>(from_float<BF16>(v[2])) | (static_cast<unsigned>(from_float<BF16>(v[3])) << 16)};
                __builtin_amdgcn_raw_buffer_store_b64(pk, d_rs, static_cast<unsigned>(cg + 4 * i16) * 2, (k0 + 4 * g + r) * d_ls, 0);
            }
        }
        __syncthreads();                          // next chunk overwrites s_w
    }
}

static bool fits31(int64_t rows_m1, int64_t pitch_elems, int64_t row_elems) {
    return rows_m1 * pitch_elems * 2 + row_elems * 2 < (int64_t(1) << 31);
}

}  // namespace zigma

using namespace zigma;

extern "C" int zigma_mamba_pre_fwd(const zigma_pre_params_t *pp, void *stream_) {
    if (!pp) return ZIGMA_ERR_NULL;
    (void)hipGetLastError();
    const zigma_pre_params_t &p = *pp;
    if (p.batch < 0 || p.dim < 0 || p.seqlen < 0 || p.dt_rank < 1 || p.dstate < 1) return ZIGMA_ERR_SHAPE;
    if (p.batch == 0 || p.dim == 0 || p.seqlen == 0) return ZIGMA_OK;
    if (!p.x || !p.conv_weight || !p.xproj_weight || !p.dtproj_weight || !p.u || !p.x_dbl || !p.delta) return ZIGMA_ERR_NULL;
    if (p.io_dtype != ZIGMA_BF16) return ZIGMA_ERR_DTYPE;
    if (p.w_dtype != ZIGMA_BF16 && p.w_dtype != ZIGMA_F32) return ZIGMA_ERR_DTYPE;
    if (p.width != 4 || p.dim % kPreSlab != 0 || p.seqlen % kPreTok != 0 || p.dt_rank % 8 != 0 || p.dt_rank > 64 ||
        p.dt_rank + 2 * p.dstate > 16 * kPreNB || p.batch > 65535)
        return ZIGMA_ERR_SHAPE;
    auto mis = [](const void *q, uintptr_t a) { return reinterpret_cast<uintptr_t>(q) % a != 0; };
    if (p.x_l_stride % 4 != 0 || p.x_batch_stride % 4 != 0 || p.u_l_stride % 4 != 0 || p.u_batch_stride % 4 != 0 ||
        p.delta_l_stride % 4 != 0 || p.delta_batch_stride % 4 != 0 || p.xproj_w_row_stride % 8 != 0 ||
        p.dtproj_w_row_stride % 8 != 0 || mis(p.x, 8) || mis(p.u, 8) || mis(p.delta, 8) || mis(p.xproj_weight, 16) ||
        mis(p.dtproj_weight, 16) || mis(p.x_dbl, 2))
        return ZIGMA_ERR_STRIDE;
    if (!fits31(p.seqlen - 1, p.x_l_stride, p.dim) || !fits31(p.seqlen - 1, p.u_l_stride, p.dim) ||
        !fits31(p.seqlen - 1, p.delta_l_stride, p.dim))
        return ZIGMA_ERR_STRIDE;
    const int tok_per_wg = kPreTok * kPreWaves;
    dim3 grid((p.seqlen + tok_per_wg - 1) / tok_per_wg, p.batch), block(64 * kPreWaves);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (p.w_dtype == ZIGMA_BF16) hipLaunchKernelGGL((mamba_pre_kernel<BF16, 4>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((mamba_pre_kernel<F32, 4>), grid, block, 0, stream, p);
    set_last_kernel("mamba_pre_mfma");
    return check_launch();
}
