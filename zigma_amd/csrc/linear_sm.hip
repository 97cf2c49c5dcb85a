// linear_sm: out = x @ W^T for FEW tokens (serving-size batches, the video model at B = 2), gfx950 — out_proj of the ZigMa block below the
// token floor of the 4-wave tiled kernel.  Reference call site: selective_scan_interface.py:365 (F.linear).
//
// Why a third tile shape.  At 8192 tokens x 640 features the 256 x 128 tiles of linear_tn_kernel are 160 workgroups on 256 CUs, the
// 256 x 256 tiles of linear4w 96; the weight-stationary kernel with 128-feature panels spends a third of its time loading its panel.  All
// three lose to hipBLASLt there (27 / - / 25 us against 22; 48 / - / 37 against 34 at 16 384 tokens).  The product is small enough that tile
// QUANTISATION decides: with tiles of 128 tokens x N / 4 features (160 at E = 640, 192 at E = 768) 8192 tokens are exactly 256 tiles — one
// per CU, one round — and 16 384 tokens exactly two.
//
//   workgroup = 4 waves = one tile; wave w owns tokens [32 w, 32 w + 32) x all NBLK feature blocks of 32: NBLK accumulators of
//   v_mfma_f32_32x32x16_bf16, evaluated transposed like the other projection kernels (D[n][m]: W rows are the A operand, tokens the B
//   operand), so a lane ends up with 4 consecutive features of ONE token per accumulator quad.
//   BK = 64, four LDS stages of (32 NBLK + 128) rows x 128 B filled by global_load_lds_dwordx4 three k-steps ahead (the bank swizzle
//   — 16-byte slot ^= (row >> 1) & 7 — is applied to the per-lane SOURCE address and again on the fragment reads); ONE raw s_barrier
//   per k-step behind a counted s_waitcnt vmcnt.
//   Epilogue: accumulators -> bf16 -> the wave's LDS tile (32 tokens x 32 NBLK features, 16-byte padded pitch) -> 16-byte stores along
//   the token rows.
// Limits: bf16, no activation (bias and the gated residual: template flag EPI), k % 64 == 0 and k >= 128, m % 128 == 0, n % (32 NBLK) == 0 for NBLK = 5, 6 or 4 (tried in that order).
#include "scan_helpers.h"

namespace zigma {
namespace lsm {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) unsigned char *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *gbl_ptr_t;

constexpr int kBM = 128, kBK = 64, kNST = 4;       // four stages: loads three k-steps ahead (a direct-to-LDS load lands ~1.1 us after its issue)

template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }

// EPI: + bias (fp32, before the single rounding) and / or the block's gated branch add out = residual + gate[sample] * bf16(x W^T + bias) (reference
// model_zigma.py:441-449) — the arithmetic and rounding points of linear_tn_kernel's epilogue (csrc/linear.hip), so the two agree bit for bit.
template <int NBLK, bool EPI = false>
__global__ __launch_bounds__(256) void linear_sm_kernel(const zigma_linear_params_t p, const int tiles_n) {
    constexpr int BN = 32 * NBLK, ROWS = BN + kBM, STAGE = ROWS * 128;        // bytes per stage: W rows first, then token rows
    constexpr int NLD = ROWS / 32;                                            // direct-to-LDS loads per wave and stage (8 rows each, 4 waves)
    constexpr int PITCH = BN * 2 + 16;                                        // epilogue tile: bytes per token row (padded)
    static_assert(ROWS % 32 == 0, "whole load instructions");
    static_assert(kNST * STAGE <= 160 * 1024 && 4 * 32 * PITCH <= kNST * STAGE, "LDS");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[kNST * STAGE];   // ONE LDS object (cdna_hip_programming.md §5)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, kh = lane >> 5;
    // tile of this workgroup: consecutive workgroup ids go round the 8 XCDs, so XCD x takes the contiguous eighth [x T / 8, (x + 1) T / 8) of the
    // (m-tile, n-tile) raster — the n-tiles of a token panel share one L2
    int tile = blockIdx.x;
    if ((gridDim.x & 7) == 0) tile = (blockIdx.x & 7) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    const int mt = tile / tiles_n, nt = tile - mt * tiles_n;
    const int nk = p.k / kBK;
    const int64_t x_pitch = p.x_row_stride * 2, w_pitch = p.w_row_stride * 2, o_pitch = p.out_row_stride * 2;
    const unsigned char *wb = reinterpret_cast<const unsigned char *>(p.w) + static_cast<int64_t>(nt) * BN * w_pitch;
    const unsigned char *xb = reinterpret_cast<const unsigned char *>(p.x) + static_cast<int64_t>(mt) * kBM * x_pitch;

    // staging: instruction i of wave w fills the 8 rows q * 8 .. q * 8 + 7, q = i * 4 + w, of the stage (lane -> row q * 8 + (lane >> 3),
    // 16-byte piece lane & 7).  The source piece is swizzled by the row: (lane & 7) ^ ((row >> 1) & 7), row & 15 = (w & 1) * 8 + (lane >> 3).
    const int srow = (wave & 1) * 8 + (lane >> 3);
    const unsigned piece = static_cast<unsigned>(((lane & 7) ^ ((srow >> 1) & 7)) << 4);
    // one direct-to-LDS instruction: piece i of the batch of k-step kt -> its stage.  k-steps past the end are clamped to the last one: a harmless
    // refill of a free stage that keeps the loop body branch-free and the counted wait uniform (the epilogue drains them before it reuses the LDS)
    auto issue_piece = [&](int kt, int i) {
        const int ks_src = kt < nk ? kt : nk - 1;
        unsigned char *dst = smem + (kt % kNST) * STAGE;
        const int q = i * 4 + wave, row = q * 8 + (lane >> 3);                  // row of the stage this lane's 16 bytes belong to
        const unsigned char *src = row < BN ? wb + static_cast<int64_t>(row) * w_pitch : xb + static_cast<int64_t>(row - BN) * x_pitch;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + ks_src * (kBK * 2) + piece), (lds_ptr_t)(dst) + q * 1024, 16, 0, 0);
    };
    // fragment reads: row * 128 + (((ks << 1) | kh) ^ ((row >> 1) & 7)) * 16; every row base used here is a multiple of 32, so (row >> 1) & 7 = (j >> 1) & 7
    const int sw = (j >> 1) & 7;
    const int a_row0 = j * 128, b_row0 = (BN + wave * 32 + j) * 128;

    f32x16 acc[NBLK];
#pragma unroll
    for (int nb = 0; nb < NBLK; ++nb) acc[nb] = f32x16{};

#pragma unroll
    for (int b0 = 0; b0 < kNST - 1; ++b0)
#pragma unroll
        for (int i = 0; i < NLD; ++i) issue_piece(b0, i);
    constexpr int PPS = (NLD + 2) / 3;                      // pieces of the next batch per sub-step (sub-steps 0 .. 2)
#pragma unroll 1
    for (int kt = 0; kt < nk; ++kt) {
        // stage kt has landed when only the two younger batches are outstanding — for this wave's parts; the barrier makes it true for all
        wait_vm<2 * NLD>();
        __builtin_amdgcn_s_barrier();                      // ... and every wave is done with stage kt - 1, which the next batch overwrites
        const unsigned char *sb = smem + (kt % kNST) * STAGE;
        // fragments one k-substep ahead of the MFMAs that use them, and the direct-to-LDS pieces of k-step kt + 3 spread between the MFMAs of the
        // first three sub-steps: with one wave per SIMD nothing else covers an LDS round trip or the ~55 cycles a piece takes to issue
        bf16x8 fa[2][NBLK], fb[2];
        auto frags = [&](int ks, bf16x8 (&a)[NBLK], bf16x8 &b) {
            const int off = ((((ks << 1) | kh) ^ sw) << 4);
            b = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(sb + b_row0 + off));
#pragma unroll
            for (int nb = 0; nb < NBLK; ++nb)
                a[nb] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(sb + a_row0 + nb * 32 * 128 + off));
        };
        frags(0, fa[0], fb[0]);
#pragma unroll
        for (int ks = 0; ks < kBK / 16; ++ks) {
            if (ks + 1 < kBK / 16) frags(ks + 1, fa[(ks + 1) & 1], fb[(ks + 1) & 1]);
#pragma unroll
            for (int i = ks * PPS; i < (ks + 1) * PPS && i < NLD && ks < 3; ++i) issue_piece(kt + kNST - 1, i);
#pragma unroll
            for (int nb = 0; nb < NBLK; ++nb)
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[ks & 1][nb], fb[ks & 1], acc[nb], 0, 0, 0);
        }
        // pin the order (masks: 0x100 DS read, 0x008 MFMA): behind every MFMA of sub-steps 0 .. 2 one fragment read of the next sub-step (two
        // behind the first); the direct-to-LDS pieces stay where the source puts them, between the sub-steps
        {
            __builtin_amdgcn_sched_group_barrier(0x100, NBLK + 1, 0);
#pragma unroll
            for (int ks = 0; ks + 1 < kBK / 16; ++ks) {
#pragma unroll
                for (int r = 0; r < NBLK; ++r) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    if (r == 0) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
                    else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
            }
            __builtin_amdgcn_sched_group_barrier(0x008, NBLK, 0);
        }
    }
    wait_vm<0>();                                          // the refills past the end: they must not land in the epilogue's tiles
    // ---- epilogue: D[i][jj], jj = token (lane & 31), i = feature = (r & 3) + 8 (r >> 2) + 4 (lane >> 5): the wave transposes its 32 x BN tile
    // through LDS (the stages are free: barrier) and stores 16-byte pieces along the token rows
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    unsigned char *scr = smem + wave * (32 * PITCH);
    const uint16_t *biasp = EPI ? reinterpret_cast<const uint16_t *>(p.bias) : nullptr;
#pragma unroll
    for (int nb = 0; nb < NBLK; ++nb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            float v[4] = {acc[nb][q * 4], acc[nb][q * 4 + 1], acc[nb][q * 4 + 2], acc[nb][q * 4 + 3]};
            if (EPI && biasp) {
                const uint2 bq = *reinterpret_cast<const uint2 *>(biasp + nt * BN + nb * 32 + q * 8 + kh * 4);
                v[0] += __uint_as_float(bq.x << 16);
                v[1] += __uint_as_float(bq.x & 0xffff0000u);
                v[2] += __uint_as_float(bq.y << 16);
                v[3] += __uint_as_float(bq.y & 0xffff0000u);
            }
            uint2 pk;
            pk.x = static_cast<uint32_t>(from_float<BF16>(v[0])) | (static_cast<uint32_t>(from_float<BF16>(v[1])) << 16);
            pk.y = static_cast<uint32_t>(from_float<BF16>(v[2])) | (static_cast<uint32_t>(from_float<BF16>(v[3])) << 16);
            *reinterpret_cast<uint2 *>(scr + j * PITCH + (nb * 32 + q * 8 + kh * 4) * 2) = pk;
        }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();                       // wave-private tile: writes and reads of one wave
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    constexpr int PPR = BN * 2 / 16;                       // 16-byte pieces per token row
    const int64_t row0 = static_cast<int64_t>(mt) * kBM + wave * 32;
    unsigned char *ob = reinterpret_cast<unsigned char *>(p.out) + row0 * o_pitch + static_cast<int64_t>(nt) * BN * 2;
    // gated residual: the 128-token tile lies inside one sample (rows_per_batch % 256 == 0): one gate row
    const unsigned char *rb = nullptr, *gb = nullptr;
    int64_t r_pitch = 0;
    if (EPI && p.residual) {
        r_pitch = p.res_row_stride * 2;
        rb = reinterpret_cast<const unsigned char *>(p.residual) + row0 * r_pitch + static_cast<int64_t>(nt) * BN * 2;
        gb = reinterpret_cast<const unsigned char *>(p.gate) + ((static_cast<int64_t>(mt) * kBM) / p.rows_per_batch) * p.gate_batch_stride * 2 + static_cast<int64_t>(nt) * BN * 2;
    }
#pragma unroll
    for (int it = 0; it < (32 * PPR + 63) / 64; ++it) {
        const int idx = it * 64 + lane;
        if (idx < 32 * PPR) {
            const int tok = idx / PPR, pc = idx - tok * PPR;
            uint4 v = *reinterpret_cast<const uint4 *>(scr + tok * PITCH + pc * 16);
            if (EPI && rb) {
                const uint4 rs = *reinterpret_cast<const uint4 *>(rb + tok * r_pitch + pc * 16);
                const uint4 gt = *reinterpret_cast<const uint4 *>(gb + pc * 16);
                const unsigned vv[4] = {v.x, v.y, v.z, v.w}, rr[4] = {rs.x, rs.y, rs.z, rs.w}, gg[4] = {gt.x, gt.y, gt.z, gt.w};
                unsigned oo[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float lo = __builtin_fmaf(__uint_as_float(gg[e] << 16), __uint_as_float(vv[e] << 16), __uint_as_float(rr[e] << 16));
                    const float hi = __builtin_fmaf(__uint_as_float(gg[e] & 0xffff0000u), __uint_as_float(vv[e] & 0xffff0000u), __uint_as_float(rr[e] & 0xffff0000u));
                    oo[e] = static_cast<uint32_t>(from_float<BF16>(lo)) | (static_cast<uint32_t>(from_float<BF16>(hi)) << 16);
                }
                v = make_uint4(oo[0], oo[1], oo[2], oo[3]);
            }
            *reinterpret_cast<uint4 *>(ob + tok * o_pitch + pc * 16) = v;
        }
    }
}

}  // namespace lsm

// feature blocks per tile (5: n % 160 == 0; 6: n % 192 == 0; 4: n % 128 == 0) the few-token kernel uses for the call, or 0 if it does not serve it
static int linear_sm_blocks(const zigma_linear_params_t &p) {
    if (p.silu_from_col < p.n) return 0;
    if (p.k % 64 != 0 || p.k < 128 || p.m % 128 != 0 || p.m < 128) return 0;
    if (p.out_row_stride % 8 != 0 || reinterpret_cast<uintptr_t>(p.out) % 16 != 0) return 0;
    if (128 * p.x_row_stride * 2 > 0x7fffffff || 192 * p.w_row_stride * 2 > 0x7fffffff) return 0;
    if (p.bias && reinterpret_cast<uintptr_t>(p.bias) % 8 != 0) return 0;
    if (p.residual) {       // (zigma_linear_fwd has checked pointers, pitches and rows_per_batch % 256 == 0 already)
        if (!p.gate || p.rows_per_batch % 128 != 0) return 0;
    }
    const int nblk = p.n % 160 == 0 ? 5 : p.n % 192 == 0 ? 6 : p.n % 128 == 0 ? 4 : 0;
    if (!nblk) return 0;
    if ((p.m / 128) * (p.n / (32 * nblk)) > 0x7fffffff) return 0;
    return nblk;
}

bool linear_sm_eligible(const zigma_linear_params_t &p) { return linear_sm_blocks(p) != 0; }

int launch_linear_sm(const zigma_linear_params_t &p, hipStream_t stream) {
    const int nblk = linear_sm_blocks(p);
    if (!nblk) return ZIGMA_ERR_UNSUPPORTED;
    const int tiles_n = p.n / (32 * nblk);
    const dim3 grid(static_cast<unsigned>((p.m / 128) * tiles_n)), block(256);
    const bool epi = p.bias || p.residual;
#define ZIGMA_LSM(N_) do { if (epi) hipLaunchKernelGGL((lsm::linear_sm_kernel<N_, true>), grid, block, 0, stream, p, tiles_n); \
                           else hipLaunchKernelGGL((lsm::linear_sm_kernel<N_, false>), grid, block, 0, stream, p, tiles_n); } while (0)
    if (nblk == 5) ZIGMA_LSM(5); else if (nblk == 6) ZIGMA_LSM(6); else ZIGMA_LSM(4);
#undef ZIGMA_LSM
    set_last_kernel(nblk == 5 ? "linear_sm_128x160" : nblk == 6 ? "linear_sm_128x192" : "linear_sm_128x128");
    return check_launch();
}

}  // namespace zigma
