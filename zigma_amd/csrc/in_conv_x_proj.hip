// in_conv_x_proj: the x half of Mamba.in_proj, the zigzag gather, the depthwise causal conv1d (+ bias, SiLU) and x_proj in ONE
// kernel, gfx950.  C ABI: zigma_in_conv_x_proj_fwd (+ _workspace_bytes).
//
// Replaces the x columns of F.linear(hidden_states, in_proj.weight) (reference mamba_simple.py:290-294), xz[:, :, perm]
// (mamba_simple.py:362-370), causal_conv1d_fn(..., "silu") and F.linear(conv1d_out, x_proj_weight)
// (selective_scan_interface.py:307-322).  As three kernels the path moves  write x + read x + write u  (3 x 168 MB at the headline
// shape) besides the operands; here x exists only as MFMA accumulators and 4 KB of LDS per wave:  read h (84 MB) + write u.
//
//   workgroup = 4 waves x 32 scan positions = a tile of 128 positions, one workgroup per CU (148 KB of LDS, ~330 registers), walking
//   `tiles` consecutive tiles of one sample.  A wave keeps the k-contiguous rows of ITS 32 tokens (picked through the row table) in
//   registers as MFMA B fragments for the whole tile (k / 16 x 4 registers); d_inner is walked in stages of 64 channels:
//     x^T[64 ch][32 tok] = W_in[64 ch][k] . h^T        v_mfma_f32_32x32x16_bf16, W_in rows as the A operand from LDS — the slab
//                                                       streams through a ring of four 16 KB pieces (64 channels x 128 k,
//                                                       global_load_lds_dwordx4, 16-byte slots XOR-swizzled by the row on the
//                                                       source side), three pieces ahead, ONE counted vmcnt + barrier per piece;
//     the accumulators leave as bf16 into the wave's 35-row staging tile (rows 3..34; rows 29..31 also into rows 0..2 of the next
//     wave's tile — the causal window —, the last wave's into a carry tile for the workgroup's next tile);
//     one stage LATER, between the MFMAs of the next stage's product, the conv + SiLU + x_proj of conv_x_proj.hip run from that
//     tile (lane = position, 8 adjacent channels; u is the A... B fragment of the x_proj MFMA as produced) and u leaves as full
//     128-byte lines through the wave's own staging rows.
//   The three x rows in front of a workgroup's FIRST tile (when that is not the start of a sequence) come from a pre-pass kernel
//   (16x16x32 MFMA over those rows only, 3 rows per workgroup = 1 % of the product) through the caller's workspace.
// bf16 only; width 4; bias required; seqlen % 128 == 0; k % 128 == 0, k <= 768; d_inner % 64 == 0, <= 1536; n <= 96, n % 8 == 0.
#include "zigma_common.h"

#include <type_traits>
#include <utility>

namespace zigma {
namespace icx {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) unsigned char *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *gbl_ptr_t;

constexpr int kTok = 32, kNW = 4, kTile = kTok * kNW, kBC = 64, kHalo = 3;
constexpr int kPiece = kBC * 128 * 2;                   // one ring slot: 64 channels x 128 k, bf16
constexpr int kNR = 4;
constexpr int kStg = 40 * 128;                          // staging tile of a wave (35 rows used), per stage parity
constexpr int kXw = 96 * 128 + 1024;                    // W_x slab (96 rows x 64 channels) + conv taps (512 B) + bias (128 B)
constexpr int kMaxStages = 24;
constexpr int kRingOff = 0;
constexpr int kXwOff = kRingOff + kNR * kPiece;         //  65536
constexpr int kStgOff = kXwOff + 2 * kXw;               //  92160
constexpr int kCarryOff = kStgOff + kNW * 2 * kStg;     // 133120
constexpr int kLds = kCarryOff + 2 * kMaxStages * 384;  // 151552

__device__ __forceinline__ float bf_lo(unsigned v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float bf_hi(unsigned v) { return __uint_as_float(v & 0xffff0000u); }
__device__ __forceinline__ float dot2(unsigned a, unsigned b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a), __builtin_bit_cast(bf16x2, b), c, false);
}
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {        // one v_cvt_pk_bf16_f32 (round to nearest even)
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}
// Every LDS access is inline assembly: a ds_read hipcc can see waits for EVERY direct-to-LDS load in flight (vmcnt(0)).  Landing
// is tracked by hand (counted vmcnt + barrier per piece), reads are settled by explicit lgkmcnt waits that name their registers.
__device__ __forceinline__ void lds_rd(u32x4 &d, unsigned addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr)); }
__device__ __forceinline__ void lds_wr16(unsigned addr, const u32x4 &v) { asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void lds_wr8(unsigned addr, const u32x2 &v) { asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory"); }

struct KStep {
    u32x4 X[4];      // input rows k-3 .. k, this lane's 8 channels
    u32x4 Wc[4];     // conv taps: [r] = channels 2r, 2r+1 as (tap0 tap1)(tap2 tap3) pairs
    u32x4 Bc;        // conv bias of the 8 channels
    u32x4 Bf[3];     // W_x fragments (rows nb * 32 + j)
};
__device__ __forceinline__ void tie(KStep &k) {          // (no instruction: pins every use of the chunk behind the wait in front of it)
    asm volatile("" : "+v"(k.X[0]), "+v"(k.X[1]), "+v"(k.X[2]), "+v"(k.X[3]), "+v"(k.Wc[0]), "+v"(k.Wc[1]), "+v"(k.Wc[2]), "+v"(k.Wc[3]),
                 "+v"(k.Bc), "+v"(k.Bf[0]), "+v"(k.Bf[1]), "+v"(k.Bf[2]));
}

// ---- pre-pass: the x rows of the three positions in front of every workgroup segment (pre-conv, bf16) ----------------------------
// grid (ceil(n_seg / 4), dim / 64); a workgroup = 4 segments x 3 rows (12 of the 16 MFMA columns) x 64 channels, wave = 16 channels.
// x^T[ch][row] = W_in[ch][k] . h[row][k]^T with v_mfma_f32_16x16x32_bf16, both operands straight from global memory.
template <int KS32>
__global__ __launch_bounds__(256) void in_halo_rows_kernel(const zigma_in_conv_xproj_params_t p, const int seg_len, const int n_seg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int n16 = lane & 15, kq = lane >> 4;
    const int seg = blockIdx.x * 4 + n16 / 3, r = n16 % 3;
    const int ch0 = blockIdx.y * 64 + wave * 16;
    bool valid = n16 < 12 && seg < n_seg;
    const int64_t pos_flat = static_cast<int64_t>(valid ? seg : 0) * seg_len;
    const int b = static_cast<int>(pos_flat / p.seqlen), t0 = static_cast<int>(pos_flat - static_cast<int64_t>(b) * p.seqlen);
    valid = valid && t0 >= kHalo;
    const int pos = valid ? t0 - kHalo + r : 0;
    const int row = p.x_row_index ? p.x_row_index[pos] : pos;
    const unsigned char *hs = reinterpret_cast<const unsigned char *>(p.h) + (static_cast<int64_t>(b) * p.h_batch_stride + static_cast<int64_t>(row) * p.h_l_stride) * 2 + kq * 16;
    const unsigned char *ws = reinterpret_cast<const unsigned char *>(p.w_in) + static_cast<int64_t>(ch0 + n16) * p.win_row_stride * 2 + kq * 16;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < KS32; ++kk) {
        const bf16x8 a = *reinterpret_cast<const bf16x8 *>(ws + kk * 64);
        const bf16x8 bb = *reinterpret_cast<const bf16x8 *>(hs + kk * 64);
        acc = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, bb, acc, 0, 0, 0);
    }
    if (valid) {      // D[m = channel 4 kq + i][n = row n16]
        const u32x2 pk = {pack_bf2(acc[0], acc[1]), pack_bf2(acc[2], acc[3])};
        *reinterpret_cast<u32x2 *>(reinterpret_cast<unsigned char *>(p.workspace) + ((static_cast<int64_t>(seg) * 3 + r) * p.dim + ch0 + kq * 4) * 2) = pk;
    }
}

// ---- the main kernel ---------------------------------------------------------------------------------------------------------------
// VM instructions issued between a piece's loads and the wait of the iteration that consumes it (see the tally note in the kernel):
// the xw loads ride on piece 0 of a stage (>= 2 per wave: counted as 2), the 4 u stores of the previous stage on piece 4.
constexpr int icx_b(int kp, int stores) { return (kp == 0 ? 2 : 0) + (kp == 4 ? stores : 0); }
constexpr int icx_wait(int kp, int np, int st_prev, int st_cur) {
    int n = 8;                                                       // the two younger pieces
    for (int d = 1; d <= 3; ++d) {
        const int idx = kp - d;
        n += idx < 0 ? icx_b(idx + np, st_prev) : icx_b(idx, st_cur);
    }
    return n;
}
template <int... I, typename F> __device__ __forceinline__ void static_for(std::integer_sequence<int, I...>, F &&f) {
    (f(std::integral_constant<int, I>{}), ...);
}
template <int N> __device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int KSUB, int PROBE>      // k / 16; probe bits (results wrong): 2 = no x product, 4 = no u stores, 8 = no conv arithmetic
__global__ __launch_bounds__(256, 1) void in_conv_x_proj_kernel(const zigma_in_conv_xproj_params_t p, const int tiles) {
    constexpr int NP = KSUB / 8;                                    // ring pieces per stage
    constexpr int dbg = PROBE;
    static_assert(KSUB % 8 == 0 && NP >= 5, "conv chunks 0..3 and the u stores ride on pieces 0..4");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[kLds];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, kh = lane >> 5;
    const int L = p.seqlen, n_stages = p.dim / kBC, n_out = p.n;
    const int64_t seg0 = static_cast<int64_t>(blockIdx.x) * tiles * kTile;      // first position (all samples) of this workgroup
    const int b = static_cast<int>(seg0 / L), t_seg = static_cast<int>(seg0 - static_cast<int64_t>(b) * L);
    const unsigned smem_lds = static_cast<unsigned>(reinterpret_cast<uintptr_t>((lds_ptr_t)(smem)));
    const int *tab = p.x_row_index;
    const unsigned char *hb = reinterpret_cast<const unsigned char *>(p.h) + static_cast<int64_t>(b) * p.h_batch_stride * 2 + kh * 16;
    const int64_t h_ls2 = p.h_l_stride * 2;
    const unsigned char *win = reinterpret_cast<const unsigned char *>(p.w_in);
    const int64_t win_rs2 = p.win_row_stride * 2;

    // ---- W_in ring: wave w fetches rows 16 w .. 16 w + 15 of a piece, 4 rows per instruction; lane -> (row lane >> 4, PHYSICAL slot
    // lane & 15) holding the LOGICAL slot (lane & 15) ^ (row & 15): a fragment read (32 rows, one logical slot) then hits 16
    // different 16-byte slots per lane group = every bank once
    unsigned woff[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = wave * 16 + 4 * i + (lane >> 4);
        woff[i] = static_cast<unsigned>(row * win_rs2) + ((((lane & 15) ^ (row & 15))) << 4);
    }
    int iss_s = 0, iss_kp = 0;                                       // (stage, piece) of the next piece to issue
    int issued_pieces = 0;
    // (always 4 VM instructions: the last three iterations of the run fetch pieces nobody reads, so that every wait is a constant)
    auto issue_piece = [&]() {
        if (dbg & 16) return;
        const unsigned char *src = win + static_cast<int64_t>(iss_s) * kBC * win_rs2 + iss_kp * 256;
        unsigned char *dst = smem + kRingOff + (issued_pieces % kNR) * kPiece + wave * 4096;
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + woff[i]), (lds_ptr_t)(dst) + i * 1024, 16, 0, 0);
        ++issued_pieces;
        if (++iss_kp == NP) { iss_kp = 0; if (++iss_s == n_stages) iss_s = 0; }
    };
    const unsigned a_off = j * 256 + ((kh ^ (j & 15)) << 4);         // fragment read: row j (+ 32 cb), logical slot 2 ks + kh

    // ---- W_x slab + conv taps / bias of a stage (as in conv_x_proj.hip): 8 rows per instruction, instruction q = wave + 4 i
    const unsigned char *wsrc[3];
    int n_w = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        int row = (wave + kNW * i) * 8 + (lane >> 3);
        row = row < n_out ? row : n_out - 1;
        wsrc[i] = reinterpret_cast<const unsigned char *>(p.w) + static_cast<int64_t>(row) * p.w_row_stride * 2 + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
        n_w += (wave + kNW * i) * 8 < n_out ? 1 : 0;
    }
    const unsigned char *csrc = lane < 32 ? reinterpret_cast<const unsigned char *>(p.conv_weight) + lane * 16
                                          : reinterpret_cast<const unsigned char *>(p.conv_bias) + ((lane < 40 ? lane : 39) - 32) * 16;
    const int c_step = lane < 32 ? kBC * 8 : kBC * 2;               // bytes per stage
    auto issue_xw = [&](int st) {
        unsigned char *dst = smem + kXwOff + (st & 1) * kXw;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (i < n_w)
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)(wsrc[i] + st * (kBC * 2)), (lds_ptr_t)(dst) + (wave + kNW * i) * 1024, 16, 0, 0);
        if (wave == 1)
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(csrc + static_cast<int64_t>(st) * c_step), (lds_ptr_t)(dst) + 96 * 128, 16, 0, 0);
    };

    // ---- per-lane LDS offsets of the conv side (staging tile of this wave, stage parity added at use)
    const unsigned stg0 = smem_lds + kStgOff + wave * 2 * kStg;
    unsigned x_off[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) x_off[s] = (j + s) * 128 + ((kh ^ (((j + s) >> 1) & 7)) << 4);
    const unsigned w_off = j * 128 + ((kh ^ ((j >> 1) & 7)) << 4);
    const unsigned u_wr = j * 128 + ((kh ^ ((j >> 1) & 7)) << 4);
    const unsigned u_rd = (lane >> 3) * 128 + (((lane & 7) ^ (lane >> 4)) << 4);
    // x rows out of the accumulators: token j -> row 3 + j; register group q of block cb = channels 32 cb + 8 q + 4 kh + 0..3
    // = logical slot 4 cb + q, byte 8 kh of the slot
    const unsigned xw_row = (kHalo + j) * 128 + kh * 8, xw_swz = ((kHalo + j) >> 1) & 7;
    const int jh = j - (kTok - kHalo);                               // >= 0: this token is part of the next wave's causal window
    const unsigned hw_row = (jh > 0 ? jh : 0) * 128 + kh * 8, hw_swz = ((jh > 0 ? jh : 0) >> 1) & 7;

    // ---- carry tile of the workgroup's first tile: the pre-pass rows, in the layout of staging rows 0..2 of every stage
    if (t_seg != 0) {
        for (int idx = tid; idx < n_stages * 24; idx += 256) {
            const int s = idx / 24, rem = idx - s * 24, r = rem >> 3, c = rem & 7;
            const u32x4 v = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const unsigned char *>(p.workspace) +
                                                             ((static_cast<int64_t>(blockIdx.x) * 3 + r) * p.dim + s * kBC + c * 8) * 2);
            lds_wr16(smem_lds + kCarryOff + s * 384 + r * 128 + ((c ^ ((r >> 1) & 7)) << 4), v);
        }
    }

    // vm tally: a piece is issued three iterations before the one that consumes it; VM_CNT retires in issue order, so its landing
    // is `s_waitcnt vmcnt(n)` with n = the VM instructions issued after it = the rest of the issuing iteration + the two iterations
    // in between (icx_wait()).  Under-counting is safe (it waits for more), so only what is certain is counted: 4 loads per piece,
    // 2 of the 2..3 xw loads, the 4 u stores of a stage that has a predecessor; the tile's tail, the x_dbl stores and the B-fragment
    // loads (one vmcnt(0) per tile) are not.
    issue_piece();
    issue_piece();
    issue_piece();

    u32x4 uq[4];
    f32x16 acc[3];
    f32x16 xacc[2];
    bf16x8 bfr[KSUB];
    int g = 0;                                                       // piece index over the whole run of the workgroup

    // conv side --------------------------------------------------------------------------------------------------------
    auto conv_reads = [&](KStep &k, const int sp, const int ks) {    // stage sp, k-step ks
        const unsigned sb = stg0 + (sp & 1) * kStg, xb = smem_lds + kXwOff + (sp & 1) * kXw;
#pragma unroll
        for (int s = 0; s < 4; ++s) lds_rd(k.X[s], (sb + x_off[s]) ^ (ks << 5));
#pragma unroll
        for (int r = 0; r < 4; ++r) lds_rd(k.Wc[r], xb + 96 * 128 + (ks * 2 + kh) * 64 + r * 16);
        lds_rd(k.Bc, xb + 96 * 128 + 512 + (ks * 2 + kh) * 16);
#pragma unroll
        for (int nb = 0; nb < 3; ++nb) lds_rd(k.Bf[nb], (xb + w_off + nb * 32 * 128) ^ (ks << 5));
    };
    // The conv arithmetic of a chunk is cut into its 8 outputs (one channel each, ~11 VALU instructions) so that the product loop can
    // place one output behind every MFMA pair: hipcc leaves a block of VALU code where the source puts it, and as one block of ~100
    // instructions the conv runs with the matrix pipe idle.
    unsigned ur[4];
    float lo_keep = 0.f;
    auto conv_mask = [&](KStep &k, const bool first) {
        if (first) {
#pragma unroll
            for (int s = 0; s < 3; ++s)
                if (j + s < kHalo) k.X[s] = u32x4{0, 0, 0, 0};
        }
    };
    auto conv_one = [&](KStep &k, const int idx) {                   // output idx: channel 2r (low halves) for even, 2r + 1 (high) for odd idx
        const int r = idx >> 1;
        const bool hi = idx & 1;
        if (dbg & 8) {
            ur[r] = k.X[3][r];
            return;
        }
        const unsigned sel = hi ? 0x07060302u : 0x05040100u;
        const unsigned p01 = __builtin_amdgcn_perm(k.X[1][r], k.X[0][r], sel);
        const unsigned p23 = __builtin_amdgcn_perm(k.X[3][r], k.X[2][r], sel);
        float a = dot2(p01, hi ? k.Wc[r].z : k.Wc[r].x, hi ? bf_hi(k.Bc[r]) : bf_lo(k.Bc[r]));
        a = dot2(p23, hi ? k.Wc[r].w : k.Wc[r].y, a);
        const float v = silu(a);
        if (!hi) lo_keep = v; else ur[r] = pack_bf2(lo_keep, v);
    };
    auto conv_fin = [&](KStep &k, const int ks) {                    // the 8 outputs of the chunk are the B fragment of the x_proj product
        const u32x4 u8 = {ur[0], ur[1], ur[2], ur[3]};
        uq[ks] = u8;
        const bf16x8 a = __builtin_bit_cast(bf16x8, u8);
#pragma unroll
        for (int nb = 0; nb < 3; ++nb)
            acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, k.Bf[nb]), a, acc[nb], 0, 0, 0);
    };
    // u of stage sp (32 positions x 64 channels) leaves as full 128-byte lines, transposed through the wave's own staging tile
    u32x4 ut[4];
    auto u_xpose = [&](const int sp) {                               // (LDS side: 4 writes, 4 reads; settled by the caller's next lgkmcnt wait)
        if (dbg & 4) return;
        const unsigned sb = stg0 + (sp & 1) * kStg;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) lds_wr16((sb + u_wr) ^ (ks << 5), uq[ks]);
#pragma unroll
        for (int i = 0; i < 4; ++i) lds_rd(ut[i], (sb + u_rd + i * 1024) ^ ((i & 1) << 6));
    };
    auto u_tie = [&]() { asm volatile("" : "+v"(ut[0]), "+v"(ut[1]), "+v"(ut[2]), "+v"(ut[3])); };
    auto u_store = [&](const int sp, unsigned char *ust) {
        if (dbg & 4) return;
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4 *>(ust + static_cast<int64_t>(i * 8) * (p.u_l_stride * 2) + sp * (kBC * 2)) = ut[i];
    };

    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");              // (the carry tile is written; the ring pieces stay in flight)
#pragma unroll 1
    for (int tile = 0; tile < tiles; ++tile) {
        const int t_wave = t_seg + tile * kTile + wave * kTok;       // first position of this wave inside the sample
        const bool first = t_wave == 0;                              // wave-uniform: the causal window starts inside this wave's rows
        unsigned char *ust = reinterpret_cast<unsigned char *>(p.u) + static_cast<int64_t>(b) * p.u_batch_stride * 2 +
                             static_cast<int64_t>(t_wave + (lane >> 3)) * (p.u_l_stride * 2) + (lane & 7) * 16;
        {   // this wave's 32 token rows as B fragments: lane = (token j, k-half kh), 16 bytes per k-step of 16
            const int pos = t_wave + j;
            const int row = tab ? tab[pos] : pos;
            const unsigned char *src = hb + static_cast<int64_t>(row) * h_ls2;
            // (inline assembly: beside loads it can see, hipcc drains every LDS-DMA in flight before each use)
#pragma unroll
            for (int kk = 0; kk < KSUB; ++kk) asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(bfr[kk]) : "v"(src + kk * 32));
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
            for (int kk = 0; kk < KSUB; kk += 8)
                asm volatile("" : "+v"(bfr[kk]), "+v"(bfr[kk + 1]), "+v"(bfr[kk + 2]), "+v"(bfr[kk + 3]), "+v"(bfr[kk + 4]), "+v"(bfr[kk + 5]),
                             "+v"(bfr[kk + 6]), "+v"(bfr[kk + 7]));
        }
#pragma unroll
        for (int nb = 0; nb < 3; ++nb) acc[nb] = f32x16{};
        const unsigned carry_rd = smem_lds + kCarryOff + (tile & 1) * (kMaxStages * 384);
        const unsigned carry_wr = smem_lds + kCarryOff + ((tile + 1) & 1) * (kMaxStages * 384);

        auto stage = [&](auto prev_tag, const int s) {
            constexpr bool PREV = decltype(prev_tag)::value;
            xacc[0] = f32x16{};
            xacc[1] = f32x16{};
            static_for(std::make_integer_sequence<int, NP>{}, [&](auto kp_tag) {
                constexpr int kp = decltype(kp_tag)::value;
                // piece g has landed (this wave's part: vmcnt; every wave's: barrier); every wave is done with piece g - 1 and with
                // its LDS writes of the previous stage (lgkmcnt)
                constexpr int ST = (PREV && !(dbg & 4)) ? 4 : 0;
                if (PREV && s >= 2) wait_vm<icx_wait(kp, NP, ST, ST)>(); else wait_vm<icx_wait(kp, NP, 0, ST)>();
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                issue_piece();
                if (kp == 0) issue_xw(s);
                const unsigned slot = smem_lds + kRingOff + (g % kNR) * kPiece + a_off;
                KStep kc;
                constexpr bool conv_here = PREV && kp < 4, store_here = PREV && kp == 4;
                if (conv_here) conv_reads(kc, s - 1, kp);
                if (store_here) u_xpose(s - 1);
                // A fragments two k-steps ahead (three register sets); LDS returns in order, so `lgkmcnt(4)` = everything but the
                // two younger k-steps' reads = this k-step's fragments AND the chunk / transposition reads issued in front of them
                u32x4 A[3][2];
                if (!(dbg & 32)) {
                    lds_rd(A[0][0], slot);
                    lds_rd(A[0][1], slot + 8192);
                    lds_rd(A[1][0], slot ^ (1 << 5));
                    lds_rd(A[1][1], (slot + 8192) ^ (1 << 5));
                }
#pragma unroll
                for (int ks = 0; ks < 8; ++ks) {
                    u32x4(&Ac)[2] = A[ks % 3];
                    if (dbg & 32) {
                        if (ks == 0) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Ac[0]), "+v"(Ac[1]));
                    } else if (ks + 2 < 8) {
                        lds_rd(A[(ks + 2) % 3][0], slot ^ ((ks + 2) << 5));
                        lds_rd(A[(ks + 2) % 3][1], (slot + 8192) ^ ((ks + 2) << 5));
                        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(Ac[0]), "+v"(Ac[1]));
                    } else if (ks + 1 < 8) {
                        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(Ac[0]), "+v"(Ac[1]));
                    } else {
                        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(Ac[0]), "+v"(Ac[1]));
                    }
                    if (ks == 0 && conv_here) { tie(kc); conv_mask(kc, first); }
                    if (ks == 0 && store_here) u_tie();
                    if (!(dbg & 2)) {
                        xacc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Ac[0]), bfr[kp * 8 + ks], xacc[0], 0, 0, 0);
                        xacc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Ac[1]), bfr[kp * 8 + ks], xacc[1], 0, 0, 0);
                    }
                    if (conv_here) conv_one(kc, ks);
                    if (store_here && ks == 1) u_store(s - 1, ust);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if (conv_here) conv_fin(kc, kp);
                ++g;
            });
            // ---- the stage's x rows: accumulators -> bf16 -> staging tile (s & 1)
            const unsigned sb = stg0 + (s & 1) * kStg;
            const unsigned hb_next = wave < kNW - 1 ? sb + 2 * kStg : carry_wr + s * 384;     // rows 0..2 of the next wave / the carry tile
#pragma unroll
            for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const u32x2 pk = {pack_bf2(xacc[cb][q * 4], xacc[cb][q * 4 + 1]), pack_bf2(xacc[cb][q * 4 + 2], xacc[cb][q * 4 + 3])};
                    lds_wr8(sb + xw_row + (((4 * cb + q) ^ xw_swz) << 4), pk);
                    if (jh >= 0) lds_wr8(hb_next + hw_row + (((4 * cb + q) ^ hw_swz) << 4), pk);
                }
            if (wave == 0 && lane < 24) {                            // rows 0..2 of the first wave: the carry of the previous tile / the pre-pass
                u32x4 v;
                lds_rd(v, carry_rd + s * 384 + lane * 16);
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v));
                lds_wr16(sb + lane * 16, v);
            }
        };
        stage(std::false_type{}, 0);
#pragma unroll 1
        for (int s = 1; s < n_stages; ++s) stage(std::true_type{}, s);

        // ---- tail: conv + x_proj of the last stage, then the x_dbl tile of this wave
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        {
            KStep kb[2];
            conv_reads(kb[0], n_stages - 1, 0);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                KStep &k = kb[ks & 1];
                if (ks + 1 < 4) {
                    conv_reads(kb[(ks + 1) & 1], n_stages - 1, ks + 1);
                    asm volatile("s_waitcnt lgkmcnt(12)" ::: "memory");
                } else {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                tie(k);
                conv_mask(k, first);
#pragma unroll
                for (int o = 0; o < 8; ++o) conv_one(k, o);
                conv_fin(k, ks);
            }
            u_xpose(n_stages - 1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            u_tie();
            u_store(n_stages - 1, ust);
        }
        // D[n][position]: lane = position j, outputs n = nb * 32 + (r & 3) + 8 (r >> 2) + 4 kh; through LDS (32 rows x 208 B in the
        // wave's own staging tiles) and out as 16-byte row pieces
        {
            constexpr int kPitch = 208;
#pragma unroll
            for (int nb = 0; nb < 3; ++nb)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const u32x2 pk = {pack_bf2(acc[nb][q * 4], acc[nb][q * 4 + 1]), pack_bf2(acc[nb][q * 4 + 2], acc[nb][q * 4 + 3])};
                    lds_wr8(stg0 + j * kPitch + nb * 64 + q * 16 + kh * 8, pk);
                }
            const int pc = lane & 15, r4 = lane >> 4;                // 4 rows x 16 pieces per instruction
            const int64_t m0 = seg0 + tile * kTile + wave * kTok;
            unsigned char *ob = reinterpret_cast<unsigned char *>(p.out) + (m0 + r4) * p.out_row_stride * 2 + pc * 16;
            u32x4 t[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) lds_rd(t[i], stg0 + (i * 4 + r4) * kPitch + pc * 16);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]));
            if (pc * 8 < n_out) {
#pragma unroll
                for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4 *>(ob + static_cast<int64_t>(i * 4) * p.out_row_stride * 2) = t[i];
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // (the surplus ring pieces: nothing may land in LDS after the workgroup is gone)
}

}  // namespace icx
}  // namespace zigma

using namespace zigma;

namespace {
// tiles per workgroup: the largest divisor of seqlen / 128 that still leaves >= 256 workgroups (one per CU); 1 otherwise
int icx_tiles(const zigma_in_conv_xproj_params_t &p) {
    const int per_seq = p.seqlen / icx::kTile;
    const int64_t n_tiles = static_cast<int64_t>(p.batch) * per_seq;
    int best = 1;
    for (int t = 1; t <= per_seq && t <= 64; ++t)
        if (per_seq % t == 0 && n_tiles / t >= 256) best = t;
    return best;
}
int icx_check(const zigma_in_conv_xproj_params_t &p) {
    if (p.batch < 0 || p.seqlen < 0 || p.dim < 1 || p.n < 1 || p.k < 1) return ZIGMA_ERR_SHAPE;
    if (p.flags & ~62) return ZIGMA_ERR_UNSUPPORTED;
    if (p.dtype != ZIGMA_BF16) return ZIGMA_ERR_DTYPE;
    if (p.n > 96 || p.n % 8 != 0 || p.dim % icx::kBC != 0 || p.dim > icx::kMaxStages * icx::kBC || p.seqlen % icx::kTile != 0) return ZIGMA_ERR_SHAPE;
    if (p.k != 640 && p.k != 768) return ZIGMA_ERR_SHAPE;         // (instantiated k: the README / CelebA / FacesHQ models and the UCF101 one)
    return ZIGMA_OK;
}
}  // namespace

extern "C" int64_t zigma_in_conv_x_proj_fwd_workspace_bytes(const zigma_in_conv_xproj_params_t *pp) {
    if (!pp || icx_check(*pp) != ZIGMA_OK || pp->batch == 0 || pp->seqlen == 0) return 0;
    const int64_t n_seg = static_cast<int64_t>(pp->batch) * (pp->seqlen / icx::kTile) / icx_tiles(*pp);
    return n_seg * 3 * pp->dim * 2;
}

extern "C" int zigma_in_conv_x_proj_fwd(const zigma_in_conv_xproj_params_t *pp, void *stream_) {
    if (!pp) return ZIGMA_ERR_NULL;
    (void)hipGetLastError();
    const zigma_in_conv_xproj_params_t &p = *pp;
    const int rc = icx_check(p);
    if (rc != ZIGMA_OK) return rc;
    if (p.batch == 0 || p.seqlen == 0) return ZIGMA_OK;
    if (!p.h || !p.w_in || !p.conv_weight || !p.conv_bias || !p.w || !p.u || !p.out || !p.workspace) return ZIGMA_ERR_NULL;
    if (p.workspace_bytes < zigma_in_conv_x_proj_fwd_workspace_bytes(pp)) return ZIGMA_ERR_SHAPE;
    auto al16 = [](const void *q) { return reinterpret_cast<uintptr_t>(q) % 16 == 0; };
    if (p.h_l_stride % 8 != 0 || p.h_batch_stride % 8 != 0 || p.u_l_stride % 8 != 0 || p.u_batch_stride % 8 != 0 || p.w_row_stride % 8 != 0 ||
        p.win_row_stride % 8 != 0 || p.out_row_stride % 8 != 0 || !al16(p.h) || !al16(p.u) || !al16(p.w) || !al16(p.w_in) || !al16(p.conv_weight) ||
        !al16(p.conv_bias) || !al16(p.out) || !al16(p.workspace))
        return ZIGMA_ERR_STRIDE;
    if (static_cast<int64_t>(p.dim) * p.win_row_stride * 2 > 0x7fffffff) return ZIGMA_ERR_STRIDE;      // (32-bit offsets inside W_in)
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int tiles = icx_tiles(p);
    const int64_t n_seg = static_cast<int64_t>(p.batch) * (p.seqlen / icx::kTile) / tiles;
    if (n_seg > 0x7fffffff) return ZIGMA_ERR_SHAPE;
    const dim3 hgrid(static_cast<unsigned>((n_seg + 3) / 4), static_cast<unsigned>(p.dim / 64)), grid(static_cast<unsigned>(n_seg)), block(256);
#define ZIGMA_ICX(K16_, PR_) hipLaunchKernelGGL((icx::in_conv_x_proj_kernel<K16_, PR_>), grid, block, 0, stream, p, tiles)
    if (p.k == 640) {
        hipLaunchKernelGGL((icx::in_halo_rows_kernel<20>), hgrid, block, 0, stream, p, tiles * icx::kTile, static_cast<int>(n_seg));
        switch (p.flags) {
            case 0: ZIGMA_ICX(40, 0); break;
            case 2: ZIGMA_ICX(40, 2); break;
            case 4: ZIGMA_ICX(40, 4); break;
            case 8: ZIGMA_ICX(40, 8); break;
            case 12: ZIGMA_ICX(40, 12); break;
            case 16: ZIGMA_ICX(40, 16); break;
            case 30: ZIGMA_ICX(40, 30); break;
            case 32: ZIGMA_ICX(40, 32); break;
            case 62: ZIGMA_ICX(40, 62); break;
            default: return ZIGMA_ERR_UNSUPPORTED;
        }
    } else {
        if (p.flags) return ZIGMA_ERR_UNSUPPORTED;
        hipLaunchKernelGGL((icx::in_halo_rows_kernel<24>), hgrid, block, 0, stream, p, tiles * icx::kTile, static_cast<int>(n_seg));
        ZIGMA_ICX(48, 0);
    }
#undef ZIGMA_ICX
    set_last_kernel("in_conv_x_proj_mfma");
    return check_launch();
}
