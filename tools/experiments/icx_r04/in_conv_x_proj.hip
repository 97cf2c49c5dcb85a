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
// bf16 only; width 4; bias required; seqlen % 128 == 0; k % 128 == 0, k <= 768; d_inner % 64 == 0, <= 1536; n <= 80, n % 8 == 0.
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
// direct-to-LDS load, 16 bytes per lane: LDS address = lds (wave-uniform) + 16 lane, source = base (wave-uniform) + voff (per lane, 32 bit).
// Inline assembly for the SGPR-base form: through the builtin every request carries a 64-bit per-lane pointer (2 VGPRs + a v_lshl_add_u64).
// (M0 is reserved: hipcc sets it right in front of each of its own uses, nothing lives in it across this statement.)
__device__ __forceinline__ void glds16(const void *base, unsigned voff, unsigned lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds) : "memory");
}
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
// A [3 n_seg rows] x [d_inner] x [k] product.  grid (ceil(3 n_seg / 128), d_inner / 64): a workgroup stages ONE 64-channel slab of W_in
// (64 x k bf16 = 80 KB, the ring-piece layout of the main kernel) and its four waves take 32 rows each as B fragments straight from
// global memory: x^T[64 ch][32 rows] = W_in . h^T, 2 x k/16 MFMAs per wave.  (The first version gave every 16 rows x 16 channels their
// own fragment-shaped W loads: 130 MB of L2 traffic in 64-byte pieces, 21 us — more than a tenth of the main kernel.)
template <int KSUB>
__global__ __launch_bounds__(256) void in_halo_rows_kernel(const zigma_in_conv_xproj_params_t p, const int seg_len, const int n_seg) {
    constexpr int NP = KSUB / 8;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NP * kPiece];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int j = lane & 31, kh = lane >> 5;
    const int ch0 = blockIdx.y * kBC;
    const int64_t win_rs2 = p.win_row_stride * 2;
    {   // the slab: piece kp = k columns 128 kp .. 128 kp + 127; rows 16 wave .. + 15, 4 rows per instruction, slot swizzle by the row
        const unsigned char *win = reinterpret_cast<const unsigned char *>(p.w_in) + static_cast<int64_t>(ch0) * win_rs2;
#pragma unroll
        for (int kp = 0; kp < NP; ++kp)
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = wave * 16 + 4 * i + (lane >> 4);
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)(win + row * win_rs2 + kp * 256 + (((lane & 15) ^ (row & 15)) << 4)),
                                                 (lds_ptr_t)(smem) + kp * kPiece + wave * 4096 + i * 1024, 16, 0, 0);
            }
    }
    const int rid = (blockIdx.x * 4 + wave) * 32 + j;               // row of the pre-pass: segment rid / 3, window row rid % 3
    const int seg = rid / 3, r = rid - seg * 3;
    bool valid = seg < n_seg;
    const int64_t pos_flat = static_cast<int64_t>(valid ? seg : 0) * seg_len;
    const int b = static_cast<int>(pos_flat / p.seqlen), t0 = static_cast<int>(pos_flat - static_cast<int64_t>(b) * p.seqlen);
    valid = valid && t0 >= kHalo;
    const int pos = valid ? t0 - kHalo + r : 0;
    const int row = p.x_row_index ? p.x_row_index[pos] : pos;
    const unsigned char *hs = reinterpret_cast<const unsigned char *>(p.h) + (static_cast<int64_t>(b) * p.h_batch_stride + static_cast<int64_t>(row) * p.h_l_stride) * 2 + kh * 16;
    bf16x8 bfr[KSUB];
#pragma unroll
    for (int kk = 0; kk < KSUB; ++kk) bfr[kk] = *reinterpret_cast<const bf16x8 *>(hs + kk * 32);
    __syncthreads();                                                 // (hipcc drains the LDS-DMA and the fragment loads in front of it)
    f32x16 acc[2] = {f32x16{}, f32x16{}};
    const unsigned a_off = j * 256 + ((kh ^ (j & 15)) << 4);
#pragma unroll
    for (int kk = 0; kk < KSUB; ++kk) {
        const unsigned off = (kk >> 3) * kPiece + (a_off ^ ((kk & 7) << 5));
        const bf16x8 a0 = *reinterpret_cast<const bf16x8 *>(smem + off), a1 = *reinterpret_cast<const bf16x8 *>(smem + off + 8192);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, bfr[kk], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, bfr[kk], acc[1], 0, 0, 0);
    }
    if (valid) {      // D[channel 32 cb + 8 q + 4 kh + 0..3][row j]
        unsigned char *dst = reinterpret_cast<unsigned char *>(p.workspace) + (static_cast<int64_t>(rid) * p.dim + ch0 + kh * 4) * 2;
#pragma unroll
        for (int cb = 0; cb < 2; ++cb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const u32x2 pk = {pack_bf2(acc[cb][q * 4], acc[cb][q * 4 + 1]), pack_bf2(acc[cb][q * 4 + 2], acc[cb][q * 4 + 3])};
                *reinterpret_cast<u32x2 *>(dst + (cb * 32 + q * 8) * 2) = pk;
            }
    }
}

template <int... I, typename F> __device__ __forceinline__ void static_for(std::integer_sequence<int, I...>, F &&f) {
    (f(std::integral_constant<int, I>{}), ...);
}
// B fragments of a token row: base (wave-uniform) + voff (per lane) + an immediate, into AGPRs (AG) or VGPRs
template <int OFF, bool AG> __device__ __forceinline__ void ld_frag(bf16x8 &d, unsigned voff, const void *base) {
    if constexpr (AG) asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=&a"(d) : "v"(voff), "s"(base), "n"(OFF));
    else asm volatile("global_load_dwordx4 %0, %1, %2 offset:%3" : "=&v"(d) : "v"(voff), "s"(base), "n"(OFF));
}
template <int K0, bool AG, int N, int... I>
__device__ __forceinline__ void ld_frags(bf16x8 (&arr)[N], unsigned voff, const void *base, std::integer_sequence<int, I...>) {
    (ld_frag<(K0 + I) * 32, AG>(arr[I], voff, base), ...);
}
template <int N> __device__ __forceinline__ void wait_vm() {
    static_assert(N >= 0 && N < 64, "");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// ---- the main kernel ---------------------------------------------------------------------------------------------------------------
// 512 threads = 8 waves = TWO per SIMD with different jobs (a workgroup's waves go round the four SIMDs, so waves w and w + 4 share one):
//   waves 0..3 ("product"): the W_in ring, the fragment reads, the 80 MFMAs of a stage, the bf16 x rows into the staging tiles;
//   waves 4..7 ("conv"):    conv + SiLU + x_proj of the PREVIOUS stage from those tiles, the u and x_dbl stores.
// The matrix pipe and the VALU of a SIMD then work side by side (PMC of the one-wave form, profiles/r04_a_*: 46 k instructions per
// wave at ~4.5 cycles of issue each = half the kernel, the matrix pipe 28 % busy — at one wave per SIMD every instruction is
// serial).  s_barrier is workgroup-wide, so both kinds of waves meet at every ring piece (5 per stage); the conv waves cut their
// stage into the same five parts.  PROBE bits (results wrong): 2 = no x product, 4 = no u stores, 8 = no conv arithmetic,
// 16 = no W_in stream, 32 = no fragment reads.
template <int KSUB, int PROBE>
__global__ __launch_bounds__(512) void in_conv_x_proj_kernel(const zigma_in_conv_xproj_params_t p, const int tiles) {
    constexpr int NP = KSUB / 8;                                    // ring pieces per stage
    constexpr int dbg = PROBE;
    static_assert(KSUB % 8 == 0 && NP >= 5, "conv chunks 0..3 and the u stores ride on pieces 0..4");
    __shared__ __attribute__((aligned(1024))) unsigned char smem[kLds];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave8 & 3;                                      // which 32 positions of the tile
    const bool conv_role = wave8 >= 4;
    const int j = lane & 31, kh = lane >> 5;
    const int L = p.seqlen, n_stages = p.dim / kBC, n_out = p.n;
    const int64_t seg0 = static_cast<int64_t>(blockIdx.x) * tiles * kTile;      // first position (all samples) of this workgroup
    const int b = static_cast<int>(seg0 / L), t_seg = static_cast<int>(seg0 - static_cast<int64_t>(b) * L);
    const unsigned smem_lds = static_cast<unsigned>(reinterpret_cast<uintptr_t>((lds_ptr_t)(smem)));
    const unsigned stg0 = smem_lds + kStgOff + wave * 2 * kStg;      // staging tiles (two parities) of this position group
    const int total_stages = tiles * n_stages;

    // ---- carry tile of the workgroup's first tile: the pre-pass rows, in the layout of staging rows 0..2 of every stage
    if (t_seg != 0) {
        for (int idx = tid; idx < n_stages * 24; idx += 512) {
            const int s = idx / 24, rem = idx - s * 24, r = rem >> 3, c = rem & 7;
            const u32x4 v = *reinterpret_cast<const u32x4 *>(reinterpret_cast<const unsigned char *>(p.workspace) +
                                                             ((static_cast<int64_t>(blockIdx.x) * 3 + r) * p.dim + s * kBC + c * 8) * 2);
            lds_wr16(smem_lds + kCarryOff + s * 384 + r * 128 + ((c ^ ((r >> 1) & 7)) << 4), v);
        }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");

    // probe bit 128: the waves of workgroup 1 write s_memtime stamps behind the pre-pass rows of the workspace ([wave][1024] x 8 bytes)
    int n_stamp = 0;
    auto stamp = [&]() {
        if (!(dbg & 128)) return;
        if (blockIdx.x == 1 && n_stamp < 1024) {
            uint64_t t;
            asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t)::"memory");
            if (lane == 0)
                reinterpret_cast<uint64_t *>(reinterpret_cast<unsigned char *>(p.workspace) + static_cast<int64_t>(gridDim.x) * 3 * p.dim * 2)[wave8 * 1024 + n_stamp] = t;
            ++n_stamp;
        }
    };
    // Meetings (s_barrier is workgroup-wide; both kinds of waves count the same ones): #0 at the start, then B_g = #(g + 1) inside piece
    // g, placed at its k-step 6: "piece g + 1 has landed for every product wave, and every wave has left piece g - 1 behind" — the
    // fragment reads of piece g + 1 start under the last two k-steps of piece g, and piece g + 3 is requested into the slot of g - 1.
    // Stage S's x rows are in the staging tiles (S & 1) before its product waves reach B_(5 S + 5) = #(5 S + 6); the conv waves start
    // stage S behind that meeting and finish it (u stores included) before #(5 S + 11): one stage behind, one meeting per chunk.
    if (!conv_role) {
        // ================================================= product waves ===========================================================
        const int *tab = p.x_row_index;
        const unsigned char *hb = reinterpret_cast<const unsigned char *>(p.h) + static_cast<int64_t>(b) * p.h_batch_stride * 2;
        const int h_ls2 = static_cast<int>(p.h_l_stride) * 2;
        const unsigned char *win = reinterpret_cast<const unsigned char *>(p.w_in);
        const int64_t win_rs2 = p.win_row_stride * 2;
        // W_in ring: wave w fetches rows 16 w .. 16 w + 15 of a piece, 4 rows per instruction; lane -> (row lane >> 4, PHYSICAL slot
        // lane & 15) holding the LOGICAL slot (lane & 15) ^ (row & 15): a fragment read (32 rows, one logical slot) then hits 16
        // different 16-byte slots per lane group = every bank once
        // (request i of a piece = rows + 4 i: row & 15 = 4 i | (lane >> 4), so its slot is the slot of request 0 ^ 4 i — one offset
        // register, an add and a xor per request instead of four registers that hipcc then spills inside the stage loop)
        const unsigned woff0 = static_cast<unsigned>((wave * 16 + (lane >> 4)) * win_rs2) + (((lane & 15) ^ (lane >> 4)) << 4);
        const unsigned wrow4 = static_cast<unsigned>(4 * win_rs2);
        int iss_s = 0, iss_kp = 0, iss_slot = 0;                     // (stage, piece, ring slot) of the next piece to request
        // a piece = 4 VM instructions per wave, requested in two halves (the issue of a direct-to-LDS load costs the wave ~60 cycles);
        // the run requests pieces beyond its end (nobody reads them) so that every wait is the same vmcnt(4)
        auto issue_half = [&](const int half) {
            if (dbg & 16) return;
            const unsigned char *src = win + static_cast<int64_t>(iss_s) * kBC * win_rs2 + iss_kp * 256;
            const unsigned dst = smem_lds + kRingOff + iss_slot * kPiece + wave * 4096;
#pragma unroll
            for (int i = 2 * half; i < 2 * half + 2; ++i) glds16(src, (woff0 + i * wrow4) ^ (i << 6), dst + i * 1024);
            if (half == 1) {
                iss_slot = (iss_slot + 1) & (kNR - 1);
                if (++iss_kp == NP) { iss_kp = 0; if (++iss_s == n_stages) iss_s = 0; }
            }
        };
        const unsigned a_off = smem_lds + kRingOff + j * 256 + ((kh ^ (j & 15)) << 4);     // fragment read: row j (+ 32 cb), logical slot 2 ks + kh
        // x rows out of the accumulators: token j -> row 3 + j; register group q of block cb = channels 32 cb + 8 q + 4 kh + 0..3
        // = logical slot 4 cb + q, byte 8 kh of the slot
        // (slot s of row r sits at r * 128 + ((s ^ ((r >> 1) & 7)) << 4) = (r * 128 | swizzle << 4) ^ (s << 4): one register per row)
        const unsigned xw_pre = ((kHalo + j) * 128 + kh * 8) | ((((kHalo + j) >> 1) & 7) << 4);
        const int jh = j - (kTok - kHalo);                           // >= 0: this token is part of the next wave's causal window
        const unsigned hw_pre = ((jh > 0 ? jh : 0) * 128 + kh * 8) | ((((jh > 0 ? jh : 0) >> 1) & 7) << 4);

        f32x16 xacc[2];
        // 2 waves per SIMD = 256 registers per lane, which hipcc splits 128 VGPRs + 128 AGPRs: the 160 registers of B fragments live
        // in both files (MFMA operands may come from either) — 24 fragments + the 32 accumulators fill the AGPRs, 16 fragments sit in VGPRs
        constexpr int NA = 24;
        static_assert(KSUB == 40 && kNR == 4, "register budget of the product waves: k = 640");
        bf16x8 bfa[NA], bfv[KSUB - NA];
        u32x4 A[4][2];                                               // fragment sets of k-steps t, t + 1, t + 2 (two ahead), by t & 3
        int cur = 0;                                                 // ring slot of the current piece
        int gs = 0;                                                  // stage index over the whole run (parity of the staging tiles)

        issue_half(0); issue_half(1);                                // pieces 0, 1
        issue_half(0); issue_half(1);
        wait_vm<4>();                                                // piece 0 has landed (this wave's part)
        __builtin_amdgcn_s_barrier();                                // #0
        issue_half(0); issue_half(1);                                // piece 2
        if (!(dbg & 32)) {
            lds_rd(A[0][0], a_off);
            lds_rd(A[0][1], a_off + 8192);
            lds_rd(A[1][0], a_off ^ (1 << 5));
            lds_rd(A[1][1], (a_off + 8192) ^ (1 << 5));
        }
#pragma unroll 1
        for (int tile = 0; tile < tiles; ++tile) {
            const int t_wave = t_seg + tile * kTile + wave * kTok;   // first position of this wave inside the sample
            {   // this wave's 32 token rows as B fragments: lane = (token j, k-half kh), 16 bytes per k-step of 16.  Straight into
                // the registers the MFMAs take them from, by inline assembly: beside loads it can see, hipcc drains every LDS-DMA
                // in flight before each use.
                const int pos = t_wave + j;
                const int row = tab ? tab[pos] : pos;
                const unsigned hoff = static_cast<unsigned>(row * h_ls2 + kh * 16);          // (rows of one sample: < 2^31 bytes, checked by the launcher)
                ld_frags<0, true>(bfa, hoff, hb, std::make_integer_sequence<int, NA>{});
                ld_frags<NA, false>(bfv, hoff, hb, std::make_integer_sequence<int, KSUB - NA>{});
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
                for (int kk = 0; kk < NA; kk += 8)
                    asm volatile("" : "+a"(bfa[kk]), "+a"(bfa[kk + 1]), "+a"(bfa[kk + 2]), "+a"(bfa[kk + 3]), "+a"(bfa[kk + 4]), "+a"(bfa[kk + 5]),
                                 "+a"(bfa[kk + 6]), "+a"(bfa[kk + 7]));
#pragma unroll
                for (int kk = 0; kk < KSUB - NA; kk += 8)
                    asm volatile("" : "+v"(bfv[kk]), "+v"(bfv[kk + 1]), "+v"(bfv[kk + 2]), "+v"(bfv[kk + 3]), "+v"(bfv[kk + 4]), "+v"(bfv[kk + 5]),
                                 "+v"(bfv[kk + 6]), "+v"(bfv[kk + 7]));
            }
            const unsigned carry_rd = smem_lds + kCarryOff + (tile & 1) * (kMaxStages * 384);
            const unsigned carry_wr = smem_lds + kCarryOff + ((tile + 1) & 1) * (kMaxStages * 384);
#pragma unroll 1
            for (int s = 0; s < n_stages; ++s, ++gs) {
                static_for(std::make_integer_sequence<int, NP>{}, [&](auto kp_tag) {
                    constexpr int kp = decltype(kp_tag)::value;
                    const unsigned slot = a_off + cur * kPiece, nslot = a_off + ((cur + 1) & (kNR - 1)) * kPiece;
#pragma unroll
                    for (int ks = 0; ks < 8; ++ks) {
                        if (ks == 6) {      // B_g: piece g + 1 is there for everybody (this wave: only piece g + 2 may still be in flight)
                            stamp();
                            wait_vm<4>();
                            stamp();
                            __builtin_amdgcn_s_barrier();
                            stamp();
                        }
                        u32x4(&Ac)[2] = A[ks & 3];
                        if (!(dbg & 32)) {  // two k-steps ahead (behind k-step 5: the next piece)
                            const unsigned ra = ks < 6 ? slot ^ ((ks + 2) << 5) : nslot ^ ((ks - 6) << 5);
                            lds_rd(A[(ks + 2) & 3][0], ra);
                            lds_rd(A[(ks + 2) & 3][1], ra + 8192);
                        }
                        // LDS returns in order: everything but the four youngest reads = this k-step's fragments (and, at a stage's
                        // first k-step, the staging writes in front of them)
                        asm volatile("s_waitcnt lgkmcnt(4)" : "+v"(Ac[0]), "+v"(Ac[1]));
                        if (!(dbg & 2)) {
                            // (inline assembly so that the B fragments are taken from the registers they were loaded into: through the
                            // builtin hipcc copies AGPR operands to VGPRs first, 4 v_accvgpr_read per k-step)
                            constexpr int kk = kp * 8;
                            if (kp == 0 && ks == 0) {
                                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(xacc[0]) : "v"(Ac[0]), "a"(bfa[0]));
                                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=a"(xacc[1]) : "v"(Ac[1]), "a"(bfa[0]));
                            } else if (kk < NA) {
                                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(xacc[0]) : "v"(Ac[0]), "a"(bfa[kk + ks]));
                                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(xacc[1]) : "v"(Ac[1]), "a"(bfa[kk + ks]));
                            } else {
                                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(xacc[0]) : "v"(Ac[0]), "v"(bfv[kk + ks - NA]));
                                asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(xacc[1]) : "v"(Ac[1]), "v"(bfv[kk + ks - NA]));
                            }
                        } else if (kp == 0 && ks == 0) {
                            xacc[0] = f32x16{};
                            xacc[1] = f32x16{};
                        }
                        if (ks == 6) issue_half(0);              // piece g + 3, behind the MFMAs of the k-step
                        if (ks == 7) issue_half(1);
                    }
                    cur = (cur + 1) & (kNR - 1);
                });
                // ---- the stage's x rows: accumulators -> bf16 -> staging tile (gs & 1).  (The MFMAs above are opaque to hipcc: the
                // wait states between the last one and the first read of its result are spelled out.)
                asm volatile("s_nop 15\n\ts_nop 7" : "+a"(xacc[0]), "+a"(xacc[1]));
                const unsigned sb = stg0 + (gs & 1) * kStg;
                const unsigned hb_next = wave < kNW - 1 ? sb + 2 * kStg : carry_wr + s * 384;     // rows 0..2 of the next wave / the carry tile
#pragma unroll
                for (int cb = 0; cb < 2; ++cb)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        const u32x2 pk = {pack_bf2(xacc[cb][q * 4], xacc[cb][q * 4 + 1]), pack_bf2(xacc[cb][q * 4 + 2], xacc[cb][q * 4 + 3])};
                        lds_wr8((sb + xw_pre) ^ ((4 * cb + q) << 4), pk);
                        if (jh >= 0) lds_wr8((hb_next + hw_pre) ^ ((4 * cb + q) << 4), pk);
                    }
                if (wave == 0 && lane < 24) {                        // rows 0..2 of the first wave: the carry of the previous tile / the pre-pass
                    u32x4 v;
                    lds_rd(v, carry_rd + s * 384 + lane * 16);
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v));
                    lds_wr16(sb + lane * 16, v);
                }
            }
        }
        // the conv waves are one stage behind: NP + 1 more meetings (the first one also publishes the last staging tile)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(A[0][0]), "+v"(A[0][1]), "+v"(A[1][0]), "+v"(A[1][1]));
#pragma unroll
        for (int kp = 0; kp < NP + 1; ++kp) __builtin_amdgcn_s_barrier();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");             // (the surplus ring pieces: nothing may land in LDS after the workgroup is gone)
        return;
    }

    // ===================================================== conv waves ==============================================================
    if (dbg & 64) __builtin_amdgcn_s_setprio(3);
    // W_x slab + conv taps / bias of a stage (as in conv_x_proj.hip): 8 rows per instruction, instruction q = wave + 4 i
    unsigned wsoff[3];
    int n_w = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        int row = (wave + kNW * i) * 8 + (lane >> 3);
        row = row < n_out ? row : n_out - 1;
        wsoff[i] = static_cast<unsigned>(row * static_cast<int>(p.w_row_stride) * 2) + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
        n_w += (wave + kNW * i) * 8 < n_out ? 1 : 0;
    }
    // taps (lanes 0..31: 2 channels each) and bias (lanes 32..39: 8 channels each; the rest repeat lane 39) in ONE request: the bias
    // lanes address conv_bias relative to conv_weight (both inside the caller's parameter arena: the distance fits 32 bits, checked)
    const unsigned cb_rel = static_cast<unsigned>(reinterpret_cast<const unsigned char *>(p.conv_bias) - reinterpret_cast<const unsigned char *>(p.conv_weight));
    auto issue_xw = [&](const int st, const int par) {              // slab of stage st into the buffer of parity par
        const unsigned dst = smem_lds + kXwOff + par * kXw;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if (i < n_w) glds16(reinterpret_cast<const unsigned char *>(p.w) + st * (kBC * 2), wsoff[i], dst + (wave + kNW * i) * 1024);
        if (wave == 1) {
            const unsigned coff = lane < 32 ? static_cast<unsigned>(st * (kBC * 8) + lane * 16)
                                            : cb_rel + static_cast<unsigned>(st * (kBC * 2) + ((lane < 40 ? lane : 39) - 32) * 16);
            glds16(p.conv_weight, coff, dst + 96 * 128);
        }
    };
    unsigned x_off[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) x_off[s] = (j + s) * 128 + ((kh ^ (((j + s) >> 1) & 7)) << 4);
    const unsigned w_off = j * 128 + ((kh ^ ((j >> 1) & 7)) << 4);
    const unsigned u_wr = j * 128 + ((kh ^ ((j >> 1) & 7)) << 4);
    const unsigned u_rd = (lane >> 3) * 128 + (((lane & 7) ^ (lane >> 4)) << 4);

    u32x4 uq[4];
    f32x16 acc[3];
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) acc[nb] = f32x16{};
    KStep k;
    // One chunk = 8 channels of 32 positions: 12 LDS reads (4 x rows, taps, bias, 3 W_x fragments), ~100 VALU, 3 MFMAs.  The reads of
    // the NEXT chunk go into the same registers as soon as their last reader is through (the staging tile and the slab do not
    // change during a stage), so that only a stage's first chunk waits for the LDS round trip.
    auto rd_x = [&](const int par, const int ks) {
        const unsigned sb = stg0 + par * kStg;
#pragma unroll
        for (int s = 0; s < 4; ++s) lds_rd(k.X[s], (sb + x_off[s]) ^ (ks << 5));
    };
    auto rd_wc = [&](const int par, const int ks, const int r) { lds_rd(k.Wc[r], smem_lds + kXwOff + par * kXw + 96 * 128 + (ks * 2 + kh) * 64 + r * 16); };
    auto rd_bc = [&](const int par, const int ks) { lds_rd(k.Bc, smem_lds + kXwOff + par * kXw + 96 * 128 + 512 + (ks * 2 + kh) * 16); };
    auto rd_bf = [&](const int par, const int ks, const int nb) { lds_rd(k.Bf[nb], (smem_lds + kXwOff + par * kXw + w_off + nb * 32 * 128) ^ (ks << 5)); };
    auto chunk = [&](const int par, const int ks, const bool first, const bool next) {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        tie(k);
        if (first) {
#pragma unroll
            for (int s = 0; s < 3; ++s)
                if (j + s < kHalo) k.X[s] = u32x4{0, 0, 0, 0};
        }
        unsigned pm[4][4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {                                // channels 2r (low halves) and 2r + 1 (high halves)
            pm[r][0] = __builtin_amdgcn_perm(k.X[1][r], k.X[0][r], 0x05040100u);
            pm[r][1] = __builtin_amdgcn_perm(k.X[3][r], k.X[2][r], 0x05040100u);
            pm[r][2] = __builtin_amdgcn_perm(k.X[1][r], k.X[0][r], 0x07060302u);
            pm[r][3] = __builtin_amdgcn_perm(k.X[3][r], k.X[2][r], 0x07060302u);
        }
        unsigned ur[4] = {k.X[3].x, k.X[3].y, k.X[3].z, k.X[3].w};
        asm volatile("" : "+v"(pm[0][0]), "+v"(pm[0][1]), "+v"(pm[0][2]), "+v"(pm[0][3]), "+v"(pm[1][0]), "+v"(pm[1][1]), "+v"(pm[1][2]), "+v"(pm[1][3]),
                     "+v"(pm[2][0]), "+v"(pm[2][1]), "+v"(pm[2][2]), "+v"(pm[2][3]), "+v"(pm[3][0]), "+v"(pm[3][1]), "+v"(pm[3][2]), "+v"(pm[3][3]),
                     "+v"(ur[0]), "+v"(ur[1]), "+v"(ur[2]), "+v"(ur[3]));      // (the x rows are consumed: their registers may be refilled)
        if (next) rd_x(par, ks + 1);
        float bias[8];
#pragma unroll
        for (int r = 0; r < 4; ++r) { bias[2 * r] = bf_lo(k.Bc[r]); bias[2 * r + 1] = bf_hi(k.Bc[r]); }
        asm volatile("" : "+v"(bias[0]), "+v"(bias[1]), "+v"(bias[2]), "+v"(bias[3]), "+v"(bias[4]), "+v"(bias[5]), "+v"(bias[6]), "+v"(bias[7]));
        if (next) rd_bc(par, ks + 1);
        // all eight channels side by side (one wave has to cover its own VALU latencies: per-channel chains issue at ~10 cycles per
        // instruction, measured with s_memtime stamps): the 16 dot products first — they consume the taps, whose registers are then
        // refilled —, then the eight SiLUs as independent streams
        if (!(dbg & 8)) {
            float av[8];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                av[2 * r] = dot2(pm[r][0], k.Wc[r].x, bias[2 * r]);
                av[2 * r + 1] = dot2(pm[r][2], k.Wc[r].z, bias[2 * r + 1]);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                av[2 * r] = dot2(pm[r][1], k.Wc[r].y, av[2 * r]);
                av[2 * r + 1] = dot2(pm[r][3], k.Wc[r].w, av[2 * r + 1]);
            }
            asm volatile("" : "+v"(av[0]), "+v"(av[1]), "+v"(av[2]), "+v"(av[3]), "+v"(av[4]), "+v"(av[5]), "+v"(av[6]), "+v"(av[7]));      // (taps consumed)
            if (next) {
#pragma unroll
                for (int r = 0; r < 4; ++r) rd_wc(par, ks + 1, r);
            }
            float e[8];
#pragma unroll
            for (int c = 0; c < 8; ++c) e[c] = fast_exp2(av[c] * -kLog2e);
#pragma unroll
            for (int c = 0; c < 8; ++c) e[c] = fast_rcp(1.f + e[c]);
#pragma unroll
            for (int r = 0; r < 4; ++r) ur[r] = pack_bf2(av[2 * r] * e[2 * r], av[2 * r + 1] * e[2 * r + 1]);
        } else if (next) {
#pragma unroll
            for (int r = 0; r < 4; ++r) rd_wc(par, ks + 1, r);
        }
        const u32x4 u8 = {ur[0], ur[1], ur[2], ur[3]};
        uq[ks] = u8;
        // the 8 outputs ARE the B fragment of the x_proj product.  Inline assembly keeps the 48 accumulators in AGPRs: through the
        // builtin hipcc moves all of them to VGPRs and back around every chunk (96 v_accvgpr moves per ~110 useful instructions).
        // (s_nop: the wait states between the VALU that wrote u8 and the MFMA that reads it — opaque to hipcc inside the asm.)
        if (!(dbg & 256)) {
            asm volatile("s_nop 4\n\tv_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[0]) : "v"(k.Bf[0]), "v"(u8));
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[1]) : "v"(k.Bf[1]), "v"(u8));
            asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc[2]) : "v"(k.Bf[2]), "v"(u8));
        }
        if (next) {
            asm volatile("s_nop 7" ::: "memory");                   // (the MFMAs have read their W_x fragments)
#pragma unroll
            for (int nb = 0; nb < 3; ++nb) rd_bf(par, ks + 1, nb);
        }
    };
    // u of a stage (32 positions x 64 channels) leaves as full 128-byte lines, transposed through the wave's own (consumed) staging tile
    auto u_store = [&](const int par, unsigned char *dst) {
        if (dbg & 4) return;
        const unsigned sb = stg0 + par * kStg;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) lds_wr16((sb + u_wr) ^ (ks << 5), uq[ks]);
        u32x4 t[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) lds_rd(t[i], (sb + u_rd + i * 1024) ^ ((i & 1) << 6));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]));
#pragma unroll
        for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4 *>(dst + static_cast<int64_t>(i * 8) * (p.u_l_stride * 2)) = t[i];
    };
    // x_dbl tile of a finished tile: D[n][position], lane = position j, outputs n = nb * 32 + (r & 3) + 8 (r >> 2) + 4 kh; through LDS
    // (32 rows x 160 B = one staging tile: n <= 80) and out as 16-byte row pieces
    auto xdbl_store = [&](const int par, const int64_t m0) {
        constexpr int kPitch = 160;
        const unsigned sb = stg0 + par * kStg;
        asm volatile("s_nop 15\n\ts_nop 7" : "+a"(acc[0]), "+a"(acc[1]), "+a"(acc[2]));
#pragma unroll
        for (int nb = 0; nb < 3; ++nb)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const u32x2 pk = {pack_bf2(acc[nb][q * 4], acc[nb][q * 4 + 1]), pack_bf2(acc[nb][q * 4 + 2], acc[nb][q * 4 + 3])};
                if (nb * 32 + q * 8 + kh * 4 < n_out) lds_wr8(sb + j * kPitch + nb * 64 + q * 16 + kh * 8, pk);
            }
        const int pc = lane & 15, r4 = lane >> 4;                    // 4 rows x 16 pieces per instruction (pieces 0..9 exist)
        unsigned char *ob = reinterpret_cast<unsigned char *>(p.out) + (m0 + r4) * p.out_row_stride * 2 + pc * 16;
        u32x4 t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) lds_rd(t[i], sb + (i * 4 + r4) * kPitch + (pc < 10 ? pc : 9) * 16);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]));
        if (pc * 8 < n_out) {
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4 *>(ob + static_cast<int64_t>(i * 4) * p.out_row_stride * 2) = t[i];
        }
#pragma unroll
        for (int nb = 0; nb < 3; ++nb) acc[nb] = f32x16{};
    };

    issue_xw(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < NP + 2; ++i) { stamp(); __builtin_amdgcn_s_barrier(); }   // #0 .. #6: stage 0 is in its staging tiles, its slab in the buffer
#pragma unroll 1
    for (int S = 0; S < total_stages; ++S) {
        const int tp = S / n_stages, sp = S - tp * n_stages, par = S & 1;
        const int t_wave = t_seg + tp * kTile + wave * kTok;         // first position of this wave inside the sample, tile tp
        const bool first = t_wave == 0;                              // wave-uniform: the causal window starts inside this wave's rows
        if (S + 1 < total_stages) issue_xw((S + 1) % n_stages, (S + 1) & 1);
        rd_x(par, 0);
#pragma unroll
        for (int r = 0; r < 4; ++r) rd_wc(par, 0, r);
        rd_bc(par, 0);
#pragma unroll
        for (int nb = 0; nb < 3; ++nb) rd_bf(par, 0, nb);
        stamp();
        chunk(par, 0, first, true);
        stamp();
        __builtin_amdgcn_s_barrier();
        stamp();
        chunk(par, 1, first, true);
        stamp();
        __builtin_amdgcn_s_barrier();
        stamp();
        chunk(par, 2, first, true);
        stamp();
        __builtin_amdgcn_s_barrier();
        stamp();
        chunk(par, 3, first, false);
        stamp();
        __builtin_amdgcn_s_barrier();
        stamp();
        unsigned char *dst = reinterpret_cast<unsigned char *>(p.u) + static_cast<int64_t>(b) * p.u_batch_stride * 2 +
                             static_cast<int64_t>(t_wave + (lane >> 3)) * (p.u_l_stride * 2) + (lane & 7) * 16 + sp * (kBC * 2);
        u_store(par, dst);
        if (sp == n_stages - 1) xdbl_store(par, seg0 + tp * kTile + wave * kTok);
        // the next stage's slab: requested at the top of this one; behind it only this stage's u (and x_dbl) stores
        stamp();
        if (dbg & 4) wait_vm<0>(); else wait_vm<4>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        stamp();
        __builtin_amdgcn_s_barrier();
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
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
    if (p.flags & ~510) return ZIGMA_ERR_UNSUPPORTED;
    if (p.dtype != ZIGMA_BF16) return ZIGMA_ERR_DTYPE;
    if (p.n > 80 || p.n % 8 != 0 || p.dim % icx::kBC != 0 || p.dim > icx::kMaxStages * icx::kBC || p.seqlen % icx::kTile != 0) return ZIGMA_ERR_SHAPE;
    if (p.k != 640) return ZIGMA_ERR_SHAPE;                       // (the product waves keep a k = 640 row of their token in 160 registers)
    return ZIGMA_OK;
}
}  // namespace

extern "C" int64_t zigma_in_conv_x_proj_fwd_workspace_bytes(const zigma_in_conv_xproj_params_t *pp) {
    if (!pp || icx_check(*pp) != ZIGMA_OK || pp->batch == 0 || pp->seqlen == 0) return 0;
    const int64_t n_seg = static_cast<int64_t>(pp->batch) * (pp->seqlen / icx::kTile) / icx_tiles(*pp);
    return n_seg * 3 * pp->dim * 2 + 8 * 1024 * 8;       // (+ the stamp area of probe bit 128)
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
    const dim3 hgrid(static_cast<unsigned>((3 * n_seg + 127) / 128), static_cast<unsigned>(p.dim / 64)), grid(static_cast<unsigned>(n_seg)), hblock(256), block(512);
#define ZIGMA_ICX(K16_, PR_) hipLaunchKernelGGL((icx::in_conv_x_proj_kernel<K16_, PR_>), grid, block, 0, stream, p, tiles)
    hipLaunchKernelGGL((icx::in_halo_rows_kernel<40>), hgrid, hblock, 0, stream, p, tiles * icx::kTile, static_cast<int>(n_seg));
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
        case 64: ZIGMA_ICX(40, 64); break;
        case 128: ZIGMA_ICX(40, 128); break;
        case 192: ZIGMA_ICX(40, 192); break;
        case 256: ZIGMA_ICX(40, 256); break;
        case 258: ZIGMA_ICX(40, 258); break;
        default: return ZIGMA_ERR_UNSUPPORTED;
    }
#undef ZIGMA_ICX
    set_last_kernel("in_conv_x_proj_mfma");
    return check_launch();
}
