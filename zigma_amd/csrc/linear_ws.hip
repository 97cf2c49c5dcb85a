// linear_ws: out = x @ W^T for SHORT k (in_proj, to_q: k = 640) with the WEIGHTS STATIONARY IN REGISTERS, gfx950.
// Reference call sites: mamba_simple.py:290-294 (in_proj), model_zigma.py:104-110 (to_q), both F.linear.
//
// Why a second GEMM form: with 256 x 256 tiles (linear4w) every k-step brings 32 KB of activations AND 32 KB of weights from L2 into
// LDS, and that stream (~8 TB/s effective over the chip) — not the matrix pipe — bounds all projections of the block at 40-45 % of the
// MFMA peak.  For k <= 640 a 256-feature panel of W is 320 KB = 320 registers per lane of four waves: it fits the 512-register file
// of a one-wave-per-SIMD workgroup next to the accumulators.  Then only the tokens stream (half the bytes per flop), each activation
// slice is read by the `panels` workgroups of its range at the same time (one L2 fill), and the weights are read once per workgroup.
//
//   workgroup = 4 waves, one per SIMD, one per CU, persistent: (panel of 256 features, range of 64-token tiles inside its XCD's
//   eighth of the tokens).  Wave w keeps W rows [256 panel + 64 w, + 64) x k as MFMA A fragments (2 blocks of 32 features x k / 16
//   fragments of 4 registers; 128 in AGPRs, the rest in VGPRs), accumulates D[64 features][64 tokens] in 64 AGPRs — two sets,
//   the epilogue of a tile rides in the MFMA gaps of the next one.
//   Tokens: slices of 64 tokens x 128 k (16 KB, global_load_lds_dwordx4, 16-byte slots XOR-swizzled by the row on the source side)
//   through a ring of eight, seven slices ahead, ONE counted vmcnt + barrier per slice, placed one k-group before the slice's end so
//   that the first fragments of the next slice are in flight when it starts; B fragments by ds_read_b128, one k-group ahead.
//   Epilogue per tile: accumulators -> bf16 -> the wave's 4 KB LDS tile -> 16-byte stores, 128 contiguous bytes per token.
// Limits: bf16, no bias / activation / residual, k % 128 == 0, k <= 640, n % 256 == 0, n <= 8192, m % 512 == 0.
#include "zigma_common.h"

#include <utility>

namespace zigma {
namespace lws {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) unsigned char *lds_ptr_t;

#ifdef ZIGMA_WS_NO_NT
constexpr bool kWsDefaultPolicyStores = true;     // (A/B build of tools/fwd_nt_ab.sh)
#else
constexpr bool kWsDefaultPolicyStores = false;
#endif
constexpr int kT = 64;                    // tokens per tile
constexpr int kSlice = kT * 256;          // ring slot: 64 tokens x 128 k, bf16
constexpr int kRing = 8;
constexpr int kScrOff = kRing * kSlice;   // 131072: 4 waves x 8 KB of epilogue tiles (two token blocks)
constexpr int kLds = kScrOff + 4 * 8192;  // 163840

__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}
// every LDS access and every direct-to-LDS load is inline assembly: hipcc must not see them (it would drain vmcnt before each read)
template <int OFF>
__device__ __forceinline__ void lds_rd(u32x4 &d, unsigned addr) { asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF)); }
__device__ __forceinline__ void lds_wr8(unsigned addr, const u32x2 &v) { asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v) : "memory"); }
__device__ __forceinline__ void glds16(const void *base, unsigned voff, unsigned lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds) : "memory");
}
__device__ __forceinline__ void ld_w_a(u32x4 &d, const void *ptr, const int off) { asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=a"(d) : "v"(ptr), "n"(off)); }
__device__ __forceinline__ void ld_w_v(u32x4 &d, const void *ptr, const int off) { asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(d) : "v"(ptr), "n"(off)); }
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void barrier() { asm volatile("s_barrier" ::: "memory"); }

// D += W . T^T for one 32 x 32 block; FIRST: the accumulator starts at zero (inline constant as srcC)
template <bool FIRST>
__device__ __forceinline__ void mfma_wa(f32x16 &acc, const u32x4 &w, const u32x4 &b) {           // W fragment in AGPRs
    if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&a"(acc) : "a"(w), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "a"(w), "v"(b));
}
template <bool FIRST>
__device__ __forceinline__ void mfma_wv(f32x16 &acc, const u32x4 &w, const u32x4 &b) {           // W fragment in VGPRs
    if constexpr (FIRST) asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, 0" : "=&a"(acc) : "v"(w), "v"(b));
    else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(acc) : "v"(w), "v"(b));
}

template <int KG>
struct WFrags {
    static constexpr int NF = 2 * KG, NA = NF < 32 ? NF : 32, NV = NF - NA;
    u32x4 a[NA];
    u32x4 v[NV > 0 ? NV : 1];
};

// fragment f = 2 kg + fb (f and first fold to constants in the unrolled loop: one instruction survives)
template <int KG>
__device__ __forceinline__ void mfma_f(f32x16 &acc, const WFrags<KG> &w, const int f, const bool first, const u32x4 &b) {
    if (f < WFrags<KG>::NA) { if (first) mfma_wa<true>(acc, w.a[f], b); else mfma_wa<false>(acc, w.a[f], b); }
    else { if (first) mfma_wv<true>(acc, w.v[f - WFrags<KG>::NA], b); else mfma_wv<false>(acc, w.v[f - WFrags<KG>::NA], b); }
}

// PROBE (tools/linear_ws_probe.py; probe builds only): 0 = the kernel, 1 = no epilogue, 2 = default-policy stores instead of nt,
// 3 = MFMAs only (no stream after the prologue, no barriers, no fragment reads, no epilogue), 4 = sc0 sc1 (write-through) stores, 5 = no fragment reads (wrong results), 6 = only the MFMAs of feature block 0 (one MFMA per fragment read: the loop rate of a 128-feature panel; wrong results), 7 = one barrier per slice instead of one per two
template <int KG, int PROBE>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void linear_ws_kernel(const zigma_linear_params_t p, const int panels, const int ranges, const int tiles_per_xcd) {
    constexpr int NS = KG / 8;                           // slices per tile
    __shared__ __attribute__((aligned(1024))) unsigned char smem[kLds];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, kh = lane >> 5;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    if (slot >= panels * ranges) return;
    const int panel = slot % panels, range = slot / panels;              // the `panels` workgroups of a range sit in adjacent slots
    const int t_lo = xcd * tiles_per_xcd + (range * tiles_per_xcd) / ranges;
    const int my_tiles = xcd * tiles_per_xcd + ((range + 1) * tiles_per_xcd) / ranges - t_lo;
    const unsigned lds_base = static_cast<unsigned>(reinterpret_cast<uintptr_t>((lds_ptr_t)(smem)));
    const int64_t x_pitch = p.x_row_stride * 2, o_pitch = p.out_row_stride * 2, w_pitch = p.w_row_stride * 2;
    const unsigned char *xb = reinterpret_cast<const unsigned char *>(p.x) + static_cast<int64_t>(t_lo) * kT * x_pitch;

    // ---- activation stream: slice g = (tile g / NS, k columns 128 (g % NS) ..) -> ring slot g % 8; this wave's 16 rows in 4 instructions
    // instruction i: rows 16 wave + 4 i + (lane >> 4); the row pitch is a multiple of 256 B, so the swizzled slot is the low byte of the offset
    const unsigned voff0 = static_cast<unsigned>((wave * 16 + (lane >> 4)) * x_pitch) + (((lane & 15) ^ (lane >> 4)) << 4);
    const unsigned x_pitch4 = static_cast<unsigned>(4 * x_pitch);
    const unsigned dma_lds = lds_base + wave * 4096;
    auto issue_one = [&](int tile, int ks, int g, int i) {          // (wave-uniform arguments) rows 16 wave + 4 i .. + 3 of the slice
                const int tc = tile < my_tiles ? tile : my_tiles - 1;            // past the end: a harmless refill of a free slot keeps the counts uniform
        const unsigned char *src = xb + static_cast<int64_t>(tc) * kT * x_pitch + ks * 256;
        glds16(src, (voff0 ^ (i << 6)) + i * x_pitch4, dma_lds + (g & (kRing - 1)) * kSlice + i * 1024);
    };
    auto issue = [&](int tile, int ks, int g) {
#pragma unroll
        for (int i = 0; i < 4; ++i) issue_one(tile, ks, g, i);
    };
    {   // slices 0 .. 6
        int tile = 0, ks = 0;
#pragma unroll
        for (int g = 0; g < kRing - 1; ++g) {
            issue(tile, ks, g);
            if (++ks == NS) { ks = 0; ++tile; }
        }
    }
    // ---- the weights: fragment F = 2 kg + fb <- rows 256 panel + 64 wave + 32 fb + j, k = 16 kg + 8 kh .. + 7
    WFrags<KG> w;
    {   // straight into their registers (through hipcc the 2 KG loads would all be in flight in VGPRs before the first reaches an AGPR)
        const unsigned char *wp0 = reinterpret_cast<const unsigned char *>(p.w) + (static_cast<int64_t>(panel) * 256 + wave * 64 + j) * w_pitch + kh * 16;
        const unsigned char *wp1 = wp0 + 32 * w_pitch;
#pragma unroll
        for (int f = 0; f < 2 * KG; ++f) {
            const unsigned char *src = (f & 1) ? wp1 : wp0;
            if (f < WFrags<KG>::NA) ld_w_a(w.a[f], src, (f >> 1) * 32);
            else ld_w_v(w.v[f - WFrags<KG>::NA], src, (f >> 1) * 32);
        }
        wait_vm<0>();                                    // (the first slices of the activation stream too)
#pragma unroll
        for (int f = 0; f < WFrags<KG>::NA; ++f) asm volatile("" : "+a"(w.a[f]));
#pragma unroll
        for (int f = 0; f < WFrags<KG>::NV; ++f) asm volatile("" : "+v"(w.v[f]));
    }
    // B fragment of k-group kgl of a slice, token block tb: row 32 tb + j, logical 16-byte slot 2 kgl + kh
    const unsigned a_off = lds_base + j * 256 + ((kh ^ (j & 15)) << 4);
    // epilogue tile of this wave: 32 tokens x 128 B per token block; writer: token j, features 32 fb + 8 q + 4 kh .. + 3 -> 16-byte slot
    // (4 fb + q) ^ ((j >> 1) & 7), half kh; reader: token 8 i + tr, logical slot pc.  Addresses are kept as two registers + immediates
    // (left to hipcc, the 24 chunk addresses and 8 row pointers are hoisted out of the tile loop: 40 registers this kernel does not have)
    const unsigned scr = lds_base + kScrOff + wave * 8192;
    const unsigned scr_w = scr + j * 128 + kh * 8, sw_w = ((j >> 1) & 7) << 4;
    const int tr = lane >> 3, pc = lane & 7;
    const unsigned scr_r = scr + tr * 128 + ((pc ^ (tr >> 1)) << 4);             // row 8 i + tr: slot pc ^ ((row >> 1) & 7) = pc ^ (tr >> 1) ^ 4 (i & 1)
    const unsigned lane_out = static_cast<unsigned>(tr * o_pitch) + pc * 16;
    unsigned char *ob = reinterpret_cast<unsigned char *>(p.out) + static_cast<int64_t>(t_lo) * kT * o_pitch +
                        (static_cast<int64_t>(panel) * 256 + wave * 64) * 2;    // (wave-uniform)

    wait_vm<4 * (kRing - 3)>();                          // slices 0 and 1 landed (this wave's part) — hipcc has drained everything for the weights anyway
    barrier();
    u32x4 bf[2][2];
    lds_rd<0>(bf[0][0], a_off);
    lds_rd<8192>(bf[0][1], a_off);
    f32x16 acc[2][4];                                    // [tile parity][2 tb + fb]: the epilogue of a tile runs inside the next tile's k-loop
    // epilogue pieces: 16 write chunks (4 accumulator registers -> 8 bytes of the wave's LDS tile), then 8 read-back chunks (16 bytes -> one
    // store instruction = 8 tokens x 128 B); k-group qg of the NEXT tile carries write chunks (qg - W0) WP .. and read chunk qg - R0
    constexpr int WP = KG >= 32 ? 1 : 2, W0 = 1, R0 = W0 + 16 / WP + (KG >= 32 ? 2 : 0);
    static_assert(R0 + 9 <= KG, "the epilogue has to fit the k-loop");
    // a write chunk in two halves so that each fits one MFMA gap (a gap hides about five single-issue instructions):
    //   rd: four accumulator registers -> VGPRs;   wr: 2 x v_cvt_pk_bf16_f32, address, ds_write_b64
    auto wr_chunk_rd = [&](f32x16 (&pa)[4], const int c, float (&d)[4]) {      // c = 8 tb + 4 fb + q4
        const int b = 2 * (c >> 3) + ((c >> 2) & 1), q4 = c & 3;
        asm volatile("" : "+a"(pa[b]));                             // (pins the four register reads below behind this point of the asm stream)
        d[0] = pa[b][4 * q4]; d[1] = pa[b][4 * q4 + 1]; d[2] = pa[b][4 * q4 + 2]; d[3] = pa[b][4 * q4 + 3];
        asm volatile("" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]));     // (... and in front of this one)
    };
    auto wr_chunk_wr = [&](const int c, const float (&d)[4], const unsigned sw) {
        const int tb = c >> 3, fb = (c >> 2) & 1, q4 = c & 3;
        const u32x2 pk = {pack_bf2(d[0], d[1]), pack_bf2(d[2], d[3])};
        const unsigned addr = sw + (static_cast<unsigned>((fb * 4 + q4) << 4) ^ sw_w);
        if (tb == 0) asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(pk) : "memory");
        else asm volatile("ds_write_b64 %0, %1 offset:4096" ::"v"(addr), "v"(pk) : "memory");
    };
    auto wr_chunk = [&](f32x16 (&pa)[4], const int c, const unsigned sw) {
        float d[4];
        wr_chunk_rd(pa, c, d);
        wr_chunk_wr(c, d, sw);
    };
    auto rd_chunk = [&](u32x4 &o, const int r, const unsigned sr) {                    // r = 4 tb + i: tokens 32 tb + 8 i + tr
        const unsigned addr = (r & 1) ? sr ^ 64u : sr;   // rows 8 r + tr -> byte offset 1024 r
        switch (r) {
            case 0: lds_rd<0>(o, addr); break;
            case 1: lds_rd<1024>(o, addr); break;
            case 2: lds_rd<2048>(o, addr); break;
            case 3: lds_rd<3072>(o, addr); break;
            case 4: lds_rd<4096>(o, addr); break;
            case 5: lds_rd<5120>(o, addr); break;
            case 6: lds_rd<6144>(o, addr); break;
            default: lds_rd<7168>(o, addr); break;
        }
    };
    auto tile = [&](auto par_c, auto epi_c, const int t) {
        constexpr int PAR = decltype(par_c)::value;
        constexpr bool EPI = decltype(epi_c)::value && PROBE != 1 && PROBE != 3;
        unsigned char *ot = ob + static_cast<int64_t>(t - 1) * kT * o_pitch;       // rows of the PREVIOUS tile
        u32x4 o;
        // per-tile opaque copies of the three address bases: everything derived from them is computed where it is used (hoisted out of the
        // tile loop by hipcc, the 8 + 8 + 2 derived addresses cost the registers that make the k = 640 kernel spill)
        unsigned ao = a_off, sw = scr_w, sr = scr_r;
        asm volatile("" : "+v"(ao), "+v"(sw), "+v"(sr));
#pragma unroll
        for (int ks = 0; ks < NS; ++ks) {
            const int g = t * NS + ks;
            const unsigned sb = static_cast<unsigned>(g & (kRing - 1)) * kSlice, sb1 = static_cast<unsigned>((g + 1) & (kRing - 1)) * kSlice;
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                // One k-group = 4 MFMAs with everything else in the three gaps between them and behind the last one (with a single wave
                // per SIMD whatever sits in FRONT of the first MFMA runs with the matrix pipe idle):
                //   top    the fragments of this group have landed: lgkmcnt(number of LDS instructions issued behind their reads)
                //   front  fragment reads of the NEXT group
                //   gap A  store of the epilogue piece read one group ago, read-back of the next piece
                //   gap B  epilogue write chunk, first half (4 accumulator registers -> VGPRs); two of the slice's four direct-to-LDS loads
                //          (k-group 6 only)
                //   gap C  second half (-> bf16 -> LDS), third load
                //   gap D  fourth load, second write chunk (k = 512)
                const int qg = ks * 8 + q, r = qg - R0;
                const int c0 = (qg - W0) * WP;                                  // write chunks of this group: c0 .. c0 + WP - 1 where in [0, 16)
                const int cp = (qg - 1 - W0) * WP;                              // ... of the previous group
                int nw_prev = 0;
#pragma unroll
                for (int c = cp; c < cp + WP; ++c) nw_prev += (EPI && qg > 0 && c >= 0 && c < 16) ? 1 : 0;
                // ONE workgroup barrier per TWO slices (PROBE 7: per slice, the first form): in k-group 6 of every odd slice g the slices g + 1 and
                // g + 2 have landed everywhere (each wave has waited for its own parts) and the slots of g - 2 and g - 1 are free for g + 6, g + 7
                const bool two = PROBE != 7;
                const bool sync = two ? (((NS & 1) * PAR + ks) & 1) == 1 : true;         // g = NS t + ks: its parity is known at compile time
                if (q == 6 && sync) {
                    if (two) wait_vm<4 * (kRing - 5)>(); else wait_vm<4 * (kRing - 3)>();
                    if (PROBE != 3) barrier();
                }
                const bool rd_prev = EPI && r - 1 >= 0 && r - 1 < 8;             // a read-back chunk was issued in gap A of the previous group
                const int n_top = nw_prev + (rd_prev ? 1 : 0);
                if (PROBE == 5 || PROBE == 3) wait_lgkm<0>();
                else if (n_top == 0) wait_lgkm<0>();
                else if (n_top == 1) wait_lgkm<1>();
                else if (n_top == 2) wait_lgkm<2>();
                else wait_lgkm<3>();
                const bool first = qg == 0;
                // the direct-to-LDS loads behind a barrier: slice g + 6 in the gaps of k-group 6, slice g + 7 in those of k-group 7 (per-slice form: g + 7 in 6)
                const bool dma = PROBE != 3 && sync && (q == 6 || (two && q == 7));
                const int dd = two ? q : 7;                                      // slice g + dd
                if (PROBE == 5 || PROBE == 3) {          // the fragments of the NEXT group: a full group (4 MFMAs) of latency cover
                } else if (q < 7) {
                    const unsigned ad = (ao ^ ((q + 1) << 5)) + sb;
                    lds_rd<0>(bf[(q + 1) & 1][0], ad);
                    lds_rd<8192>(bf[(q + 1) & 1][1], ad);
                } else {
                    const unsigned ad = ao + sb1;
                    lds_rd<0>(bf[0][0], ad);
                    lds_rd<8192>(bf[0][1], ad);
                }
                mfma_f<KG>(acc[PAR][0], w, 2 * qg, first, bf[q & 1][0]);
                // ---- gap A
                if (rd_prev) {
                    const unsigned char *dst = ot + 8 * (r - 1) * o_pitch;      // (wave-uniform: SGPR base + 32-bit lane offset)
                    // the chunk read one group ago has landed: behind it in the LDS queue sit that group's write chunk(s) and the two fragment reads above
                    if (nw_prev == 0) wait_lgkm<2>(); else if (nw_prev == 1) wait_lgkm<3>(); else wait_lgkm<4>();
                    // (s_nop: a store of more than 8 bytes must not be followed directly by a write of its data registers — hipcc pads its
                    // own stores, it does not see this one; without it the first dword of four lanes in sixteen went out overwritten)
                    // nt: the 335 MB of an in_proj output do not belong in the L2 next to the activation slices (184 vs 203 us with the default policy)
                    if (PROBE == 2 || kWsDefaultPolicyStores) asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(lane_out), "v"(o), "s"(dst) : "memory");
                    else if (PROBE == 4) asm volatile("global_store_dwordx4 %0, %1, %2 sc0 sc1\n\ts_nop 1" ::"v"(lane_out), "v"(o), "s"(dst) : "memory");
                    else asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" ::"v"(lane_out), "v"(o), "s"(dst) : "memory");
                }
                if (EPI && r >= 0 && r < 8) rd_chunk(o, r, sr);
                if (PROBE != 6) mfma_f<KG>(acc[PAR][1], w, 2 * qg + 1, first, bf[q & 1][0]);
                // ---- gap B
                float dch[4];
                const bool chunk = EPI && c0 >= 0 && c0 < 16;
                if (chunk) wr_chunk_rd(acc[PAR ^ 1], c0, dch);
                if (dma) { issue_one(t + (ks + dd) / NS, (ks + dd) % NS, g + dd, 0); issue_one(t + (ks + dd) / NS, (ks + dd) % NS, g + dd, 1); }
                mfma_f<KG>(acc[PAR][2], w, 2 * qg, first, bf[q & 1][1]);
                // ---- gap C
                if (chunk) wr_chunk_wr(c0, dch, sw);
                if (dma) issue_one(t + (ks + dd) / NS, (ks + dd) % NS, g + dd, 2);
                if (PROBE != 6) mfma_f<KG>(acc[PAR][3], w, 2 * qg + 1, first, bf[q & 1][1]);
                // ---- gap D
                if (dma) issue_one(t + (ks + dd) / NS, (ks + dd) % NS, g + dd, 3);
                if (EPI && WP == 2 && c0 + 1 >= 0 && c0 + 1 < 16) wr_chunk(acc[PAR ^ 1], c0 + 1, sw);
            }
        }
    };
    auto epi_tail = [&](auto par_c, const int t) {       // the last tile's epilogue has no k-loop to hide in
        constexpr int PAR = decltype(par_c)::value;
        // the accumulators of the last MFMAs are read by VALU next: hipcc does not see MFMAs inside asm statements, pad by hand
        asm volatile("s_nop 15\n\ts_nop 15" : "+a"(acc[PAR][0]), "+a"(acc[PAR][1]), "+a"(acc[PAR][2]), "+a"(acc[PAR][3]));
        if (PROBE == 1 || PROBE == 3) return;
#pragma unroll
        for (int c = 0; c < 16; ++c) wr_chunk(acc[PAR], c, scr_w);
        unsigned char *ot = ob + static_cast<int64_t>(t) * kT * o_pitch;
#pragma unroll
        for (int r = 0; r < 8; ++r) {
            u32x4 o;
            rd_chunk(o, r, scr_r);
            wait_lgkm<0>();
            asm volatile("" : "+v"(o));
            __builtin_nontemporal_store(o, reinterpret_cast<u32x4 *>(ot + 8 * r * o_pitch + static_cast<uint64_t>(lane_out)));
        }
    };
    using std::integral_constant;
    tile(integral_constant<int, 0>{}, integral_constant<bool, false>{}, 0);
    int t = 1;
#pragma unroll 1
    for (; t + 1 < my_tiles; t += 2) {
        tile(integral_constant<int, 1>{}, integral_constant<bool, true>{}, t);
        tile(integral_constant<int, 0>{}, integral_constant<bool, true>{}, t + 1);
    }
    if (t < my_tiles) {
        tile(integral_constant<int, 1>{}, integral_constant<bool, true>{}, t);
        epi_tail(integral_constant<int, 1>{}, t);
    } else {
        epi_tail(integral_constant<int, 0>{}, t - 1);
    }
    wait_vm<0>();                                        // the refills past the end, the last stores
}

}  // namespace lws

// shapes the weight-stationary kernel serves
bool linear_ws_eligible(const zigma_linear_params_t &p) {
    if (p.bias || p.residual || p.silu_from_col < p.n) return false;
    if (p.k % 128 != 0 || p.k > 640 || p.k < 384 || p.n % 256 != 0 || p.n > 8192 || p.m % 512 != 0) return false;
    if (p.out_row_stride % 8 != 0 || reinterpret_cast<uintptr_t>(p.out) % 16 != 0) return false;
    const int panels = p.n / 256, ranges = 32 / panels;
    const int64_t tiles_per_xcd = p.m / 512;
    if (tiles_per_xcd < ranges || tiles_per_xcd > 0x7fffff) return false;
    if (p.out_row_stride * 2 * 8 > 0x7fffffff) return false;      // 32-bit lane offset of a store (8 rows)
    return p.x_row_stride % 128 == 0 && 64 * p.x_row_stride * 2 < 0x7fffffff;          // (slot swizzle in the low byte of the lane offset; 32-bit offsets inside a slice)
}

int launch_linear_ws(const zigma_linear_params_t &p, hipStream_t stream) {
    const int panels = p.n / 256, ranges = 32 / panels, tiles_per_xcd = static_cast<int>(p.m / 512);
    const int probe = (p.flags >> 16) & 7;
    const dim3 grid(256), block(256);
#define ZIGMA_LWS(KG_, P_) hipLaunchKernelGGL((lws::linear_ws_kernel<KG_, P_>), grid, block, 0, stream, p, panels, ranges, tiles_per_xcd)
#ifdef ZIGMA_LINEAR4W_PROBES
#define ZIGMA_LWS_K(KG_) { if (probe == 1) ZIGMA_LWS(KG_, 1); else if (probe == 2) ZIGMA_LWS(KG_, 2); else if (probe == 3) ZIGMA_LWS(KG_, 3); else if (probe == 4) ZIGMA_LWS(KG_, 4); else if (probe == 5) ZIGMA_LWS(KG_, 5); else if (probe == 6) ZIGMA_LWS(KG_, 6); else if (probe == 7) ZIGMA_LWS(KG_, 7); else ZIGMA_LWS(KG_, 0); }
#else
#define ZIGMA_LWS_K(KG_) { if (probe) return ZIGMA_ERR_UNSUPPORTED; ZIGMA_LWS(KG_, 0); }
#endif
    switch (p.k / 16) {
        case 40: ZIGMA_LWS_K(40) break;
        case 32: ZIGMA_LWS_K(32) break;
        case 24: ZIGMA_LWS_K(24) break;
        default: return ZIGMA_ERR_SHAPE;
    }
#undef ZIGMA_LWS_K
#undef ZIGMA_LWS
    set_last_kernel("linear_ws");
    return check_launch();
}

}  // namespace zigma
