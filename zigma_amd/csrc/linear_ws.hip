// linear_ws: out = x @ W^T with the WEIGHTS STATIONARY IN REGISTERS, gfx950 — in_proj of the ZigMa block (k = 640, 256-feature panels).
// Reference call site: mamba_simple.py:290-294, F.linear.
//
// Why a second GEMM form: with 256 x 256 tiles (linear4w) every k-step brings 32 KB of activations AND 32 KB of weights from L2 into
// LDS, and that stream (~8 TB/s effective over the chip) — not the matrix pipe — bounds all projections of the block at 40-45 % of the
// MFMA peak.  A panel of W of 320 KB is 320 registers per lane of four waves: it fits the 512-register file of a one-wave-per-SIMD
// workgroup next to the accumulators (256 features x k = 640, or 128 features x k = 1280).  Then only the tokens stream, each
// activation slice is read by the `panels` workgroups of its range at the same time (one L2 fill), and the weights are read once per
// workgroup.
//
//   workgroup = 4 waves, one per SIMD, one per CU, persistent: (panel of 128 FB features, range of 64-token tiles inside its XCD's
//   eighth of the tokens).  Wave w keeps W rows [128 FB panel + 32 FB w, + 32 FB) x k as MFMA A fragments (FB blocks of 32 features
//   x k / 16 fragments of 4 registers; as many as fit in AGPRs, the rest in VGPRs), accumulates D[32 FB features][64 tokens] in
//   32 FB AGPRs — two sets, the epilogue of a tile rides in the MFMA gaps of the next one.
//   Tokens: slices of 64 tokens x 128 k (16 KB, global_load_lds_dwordx4, 16-byte slots XOR-swizzled by the row on the source side)
//   through a ring of eight; ONE counted vmcnt + barrier per two slices; B fragments by ds_read_b128, one k-group ahead.
//   Epilogue per tile: accumulators -> bf16 -> the wave's LDS tile -> 16-byte stores (64 FB contiguous bytes per token and wave).
// Limits: bf16, no bias / residual, SiLU only on whole 128-column groups from silu_from_col on; k = 512 or 640 with n % 256 == 0, or (FB = 1) k = 1280 / 1536
// with n % 128 == 0; m % 512 == 0.
// FB is a template parameter: the 128-feature form (FB = 1: one MFMA per fragment read, k up to 1280 — out_proj / to_out shapes) was
// instantiated, is bit-identical too and does NOT beat the tiled kernel: out_proj shape 102 vs 97 us, to_out shape 48 vs 45 us, its
// MFMA-only loop 74.5 us against 65 at the FB = 2 rate (profiles/r04_h_linear_ws_probe_128_panels.jsonl) — not shipped.
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
template <int OFF>
__device__ __forceinline__ void lds_wr8(unsigned addr, const u32x2 &v) { asm volatile("ds_write_b64 %0, %1 offset:%2" ::"v"(addr), "v"(v), "n"(OFF) : "memory"); }
__device__ __forceinline__ void glds16(const void *base, unsigned voff, unsigned lds) {
    asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" ::"v"(voff), "s"(base), "s"(lds) : "memory");
}
__device__ __forceinline__ void ld_w_a(u32x4 &d, const void *ptr, const int off) { asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=a"(d) : "v"(ptr), "n"(off)); }
__device__ __forceinline__ void ld_w_v(u32x4 &d, const void *ptr, const int off) { asm volatile("global_load_dwordx4 %0, %1, off offset:%2" : "=v"(d) : "v"(ptr), "n"(off)); }
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory"); }
template <int N> __device__ __forceinline__ void wait_lgkm() { asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory"); }
__device__ __forceinline__ void wait_lgkm_n(const int n) {           // n folds to a constant in the unrolled loop
    if (n <= 0) wait_lgkm<0>(); else if (n == 1) wait_lgkm<1>(); else if (n == 2) wait_lgkm<2>(); else if (n == 3) wait_lgkm<3>(); else wait_lgkm<4>();
}
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

template <int KG, int FB>
struct WFrags {        // fragment f = FB kg + fb; 256 AGPRs = two accumulator sets of 2 FB blocks + the first NA fragments
    static constexpr int NF = FB * KG, NAmax = (256 - 2 * 2 * FB * 16) / 4, NA = NF < NAmax ? NF : NAmax, NV = NF - NA;
    u32x4 a[NA];
    u32x4 v[NV > 0 ? NV : 1];
};

// (f and first fold to constants in the unrolled loop: one instruction survives)
template <int KG, int FB>
__device__ __forceinline__ void mfma_f(f32x16 &acc, const WFrags<KG, FB> &w, const int f, const bool first, const u32x4 &b) {
    if (f < WFrags<KG, FB>::NA) { if (first) mfma_wa<true>(acc, w.a[f], b); else mfma_wa<false>(acc, w.a[f], b); }
    else { if (first) mfma_wv<true>(acc, w.v[f - WFrags<KG, FB>::NA], b); else mfma_wv<false>(acc, w.v[f - WFrags<KG, FB>::NA], b); }
}

// PROBE (tools/linear_ws_probe.py; probe builds only): 0 = the kernel, 1 = no epilogue, 2 = default-policy stores instead of nt,
// 3 = MFMAs only (no stream after the prologue, no barriers, no fragment reads, no epilogue), 4 = the fragment reads in front of the first MFMA
// of the k-group instead of behind it
// SL (round 5): output columns >= p.silu_from_col (a multiple of 64 FB, so a wave is all-or-nothing) leave as silu(.) — in_proj writing the
// PRE-ACTIVATED gate half for the scan (ZIGMA_SCAN_Z_PREACTIVATED).  The 20 instructions per write chunk (4 values: -log2e *, v_exp, 1 +,
// v_rcp, * x) are spread over the MFMA gaps 1 .. 3 of the chunk's k-group, the LDS write moves from gap 1 to gap 3.
template <int KG, int FB, int PROBE, bool SL = false>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void linear_ws_kernel(const zigma_linear_params_t p, const int panels, const int ranges, const int tiles_per_xcd) {
    constexpr int NS = KG / 8;                           // slices per tile
    constexpr int NB = 2 * FB;                           // accumulator blocks per set: [FB tb + fb]
    constexpr int RB = 64 * FB;                          // bytes per token of a wave's output / of its LDS tile
    constexpr int PPR = RB / 16, TPR = 64 / PPR;         // 16-byte pieces per token row; tokens per read-back / store instruction
    constexpr int NWC = NB * 4, NRC = PPR;               // epilogue pieces per tile: write chunks (4 accumulator registers), read-back chunks
    typedef WFrags<KG, FB> WF;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[kLds];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, kh = lane >> 5;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    if (slot >= panels * ranges) return;
    const int panel = slot % panels, range = slot / panels;              // the `panels` workgroups of a range sit in adjacent slots
    const bool do_silu = SL && panel * 128 * FB + wave * 32 * FB >= p.silu_from_col;       // (wave-uniform)
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
        const int tc = tile < my_tiles ? tile : my_tiles - 1;      // past the end: a harmless refill of a free slot keeps the counts uniform
        const unsigned char *src = xb + static_cast<int64_t>(tc) * kT * x_pitch + ks * 256;
        glds16(src, (voff0 ^ (i << 6)) + i * x_pitch4, dma_lds + (g & (kRing - 1)) * kSlice + i * 1024);
    };
    {   // slices 0 .. 6
        int tile = 0, ks = 0;
#pragma unroll
        for (int g = 0; g < kRing - 1; ++g) {
#pragma unroll
            for (int i = 0; i < 4; ++i) issue_one(tile, ks, g, i);
            if (++ks == NS) { ks = 0; ++tile; }
        }
    }
    // ---- the weights: fragment f = FB kg + fb <- rows 128 FB panel + 32 FB wave + 32 fb + j, k = 16 kg + 8 kh .. + 7
    WF w;
    {   // straight into their registers (through hipcc all the loads would be in flight in VGPRs before the first reaches an AGPR)
        const unsigned char *wp0 = reinterpret_cast<const unsigned char *>(p.w) + (static_cast<int64_t>(panel) * 128 * FB + wave * 32 * FB + j) * w_pitch + kh * 16;
        const unsigned char *wp1 = wp0 + 32 * w_pitch;
#pragma unroll
        for (int f = 0; f < WF::NF; ++f) {
            const unsigned char *src = (FB == 2 && (f & 1)) ? wp1 : wp0;
            if (f < WF::NA) ld_w_a(w.a[f], src, (f / FB) * 32);
            else ld_w_v(w.v[f - WF::NA], src, (f / FB) * 32);
        }
        wait_vm<0>();                                    // (the first slices of the activation stream too)
#pragma unroll
        for (int f = 0; f < WF::NA; ++f) asm volatile("" : "+a"(w.a[f]));
#pragma unroll
        for (int f = 0; f < WF::NV; ++f) asm volatile("" : "+v"(w.v[f]));
    }
    // B fragment of k-group kgl of a slice, token block tb: row 32 tb + j, logical 16-byte slot 2 kgl + kh
    const unsigned a_off = lds_base + j * 256 + ((kh ^ (j & 15)) << 4);
    // epilogue tile of this wave: 32 tokens x RB bytes per token block; writer: token j, features 32 fb + 8 q + 4 kh .. + 3 -> 16-byte slot
    // (4 fb + q) ^ ((j >> 1) & (PPR - 1)), half kh; reader: token TPR i + tr, logical slot pc.  Addresses are kept as two registers +
    // immediates (left to hipcc, the chunk addresses and row pointers are hoisted out of the tile loop: 40 registers the k = 640 kernel does not have)
    const unsigned scr = lds_base + kScrOff + wave * 8192;
    const unsigned scr_w = scr + j * RB + kh * 8, sw_w = ((j >> 1) & (PPR - 1)) << 4;
    const int tr = lane / PPR, pc = lane % PPR;
    // row TPR i + tr: physical slot pc ^ ((row >> 1) & (PPR - 1)); FB = 2: = pc ^ (tr >> 1) ^ 4 (i & 1); FB = 1: = pc ^ ((tr >> 1) & 3)
    const unsigned scr_r = scr + tr * RB + ((pc ^ ((tr >> 1) & (PPR - 1))) << 4);
    const unsigned lane_out = static_cast<unsigned>(tr * o_pitch) + pc * 16;
    unsigned char *ob = reinterpret_cast<unsigned char *>(p.out) + static_cast<int64_t>(t_lo) * kT * o_pitch +
                        (static_cast<int64_t>(panel) * 128 * FB + wave * 32 * FB) * 2;    // (wave-uniform)

    wait_vm<4 * (kRing - 3)>();                          // slices 0 and 1 landed (this wave's part) — the weights' wait has drained everything anyway
    barrier();
    u32x4 bf[2][2];
    lds_rd<0>(bf[0][0], a_off);
    lds_rd<8192>(bf[0][1], a_off);
    f32x16 acc[2][NB];                                   // [tile parity][FB tb + fb]: the epilogue of a tile runs inside the next tile's k-loop
    // epilogue pieces of the PREVIOUS tile inside k-group qg of this one: write chunk qg - W0 (4 accumulator registers -> 8 bytes of the
    // wave's LDS tile; in two halves so that each fits one MFMA gap — a gap hides about five single-issue instructions), then read-back chunk
    // qg - R0 (16 bytes per lane) whose store (TPR tokens x RB bytes) follows one k-group later
    constexpr int W0 = 1, R0 = W0 + NWC + 2;
    static_assert(R0 + NRC + 1 <= KG, "the epilogue has to fit the k-loop");
    auto wr_chunk_rd = [&](f32x16 (&pa)[NB], const int c, float (&d)[4]) {      // c = 4 (FB tb + fb) + q4
        const int b = c >> 2, q4 = c & 3;
        asm volatile("" : "+a"(pa[b]));                             // (pins the four register reads below behind this point of the asm stream)
        d[0] = pa[b][4 * q4]; d[1] = pa[b][4 * q4 + 1]; d[2] = pa[b][4 * q4 + 2]; d[3] = pa[b][4 * q4 + 3];
        asm volatile("" : "+v"(d[0]), "+v"(d[1]), "+v"(d[2]), "+v"(d[3]));     // (... and in front of this one)
    };
    // silu(d) = d / (1 + exp2(-log2e d)) in three stages of four independent instructions each pair (one MFMA gap per stage)
    auto silu_a = [&](const float (&d)[4], float (&e)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) e[i] = __builtin_amdgcn_exp2f(d[i] * -1.4426950408889634f);
        asm volatile("" : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]));
    };
    auto silu_b = [&](float (&e)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) e[i] = __builtin_amdgcn_rcpf(1.f + e[i]);
        asm volatile("" : "+v"(e[0]), "+v"(e[1]), "+v"(e[2]), "+v"(e[3]));
    };
    auto silu_c = [&](float (&d)[4], const float (&e)[4]) {
#pragma unroll
        for (int i = 0; i < 4; ++i) d[i] *= e[i];
    };
    auto wr_chunk_wr = [&](const int c, const float (&d)[4], const unsigned sw) {
        const int tb = (c >> 2) / FB, fb = (c >> 2) % FB, q4 = c & 3;
        const u32x2 pk = {pack_bf2(d[0], d[1]), pack_bf2(d[2], d[3])};
        const unsigned addr = sw + (static_cast<unsigned>((fb * 4 + q4) << 4) ^ sw_w);
        if (tb == 0) lds_wr8<0>(addr, pk); else lds_wr8<32 * RB>(addr, pk);
    };
    auto rd_chunk = [&](u32x4 &o, const int r, const unsigned sr) {            // rows TPR r + tr of the tile -> byte offset TPR RB r = 1024 r
        const unsigned addr = (FB == 2 && (r & 1)) ? sr ^ 64u : sr;
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
    auto st_chunk = [&](const u32x4 &o, unsigned char *ot, const int r) {      // tokens TPR r + tr (SGPR base + 32-bit lane offset)
        const unsigned char *dst = ot + TPR * r * o_pitch;
        // (s_nop: a store of more than 8 bytes must not be followed directly by a write of its data registers — hipcc pads its own stores,
        // it does not see this one; without it the first dword of four lanes in sixteen went out overwritten)
        // nt: an in_proj output (335 MB) does not belong in the L2 next to the activation slices (stand-alone 184 vs 203 us)
        if (PROBE == 2 || kWsDefaultPolicyStores) asm volatile("global_store_dwordx4 %0, %1, %2\n\ts_nop 1" ::"v"(lane_out), "v"(o), "s"(dst) : "memory");
        else asm volatile("global_store_dwordx4 %0, %1, %2 nt\n\ts_nop 1" ::"v"(lane_out), "v"(o), "s"(dst) : "memory");
    };
    auto tile = [&](auto par_c, auto epi_c, const int t) {
        constexpr int PAR = decltype(par_c)::value;
        constexpr bool EPI = decltype(epi_c)::value && PROBE != 1 && PROBE != 3;
        unsigned char *ot = ob + static_cast<int64_t>(t - 1) * kT * o_pitch;       // rows of the PREVIOUS tile
        u32x4 o;
        float dch[4], ech[4];
        // per-tile opaque copies of the three address bases: everything derived from them is computed where it is used
        unsigned ao = a_off, sw = scr_w, sr = scr_r;
        asm volatile("" : "+v"(ao), "+v"(sw), "+v"(sr));
#pragma unroll
        for (int ks = 0; ks < NS; ++ks) {
            const int g = t * NS + ks;
            const unsigned sb = static_cast<unsigned>(g & (kRing - 1)) * kSlice, sb1 = static_cast<unsigned>((g + 1) & (kRing - 1)) * kSlice;
            // ONE workgroup barrier per TWO slices: in k-group QS of every odd slice g the slices g + 1 and g + 2 have landed everywhere
            // (each wave has waited for its own parts) and the slots of g - 2 and g - 1 are free for g + 6, g + 7
            const bool sync = (((NS & 1) * PAR + ks) & 1) == 1;             // g = NS t + ks: its parity is known at compile time
            constexpr int QS = 4, LPG = 2;           // first k-group of the loads behind the barrier; loads per k-group (four per group from 6 on: 180.8 vs 177.6 us)
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                // One k-group = 2 FB MFMAs with everything else in the gaps behind them (with a single wave per SIMD whatever sits in
                // FRONT of the first MFMA runs with the matrix pipe idle):
                //   front  the fragments of this group have landed: lgkmcnt(LDS instructions issued behind their reads)
                //   gap 0  the fragment reads of the NEXT group; store of the epilogue piece read one group ago, read-back of the next piece | write
                //          chunk, first half
                //   gap 1  write chunk, second half; direct-to-LDS loads (LPG per k-group from QS on, in the last gaps)
                const int qg = ks * 8 + q, r = qg - R0, c = qg - W0;
                const bool wr_prev = EPI && c - 1 >= 0 && c - 1 < NWC, rd_prev = EPI && r - 1 >= 0 && r - 1 < NRC;
                const bool wr_now = EPI && c >= 0 && c < NWC, rd_now = EPI && r >= 0 && r < NRC;
                if (q == QS && sync) {
                    wait_vm<4 * (kRing - 5)>();
                    if (PROBE != 3) barrier();
                }
                if (PROBE == 3) wait_lgkm<0>(); else wait_lgkm_n((wr_prev ? 1 : 0) + (rd_prev ? 1 : 0));
                const bool first = qg == 0;
                auto frag_reads = [&]() {                // the fragments of the NEXT k-group (this slice or, from k-group 7, the next one)
                    if (PROBE == 3) return;
                    const unsigned ad = q < 7 ? (ao ^ ((q + 1) << 5)) + sb : ao + sb1;
                    lds_rd<0>(bf[(q + 1) & 1][0], ad);
                    lds_rd<8192>(bf[(q + 1) & 1][1], ad);
                };
                if (PROBE == 4) frag_reads();            // (PROBE 4: in front of the first MFMA, where the matrix pipe waits for them to issue: 180 vs 175 us)
                // the direct-to-LDS loads behind a barrier: slice g + 6 first, then slice g + 7, LPG per k-group
                const int dl = (q - QS) * LPG;                                   // first load of this k-group: 0 .. 7
                const bool dma = PROBE != 3 && sync && q >= QS && dl < 8;
#pragma unroll
                for (int mi = 0; mi < NB; ++mi) {
                    const int tb = mi / FB, fb = mi % FB;
                    mfma_f<KG, FB>(acc[PAR][mi], w, FB * qg + fb, first, bf[q & 1][tb]);
                    if (mi == 0) {                       // ---- gap 0
                        if (PROBE != 4) frag_reads();
                        if (rd_prev) {
                            wait_lgkm<2>();              // the chunk read one group ago has landed: behind it in the LDS queue sit only the two fragment reads above
                            st_chunk(o, ot, r - 1);
                        }
                        if (rd_now) rd_chunk(o, r, sr);
                        if (wr_now) wr_chunk_rd(acc[PAR ^ 1], c, dch);
                    }
                    if constexpr (SL) {                  // ---- gaps 1 .. 3: the activation of the gate panels, then the write
                        if (wr_now && do_silu) {
                            if (mi == 1) silu_a(dch, ech);
                            if (mi == 2) silu_b(ech);
                            if (mi == 3) silu_c(dch, ech);
                        }
                        if (mi == NB - 1 && wr_now) wr_chunk_wr(c, dch, sw);
                    } else {
                        if (mi == 1 && wr_now) wr_chunk_wr(c, dch, sw);             // ---- gap 1
                    }
                    if (dma) {                           // the last LPG gaps: FB = 2: gaps 2 and 3; FB = 1: gaps 0 and 1
                        const int first_l = FB == 2 ? mi - 2 : mi;
                        const int n_l = FB == 2 ? (mi >= 2 ? 1 : 0) : 1;
#pragma unroll
                        for (int li = 0; li < n_l; ++li) {
                            const int l = dl + first_l + li, dd = 6 + (l >> 2);  // load l & 3 of slice g + dd
                            issue_one(t + (ks + dd) / NS, (ks + dd) % NS, g + dd, l & 3);
                        }
                    }
                }
            }
        }
    };
    auto epi_tail = [&](auto par_c, const int t) {       // the last tile's epilogue has no k-loop to hide in
        constexpr int PAR = decltype(par_c)::value;
        // the accumulators of the last MFMAs are read by VALU next: hipcc does not see MFMAs inside asm statements, pad by hand
        if constexpr (FB == 2) asm volatile("s_nop 15\n\ts_nop 15" : "+a"(acc[PAR][0]), "+a"(acc[PAR][1]), "+a"(acc[PAR][2]), "+a"(acc[PAR][3]));
        else asm volatile("s_nop 15\n\ts_nop 15" : "+a"(acc[PAR][0]), "+a"(acc[PAR][1]));
        if (PROBE == 1 || PROBE == 3) return;
#pragma unroll
        for (int c = 0; c < NWC; ++c) {
            float d[4];
            wr_chunk_rd(acc[PAR], c, d);
            if (SL && do_silu) {
                float e[4];
                silu_a(d, e); silu_b(e); silu_c(d, e);
            }
            wr_chunk_wr(c, d, scr_w);
        }
        unsigned char *ot = ob + static_cast<int64_t>(t) * kT * o_pitch;
#pragma unroll
        for (int r = 0; r < NRC; ++r) {
            u32x4 o;
            rd_chunk(o, r, scr_r);
            wait_lgkm<0>();
            st_chunk(o, ot, r);
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

// features per panel the weight-stationary kernel uses for the call (256: k = 512 / 640, two 32-feature blocks per wave; 128: k = 1280 / 1536, one
// block per wave — the out_proj shapes at serving-size token counts, round 5), or 0 if it does not serve it
static int linear_ws_panel(const zigma_linear_params_t &p) {
    if (p.bias || p.residual) return 0;
    const bool narrow = p.k == 1280 || p.k == 1536;
    if (!narrow && p.k != 512 && p.k != 640) return 0;                                                // (instantiation set: k / 16 = 32, 40 | 80, 96)
    const int pw = narrow ? 128 : 256;
    if (p.silu_from_col < p.n && (narrow || p.silu_from_col < 0 || p.silu_from_col % 128 != 0)) return 0;     // a wave (64 features) is all-or-nothing
    if (p.n % pw != 0 || p.n > 8192 || p.m % 512 != 0) return 0;
    if (p.out_row_stride % 8 != 0 || reinterpret_cast<uintptr_t>(p.out) % 16 != 0) return 0;
    const int panels = p.n / pw;
    if (panels > 32) return 0;
    const int ranges = 32 / panels;
    const int64_t tiles_per_xcd = p.m / 512;
    if (tiles_per_xcd < ranges || tiles_per_xcd > 0x7fffff) return 0;
    if (p.out_row_stride * 2 * 16 > 0x7fffffff) return 0;      // 32-bit lane offset of a store (up to 16 rows)
    // (slot swizzle in the low byte of the lane offset; 32-bit offsets inside a slice)
    if (p.x_row_stride % 128 != 0 || 64 * p.x_row_stride * 2 >= 0x7fffffff) return 0;
    return pw;
}

bool linear_ws_eligible(const zigma_linear_params_t &p) { return linear_ws_panel(p) != 0; }

int launch_linear_ws(const zigma_linear_params_t &p, hipStream_t stream) {
    const int pw = linear_ws_panel(p);
    if (!pw) return ZIGMA_ERR_UNSUPPORTED;
    const int panels = p.n / pw, ranges = 32 / panels, tiles_per_xcd = static_cast<int>(p.m / 512);
    const int probe = (p.flags >> 16) & 7;
    const dim3 grid(256), block(256);
    const bool sl = p.silu_from_col < p.n;
#define ZIGMA_LWS(KG_, FB_, P_) do { if (sl) hipLaunchKernelGGL((lws::linear_ws_kernel<KG_, FB_, P_, true>), grid, block, 0, stream, p, panels, ranges, tiles_per_xcd); \
                                     else hipLaunchKernelGGL((lws::linear_ws_kernel<KG_, FB_, P_, false>), grid, block, 0, stream, p, panels, ranges, tiles_per_xcd); } while (0)
#ifdef ZIGMA_LINEAR4W_PROBES
#define ZIGMA_LWS_K(KG_, FB_) { if (probe == 1) ZIGMA_LWS(KG_, FB_, 1); else if (probe == 2) ZIGMA_LWS(KG_, FB_, 2); else if (probe == 3) ZIGMA_LWS(KG_, FB_, 3); else if (probe == 4) ZIGMA_LWS(KG_, FB_, 4); else ZIGMA_LWS(KG_, FB_, 0); }
#else
#define ZIGMA_LWS_K(KG_, FB_) { if (probe) return ZIGMA_ERR_UNSUPPORTED; ZIGMA_LWS(KG_, FB_, 0); }
#endif
    if (p.k == 640) ZIGMA_LWS_K(40, 2) else if (p.k == 512) ZIGMA_LWS_K(32, 2)
    else {      // one 32-feature block per wave (128-feature panels): k = 1280 / 1536
        if (probe || sl) return ZIGMA_ERR_UNSUPPORTED;
        if (p.k == 1280) hipLaunchKernelGGL((lws::linear_ws_kernel<80, 1, 0, false>), grid, block, 0, stream, p, panels, ranges, tiles_per_xcd);
        else hipLaunchKernelGGL((lws::linear_ws_kernel<96, 1, 0, false>), grid, block, 0, stream, p, panels, ranges, tiles_per_xcd);
    }
#undef ZIGMA_LWS_K
#undef ZIGMA_LWS
    set_last_kernel(sl ? "linear_ws_silu" : pw == 128 ? "linear_ws_128" : "linear_ws");
    return check_launch();
}

}  // namespace zigma
