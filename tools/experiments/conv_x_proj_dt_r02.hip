// conv_x_proj: depthwise causal conv1d (+ bias, SiLU) over the zigzag-reordered sequence AND the skinny x_proj product of the
// result, in one pass over x, gfx950.  C ABI: zigma_conv_x_proj_fwd.
//
// Replaces, together, causal_conv1d_fn(x, conv1d_weight, conv1d_bias, activation="silu") over the gathered sequence and
// F.linear(conv1d_out, x_proj_weight) of MambaInnerFn.forward (reference selective_scan_interface.py:307-322, gather of
// mamba_simple.py:362-370).  Apart, the two kernels move  read x + write u + read u  (3 x 168 MB at the headline shape);
// here u is produced in registers in exactly the MFMA A-fragment layout of the projection, so it is written once (the scan
// needs it) and never read back:  read x + write u.
//
//   workgroup = 128 positions (scan order) = 4 waves x 32 positions, two workgroups per CU; K = d_inner is walked in stages of
//   64 channels.  Per stage every wave fetches ITS 35 input rows (3 halo + 32, picked through the row table, one full 128-byte
//   line each) straight into LDS (global_load_lds_dwordx4, source-side swizzle), the workgroup shares one 64-channel slab of W_x
//   (n rows x 128 B) and of the conv taps/bias; stage s + 1 is in flight while stage s is consumed.
//   A lane = (position j, 8 adjacent channels): the four taps are the LDS rows j .. j+3 of its own channels (no cross-lane
//   traffic), the conv is 2 x v_dot2c_f32_bf16 per channel on (tap, tap) pairs built with v_perm_b32 against the taps as they
//   lie in the (d_inner, 4) bf16 weight, SiLU, pack: the 8 bf16 are the A fragment of v_mfma_f32_32x32x16_bf16 against the W_x
//   slab (32 x 96 fp32 accumulator per wave, rows beyond n are never stored); u leaves once per stage as full 128-byte lines,
//   transposed through the wave's own (consumed) input rows.
//   The products are taken transposed (W_x rows as the MFMA A operand, D[n][position]) so that a lane holds 4 consecutive n of its
//   position: the x_dbl tile goes through LDS as 8-byte pieces and leaves as 16-byte row pieces.  Optionally (delta != NULL) each
//   wave then runs dt_proj.hip's wave tile on that LDS tile for all d_inner channels: delta = softplus(x_dbl[:, :R] W_dt^T + b) —
//   bit-identical to the stand-alone kernel and NOT faster than it (138.7 vs 71.3 + 64.4 us), so the host leaves it off.
// Measured at the headline shape (B=64, L=1024, d_inner=1280, n=72): 72 us against 131 us for conv_tok + x_proj_mfma
// (tools/conv_xproj_ab.py); loads + MFMA alone 35 us, + conv 64 us, + stores 64 us (probe flags).
// bf16 only; width 4; bias required; seqlen % 32 == 0; d_inner % 64 == 0; n <= 96, n % 8 == 0.
#include "zigma_common.h"

namespace zigma {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) unsigned char *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *gbl_ptr_t;

constexpr int kCxTok = 32, kCxBK = 64, kCxHalo = 3;
constexpr int kCxXBytes = 40 * 128;                                  // per wave and stage: 35 rows used, 5 load instructions
// stage layout for NW waves: [NW x input rows][W_x slab: 96 rows x 128 B][conv taps (64 ch x 8 B) + bias (64 x 2 B), 1 KB]
constexpr int cx_w_off(int nw) { return nw * kCxXBytes; }
constexpr int cx_c_off(int nw) { return cx_w_off(nw) + 96 * 128; }
constexpr int cx_stage(int nw) { return cx_c_off(nw) + 1024; }      // 8 waves: 54272 B, 4 waves: 33792 B

__device__ __forceinline__ float bf_lo(unsigned v) { return __uint_as_float(v << 16); }
__device__ __forceinline__ float bf_hi(unsigned v) { return __uint_as_float(v & 0xffff0000u); }
__device__ __forceinline__ float dot2(unsigned a, unsigned b, float c) {
    return __builtin_amdgcn_fdot2_f32_bf16(__builtin_bit_cast(bf16x2, a), __builtin_bit_cast(bf16x2, b), c, false);
}
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned pack_bf2(float lo, float hi) {        // one v_cvt_pk_bf16_f32 (round to nearest even)
    return __builtin_bit_cast(unsigned, __builtin_convertvector(f32x2{lo, hi}, bf16x2));
}
// LDS reads are inline assembly on purpose: hipcc makes a ds_read it can see wait for EVERY direct-to-LDS load in flight
// (s_waitcnt vmcnt(0)), the younger stages included, which would serialise the pipeline.  Landing is tracked by hand (counted
// s_waitcnt vmcnt + s_barrier at the top of a stage); the reads of one k-step are issued as a group and settled by an explicit
// s_waitcnt lgkmcnt that names the destination registers (so that no use can be scheduled above it).
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
struct KStep {
    u32x4 X[4];      // input rows k-3 .. k, this lane's 8 channels
    u32x4 Wc[4];     // conv taps: [r] = channels 2r, 2r+1 as (tap0 tap1)(tap2 tap3) pairs
    u32x4 Bc;        // conv bias of the 8 channels
    u32x4 Bf[3];     // W_x fragments (rows nb * 32 + j)
};
__device__ __forceinline__ void lds_rd(u32x4 &d, unsigned addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr)); }
template <int N>
__device__ __forceinline__ void settle(KStep &k) {
    static_assert(N == 0 || N == 12, "");
    if (N == 0)
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(k.X[0]), "+v"(k.X[1]), "+v"(k.X[2]), "+v"(k.X[3]), "+v"(k.Wc[0]), "+v"(k.Wc[1]), "+v"(k.Wc[2]),
                     "+v"(k.Wc[3]), "+v"(k.Bc), "+v"(k.Bf[0]), "+v"(k.Bf[1]), "+v"(k.Bf[2]));
    else
        asm volatile("s_waitcnt lgkmcnt(12)" : "+v"(k.X[0]), "+v"(k.X[1]), "+v"(k.X[2]), "+v"(k.X[3]), "+v"(k.Wc[0]), "+v"(k.Wc[1]), "+v"(k.Wc[2]),
                     "+v"(k.Wc[3]), "+v"(k.Bc), "+v"(k.Bf[0]), "+v"(k.Bf[1]), "+v"(k.Bf[2]));
}

// s_waitcnt vmcnt(n), n a run-time (wave-uniform) value: the instruction takes an immediate
__device__ __forceinline__ void wait_vm_n(int n) {
    switch (n) {
#define ZIGMA_VM_CASE(N_) case N_: asm volatile("s_waitcnt vmcnt(" #N_ ")" ::: "memory"); break;
        ZIGMA_VM_CASE(1) ZIGMA_VM_CASE(2) ZIGMA_VM_CASE(3) ZIGMA_VM_CASE(4) ZIGMA_VM_CASE(5) ZIGMA_VM_CASE(6) ZIGMA_VM_CASE(7)
        ZIGMA_VM_CASE(8) ZIGMA_VM_CASE(9) ZIGMA_VM_CASE(10) ZIGMA_VM_CASE(11) ZIGMA_VM_CASE(12) ZIGMA_VM_CASE(13) ZIGMA_VM_CASE(14)
        ZIGMA_VM_CASE(15) ZIGMA_VM_CASE(16)
#undef ZIGMA_VM_CASE
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

template <int NST, int NW, bool DT>
__global__ __launch_bounds__(64 * NW) void conv_x_proj_kernel(const zigma_conv_xproj_params_t p) {
    constexpr int kCxWaves = NW, kCxWOff = cx_w_off(NW), kCxCOff = cx_c_off(NW), kCxStage = cx_stage(NW), NWI = (12 + NW - 1) / NW;
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NST * kCxStage];
    const int dbg = p.flags;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, kh = lane >> 5;
    const int L = p.seqlen;
    const int64_t m0 = (static_cast<int64_t>(blockIdx.x) * kCxWaves + wave) * kCxTok;      // first position of this wave (all samples)
    const int b = static_cast<int>(m0 / L), t0 = static_cast<int>(m0 - static_cast<int64_t>(b) * L);
    const int n_stages = p.dim / kCxBK;
    const unsigned smem_lds = static_cast<unsigned>(reinterpret_cast<uintptr_t>((lds_ptr_t)(smem)));      // LDS byte address

    // ---- sources of the per-stage loads (fixed over the stages apart from the + 128 B per stage) ----
    const unsigned char *xsrc[5];
    {
        const unsigned char *xb = reinterpret_cast<const unsigned char *>(p.x) + static_cast<int64_t>(b) * p.x_batch_stride * 2;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            int rr = i * 8 + (lane >> 3);
            rr = rr < kCxTok + kCxHalo ? rr : kCxTok + kCxHalo - 1;
            int pos = t0 - kCxHalo + rr;
            pos = pos < 0 ? 0 : pos;                                   // left padding: fetched from a valid row, zeroed when read
            const int row = p.x_row_index ? p.x_row_index[pos] : pos;
            xsrc[i] = xb + static_cast<int64_t>(row) * p.x_l_stride * 2 + (((lane & 7) ^ ((rr >> 1) & 7)) << 4);
        }
    }
    // W_x slab: 8 rows per load instruction, instruction q = wave + NW i (wave-uniform: skipped when its rows are all beyond n)
    const unsigned char *wsrc[NWI];
    int n_w = 0;
#pragma unroll
    for (int i = 0; i < NWI; ++i) {
        int row = (wave + NW * i) * 8 + (lane >> 3);
        row = row < p.n ? row : p.n - 1;
        wsrc[i] = reinterpret_cast<const unsigned char *>(p.w) + static_cast<int64_t>(row) * p.w_row_stride * 2 + (((lane & 7) ^ ((row >> 1) & 7)) << 4);
        n_w += (wave + NW * i) * 8 < p.n ? 1 : 0;
    }
    // conv taps / bias slab (wave 1): lanes 0..31 taps of 2 channels each, lanes 32..39 bias of 8 channels each
    const unsigned char *csrc = lane < 32 ? reinterpret_cast<const unsigned char *>(p.conv_weight) + lane * 16
                                          : reinterpret_cast<const unsigned char *>(p.conv_bias) + ((lane < 40 ? lane : 39) - 32) * 16;
    const int c_step = lane < 32 ? kCxBK * 8 : kCxBK * 2;              // bytes per stage

    auto issue = [&](int st) {
        unsigned char *dst = smem + (st % NST) * kCxStage;
#pragma unroll
        for (int i = 0; i < 5; ++i)
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(xsrc[i] + st * (kCxBK * 2)), (lds_ptr_t)(dst) + wave * kCxXBytes + i * 1024, 16, 0, 0);
#pragma unroll
        for (int i = 0; i < NWI; ++i)
            if (i < n_w)
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)(wsrc[i] + st * (kCxBK * 2)), (lds_ptr_t)(dst) + kCxWOff + (wave + NW * i) * 1024, 16, 0, 0);
        if (wave == 1)
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(csrc + static_cast<int64_t>(st) * c_step), (lds_ptr_t)(dst) + kCxCOff, 16, 0, 0);
    };

    const int nld = 5 + n_w + (wave == 1 ? 1 : 0);                      // load instructions this wave issues per stage

    f32x16 acc[3];
#pragma unroll
    for (int nb = 0; nb < 3; ++nb) acc[nb] = f32x16{};

    // u stores: lane -> (row lane >> 3 of every group of 8, 16-byte piece lane & 7) of the transposed tile
    const int64_t u_pitch = p.u_l_stride * 2;
    unsigned char *ust = reinterpret_cast<unsigned char *>(p.u) + static_cast<int64_t>(b) * p.u_batch_stride * 2 + (t0 + (lane >> 3)) * u_pitch + (lane & 7) * 16;
    const unsigned u_wr = wave * kCxXBytes + j * 128 + ((kh ^ ((j >> 1) & 7)) << 4);                     // row j, slot ((ks << 1) | kh) ^ swizzle
    // read back rows i * 8 + (lane >> 3), piece lane & 7: (row >> 1) & 7 == (lane >> 4) ^ ((i & 1) << 2), the i part is applied at the read
    const unsigned u_rd = wave * kCxXBytes + (lane >> 3) * 128 + (((lane & 7) ^ (lane >> 4)) << 4);
    unsigned x_off[4];                                                 // this wave's rows j .. j+3, piece kh, swizzled
#pragma unroll
    for (int s = 0; s < 4; ++s) x_off[s] = wave * kCxXBytes + (j + s) * 128 + ((kh ^ (((j + s) >> 1) & 7)) << 4);
    const unsigned w_off = kCxWOff + j * 128 + ((kh ^ ((j >> 1) & 7)) << 4);      // rows nb * 32 + j: (row >> 1) & 7 == (j >> 1) & 7
    const bool first_tile = t0 == 0;                                   // wave-uniform: the causal window starts inside this tile

#pragma unroll
    for (int s = 0; s < NST - 1; ++s)
        if (s < n_stages) issue(s);

#pragma unroll 1
    for (int st = 0; st < n_stages; ++st) {
        // stage st has landed (this wave's loads: vmcnt; every wave's: barrier); stage st - 1 is consumed by every wave
        // VM_CNT retires in issue order; what may stay in flight behind stage st's loads: the loads of the younger stages and the
        // u stores (4 per stage) issued after them
        {
            const int st_stores = (dbg & 4) ? 0 : 4 * (st < NST - 1 ? st : NST - 1);
            const int younger_loads = n_stages - 1 - st < NST - 2 ? n_stages - 1 - st : NST - 2;
            wait_vm_n(st_stores + nld * younger_loads);
        }
        __builtin_amdgcn_s_barrier();
        if (st + NST - 1 < n_stages) issue(st + NST - 1);
        const unsigned sb = smem_lds + (st % NST) * kCxStage;
        // byte addresses of this lane's pieces at k-step 0; k-step ks is ^ (ks << 5): the 16-byte slot is ((ks << 1) | kh) ^ swizzle
        auto reads = [&](KStep &k, const int ks) {
#pragma unroll
            for (int s = 0; s < 4; ++s) lds_rd(k.X[s], (sb + x_off[s]) ^ (ks << 5));
#pragma unroll
            for (int r = 0; r < 4; ++r) lds_rd(k.Wc[r], sb + kCxCOff + (ks * 2 + kh) * 64 + r * 16);
            lds_rd(k.Bc, sb + kCxCOff + 512 + (ks * 2 + kh) * 16);
#pragma unroll
            for (int nb = 0; nb < 3; ++nb) lds_rd(k.Bf[nb], (sb + w_off + nb * 32 * 128) ^ (ks << 5));
        };
        KStep kb[2];
        u32x4 uq[kCxBK / 16];
        reads(kb[0], 0);
#pragma unroll
        for (int ks = 0; ks < kCxBK / 16; ++ks) {
            KStep &k = kb[ks & 1];
            if (ks + 1 < kCxBK / 16) {
                reads(kb[(ks + 1) & 1], ks + 1);
                settle<12>(k);
            } else {
                settle<0>(k);
            }
            if (first_tile) {
#pragma unroll
                for (int s = 0; s < 3; ++s)
                    if (j + s < kCxHalo) k.X[s] = u32x4{0, 0, 0, 0};
            }
            const unsigned xs[4][4] = {{k.X[0].x, k.X[0].y, k.X[0].z, k.X[0].w}, {k.X[1].x, k.X[1].y, k.X[1].z, k.X[1].w},
                                       {k.X[2].x, k.X[2].y, k.X[2].z, k.X[2].w}, {k.X[3].x, k.X[3].y, k.X[3].z, k.X[3].w}};
            const unsigned bs[4] = {k.Bc.x, k.Bc.y, k.Bc.z, k.Bc.w};
            unsigned ur[4] = {xs[3][0], xs[3][1], xs[3][2], xs[3][3]};
            if (!(dbg & 8))
#pragma unroll
            for (int r = 0; r < 4; ++r) {                              // channels 2r (low halves) and 2r + 1 (high halves)
                const unsigned lo01 = __builtin_amdgcn_perm(xs[1][r], xs[0][r], 0x05040100u);
                const unsigned lo23 = __builtin_amdgcn_perm(xs[3][r], xs[2][r], 0x05040100u);
                const unsigned hi01 = __builtin_amdgcn_perm(xs[1][r], xs[0][r], 0x07060302u);
                const unsigned hi23 = __builtin_amdgcn_perm(xs[3][r], xs[2][r], 0x07060302u);
                float a_lo = dot2(lo01, k.Wc[r].x, bf_lo(bs[r]));
                a_lo = dot2(lo23, k.Wc[r].y, a_lo);
                float a_hi = dot2(hi01, k.Wc[r].z, bf_hi(bs[r]));
                a_hi = dot2(hi23, k.Wc[r].w, a_hi);
                ur[r] = pack_bf2(silu(a_lo), silu(a_hi));
            }
            const u32x4 u8 = {ur[0], ur[1], ur[2], ur[3]};
            uq[ks] = u8;
            const bf16x8 a = __builtin_bit_cast(bf16x8, u8);
#pragma unroll
            for (int nb = 0; nb < 3; ++nb)
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, k.Bf[nb]), a, acc[nb], 0, 0, 0);
        }
        // u of this stage (32 positions x 64 channels per wave) leaves as FULL 128-byte lines: a lane holds 16-byte pieces of its own
        // row only (two lanes = 32 contiguous bytes per store instruction and row — measured: the partial-line stores cost more than
        // the loads), so the wave transposes through its own input rows of this stage (every read of them has been settled above; the
        // region is wave-private and the LDS queue of a wave is in order: no barrier).
        if (!(dbg & 4)) {
#pragma unroll
            for (int ks = 0; ks < kCxBK / 16; ++ks)
                asm volatile("ds_write_b128 %0, %1" ::"v"((sb + u_wr) ^ (ks << 5)), "v"(uq[ks]) : "memory");
            u32x4 t[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) lds_rd(t[i], (sb + u_rd + i * 1024) ^ ((i & 1) << 6));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]));
#pragma unroll
            for (int i = 0; i < 4; ++i) *reinterpret_cast<u32x4 *>(ust + static_cast<int64_t>(i * 8) * u_pitch + st * (kCxBK * 2)) = t[i];
        }
    }
    // ---- x_dbl tile of this wave.  The products were taken transposed (W_x rows as the MFMA A operand): D[n][position], a lane holds
    // position j and the outputs n = nb * 32 + (r & 3) + 8 (r >> 2) + 4 kh — 4 consecutive n per register group, one 8-byte piece.
    // The tile goes through LDS (32 rows x 208 B, wave-private; the stage ring is free behind the barrier) and leaves as 16-byte
    // row pieces; with DT it is also the A operand of the dt_proj product below.
    __builtin_amdgcn_s_barrier();
    constexpr int kPitch = 208;
    const unsigned tile = smem_lds + wave * 8192;
#pragma unroll
    for (int nb = 0; nb < 3; ++nb)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            const u32x2 pk = {pack_bf2(acc[nb][q * 4], acc[nb][q * 4 + 1]), pack_bf2(acc[nb][q * 4 + 2], acc[nb][q * 4 + 3])};
            asm volatile("ds_write_b64 %0, %1" ::"v"(tile + j * kPitch + nb * 64 + q * 16 + kh * 8), "v"(pk) : "memory");
        }
    {
        const int pc = lane & 15, r4 = lane >> 4;                     // 4 rows x 16 pieces per instruction
        unsigned char *ob = reinterpret_cast<unsigned char *>(p.out) + (m0 + r4) * p.out_row_stride * 2 + pc * 16;
        u32x4 t[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) lds_rd(t[i], tile + (i * 4 + r4) * kPitch + pc * 16);
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(t[0]), "+v"(t[1]), "+v"(t[2]), "+v"(t[3]), "+v"(t[4]), "+v"(t[5]), "+v"(t[6]), "+v"(t[7]));
        if (pc * 8 < p.n) {
#pragma unroll
            for (int i = 0; i < 8; ++i) *reinterpret_cast<u32x4 *>(ob + static_cast<int64_t>(i * 4) * p.out_row_stride * 2) = t[i];
        }
    }
    if (!DT) return;
    // ---- delta = softplus(x_dbl[:, :dt_rank] @ W_dt^T + bias) for these 32 positions, all d_inner channels (dt_proj.hip's wave tile:
    // even / odd channel B fragments, a lane ends up with channels 2j, 2j+1 of a position: packed 4-byte stores, 128 B per row) ------
    {
        const int R = p.dt_rank;
        bf16x8 af[3];
#pragma unroll
        for (int s = 0; s < 3; ++s) {
            const int k0 = s * 16 + kh * 8;
            u32x4 v;
            lds_rd(v, tile + j * kPitch + (k0 < R ? k0 : 0) * 2);
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(v));
            af[s] = __builtin_bit_cast(bf16x8, k0 < R ? v : u32x4{0, 0, 0, 0});
        }
        const uint16_t *ww = reinterpret_cast<const uint16_t *>(p.dt_w);
        const float *bias = reinterpret_cast<const float *>(p.dt_bias);
        auto wfrags = [&](int d0, bf16x8 (&be)[3], bf16x8 (&bo)[3], float &b_e, float &b_o) {
            const uint16_t *we = ww + static_cast<int64_t>(d0 + 2 * j) * p.dt_w_row_stride;
            const uint16_t *wo = we + p.dt_w_row_stride;
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                const int k0 = s * 16 + kh * 8;
                const uint4 ve = *reinterpret_cast<const uint4 *>(we + (k0 < R ? k0 : 0)), vo = *reinterpret_cast<const uint4 *>(wo + (k0 < R ? k0 : 0));
                be[s] = __builtin_bit_cast(bf16x8, k0 < R ? ve : make_uint4(0, 0, 0, 0));
                bo[s] = __builtin_bit_cast(bf16x8, k0 < R ? vo : make_uint4(0, 0, 0, 0));
            }
            b_e = bias ? bias[d0 + 2 * j] : 0.f;
            b_o = bias ? bias[d0 + 2 * j + 1] : 0.f;
        };
        bf16x8 be[3], bo[3], be_n[3], bo_n[3];
        float b_e, b_o, b_en = 0.f, b_on = 0.f;
        wfrags(0, be, bo, b_e, b_o);
        uint16_t *orow = reinterpret_cast<uint16_t *>(p.delta) + (m0 + 4 * kh) * p.delta_row_stride + 2 * j;
#pragma unroll 1
        for (int d0 = 0; d0 < p.dim; d0 += 64) {
            if (d0 + 64 < p.dim) wfrags(d0 + 64, be_n, bo_n, b_en, b_on);   // requested before this block's softplus / stores
            f32x16 ce = {}, co = {};
#pragma unroll
            for (int s = 0; s < 3; ++s) {
                ce = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s], be[s], ce, 0, 0, 0);
                co = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[s], bo[s], co, 0, 0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dm = (r & 3) + 8 * (r >> 2);
                float ve = ce[r] + b_e, vo = co[r] + b_o;
                if (p.dt_softplus) { ve = softplus20_r16(ve); vo = softplus20_r16(vo); }
                *reinterpret_cast<uint32_t *>(orow + dm * p.delta_row_stride + d0) = pack_bf2(ve, vo);
            }
#pragma unroll
            for (int s = 0; s < 3; ++s) { be[s] = be_n[s]; bo[s] = bo_n[s]; }
            b_e = b_en; b_o = b_on;
        }
    }
}

}  // namespace zigma

using namespace zigma;

extern "C" int zigma_conv_x_proj_fwd(const zigma_conv_xproj_params_t *pp, void *stream_) {
    if (!pp) return ZIGMA_ERR_NULL;
    (void)hipGetLastError();
    const zigma_conv_xproj_params_t &p = *pp;
    if (p.batch < 0 || p.seqlen < 0 || p.dim < 1 || p.n < 1) return ZIGMA_ERR_SHAPE;
    if (p.flags & ~15) return ZIGMA_ERR_UNSUPPORTED;
    if (p.batch == 0 || p.seqlen == 0) return ZIGMA_OK;
    if (!p.x || !p.conv_weight || !p.conv_bias || !p.w || !p.u || !p.out) return ZIGMA_ERR_NULL;
    if (p.dtype != ZIGMA_BF16) return ZIGMA_ERR_DTYPE;
    if (p.n > 96 || p.n % 8 != 0 || p.dim % kCxBK != 0 || p.seqlen % kCxTok != 0) return ZIGMA_ERR_SHAPE;
    if (p.delta) {                       // optional third product: delta = softplus(x_dbl[:, :dt_rank] @ dt_w^T + dt_bias)
        if (!p.dt_w) return ZIGMA_ERR_NULL;
        if (p.dt_rank < 8 || p.dt_rank > 48 || p.dt_rank % 8 != 0 || p.dt_rank > p.n) return ZIGMA_ERR_SHAPE;
        if (p.dt_w_row_stride % 8 != 0 || p.delta_row_stride % 2 != 0 || reinterpret_cast<uintptr_t>(p.dt_w) % 16 != 0 ||
            reinterpret_cast<uintptr_t>(p.delta) % 4 != 0)
            return ZIGMA_ERR_STRIDE;
    }
    if (p.out_row_stride % 8 != 0 || reinterpret_cast<uintptr_t>(p.out) % 16 != 0) return ZIGMA_ERR_STRIDE;
    const int64_t m = static_cast<int64_t>(p.batch) * p.seqlen;
    if (m % (kCxTok * 8) != 0) return ZIGMA_ERR_SHAPE;       // (either workgroup size)
    auto al16 = [](const void *q) { return reinterpret_cast<uintptr_t>(q) % 16 == 0; };
    if (p.x_l_stride % 8 != 0 || p.x_batch_stride % 8 != 0 || p.u_l_stride % 8 != 0 || p.u_batch_stride % 8 != 0 || p.w_row_stride % 8 != 0 ||
        !al16(p.x) || !al16(p.u) || !al16(p.w) || !al16(p.conv_weight) || !al16(p.conv_bias))
        return ZIGMA_ERR_STRIDE;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    // default: 4-wave workgroups (128 positions), two stages = 66 KB of LDS: two workgroups per CU that drift apart, one computing
    // while the other waits for its loads (measured 72 us; 8 waves in lockstep 75 us; a third stage does not pay, the second
    // workgroup does its job).  flags: 1 = three stages, 2 = eight-wave workgroups; probes (wrong results): 4 = no u stores,
    // 8 = no conv arithmetic.
    const bool dt = p.delta != nullptr;
#define ZIGMA_CX(S_, W_)                                                                                        \
    if (dt) hipLaunchKernelGGL((conv_x_proj_kernel<S_, W_, true>), grid, block, 0, stream, p);                  \
    else hipLaunchKernelGGL((conv_x_proj_kernel<S_, W_, false>), grid, block, 0, stream, p);
    if (p.flags & 2) {
        const dim3 grid(static_cast<unsigned>(m / (kCxTok * 8))), block(64 * 8);
        if (p.flags & 1) { ZIGMA_CX(3, 8) } else { ZIGMA_CX(2, 8) }
    } else {
        const dim3 grid(static_cast<unsigned>(m / (kCxTok * 4))), block(64 * 4);
        if (p.flags & 1) { ZIGMA_CX(3, 4) } else { ZIGMA_CX(2, 4) }
    }
#undef ZIGMA_CX
    set_last_kernel(dt ? "conv_x_proj_dt_mfma" : "conv_x_proj_mfma");
    return check_launch();
}
