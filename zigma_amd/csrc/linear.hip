// Dense projections of the ZigMa block on the CDNA4 matrix cores: out = x @ W^T (+ bias) (+ SiLU on a column range), bf16.
// C ABI: zigma_linear_fwd.
//
// Replaces the cuBLAS GEMMs behind F.linear at the reference's call sites Mamba.in_proj (mamba_simple.py:290-294),
// out_proj (selective_scan_interface.py:365) and CrossAttention.to_q / to_out (model_zigma.py:104-135).  Shapes at the
// headline config (M = B*L = 65 536 tokens): in_proj 640 -> 2560, out_proj 1280 -> 640, to_q 640 -> 512, to_out 512 -> 640:
// K is SHORT (8 - 20 k-steps of 64) and the output is as large as the input, so a tile's prologue / epilogue and the C
// write-back matter as much as the main loop.  Design (MI355X: 256 CUs, 160 KB LDS, v_mfma_f32_32x32x16_bf16):
//
//   * "TN": both operands are K-contiguous (activations token-major, nn.Linear weights (N, K)): every MFMA fragment is one
//     16-byte piece of a row.  The product is evaluated TRANSPOSED, D[n][m] = sum_k W[n][k] x[m][k] (W rows as the MFMA
//     A operand, tokens as the B operand): a lane then holds 4 CONSECUTIVE output features of ONE token per accumulator
//     quad, so results leave as packed 8-byte pieces of an output row (row-major out, no transpose pass).
//   * workgroup tile 256 tokens x BN features (BN = 256: 8 waves as 4 (N) x 2 (M), wave tile 64 x 128; BN = 128: 2 x 4, wave
//     tile 64 x 64), BK = 64, one workgroup per CU, persistent: a workgroup walks its list of tiles and the k-steps of all
//     of them as ONE software pipeline — the first k-step of the next tile is already in flight while the epilogue of
//     the current tile stores.
//   * operands reach LDS by direct-to-LDS loads (global_load_lds_dwordx4, 1 KB per wave instruction, no VGPR round trip,
//     no ds_write), two stages of (BN + 256) rows x 128 B; ONE barrier per k-step.  The LDS image is lane-linear, so the
//     bank swizzle (16-byte slot ^= (row >> 1) & 7: conflict-free ds_read_b128 for 32 consecutive rows) is applied to the
//     per-lane SOURCE address and again on the fragment reads (cdna_hip_programming.md T2 / rule 21).
//   * tile order is XCD-aware: the 32 workgroups of an XCD (blockIdx % 8) work at any time on ~32 consecutive tiles of the
//     (m-tile, n-tile) raster, i.e. a few 256-token activation panels x all weight panels: both stay in that XCD's 4 MB L2.
//   * epilogue in registers: + bias, SiLU on output columns >= silu_from_col (in_proj emits silu(z) for the gate half:
//     the scan then multiplies instead of evaluating exp + rcp per element, ZIGMA_SCAN_Z_PREACTIVATED), bf16 pack.
#include "scan_helpers.h"

namespace zigma {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) unsigned char *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *gbl_ptr_t;

constexpr int kLinBM = 256, kLinBK = 64;

// Epilogue of one wave tile (64 features x MB * 32 tokens), shared by both pipelines.
// D[i][jj]: jj = token (lane & 31), i = (reg & 3) + 8 (reg >> 2) + 4 (lane >> 5) = output feature: a lane holds 8-byte pieces of
// 32 different output rows, and storing them as they lie costs one memory request per row per instruction (measured: the
// scattered form made the stores the longest phase of the kernel).  So the wave transposes its tile through a private 4 KB
// of LDS, 32 tokens x 64 features at a time: + bias (fp32, before the single rounding), SiLU on whole 32-feature blocks,
// packed 8-byte writes (16-byte slot ^= (token >> 1) & 7: conflict-free), 16-byte reads; every global store instruction then
// covers 8 tokens x 128 contiguous bytes.  `s_bias`: the bias vector staged in LDS as bf16 (or nullptr).
// RES: out = residual + gate * bf16(value) (CrossAttention's `hidden + gate_msa * to_out(...)`, reference model_zigma.py:447-449; the
// projection result is rounded to bf16 first, as the reference's bf16 tensor is): the residual rows are fetched with the store
// pattern (16 bytes per lane, 8 tokens x 128 B per instruction), `gate8` = this lane's 8 gate values (bf16).
template <int MB, int NB, bool RES = false>
__device__ __forceinline__ void linear_epilogue(const f32x16 (&acc)[NB][MB], unsigned char *scr, const uint16_t *s_bias,
                                                const rsrc_t o_rs, const int64_t o_pitch, const int n_wave0, const int silu_from_col,
                                                const int lane, const rsrc_t r_rs, const int64_t r_pitch, const uint4 gate8) {
    const int j = lane & 31, kh = lane >> 5;
    const int wr_off = j * 128 + kh * 8, wr_sw = (j >> 1) & 7;                               // this lane's row in the scratch tile
    const int rd_tok = lane >> 3, rd_slot = (lane & 7) ^ (rd_tok >> 1);                      // row = i * 8 + rd_tok: swizzle (row >> 1) & 7
    const unsigned st_off = static_cast<unsigned>(rd_tok * o_pitch + (lane & 7) * 16);
    const unsigned scr_lds = static_cast<unsigned>(reinterpret_cast<uintptr_t>((lds_ptr_t)(scr)));      // LDS byte address
#pragma unroll
    for (int mb = 0; mb < MB; ++mb) {
        typedef unsigned u4 __attribute__((ext_vector_type(4)));
        u4 res[4];
        if constexpr (RES) {                          // requested first: the latency runs under the conversion / transposition below
            const unsigned rs_off = static_cast<unsigned>(rd_tok * r_pitch + (lane & 7) * 16);
#pragma unroll
            for (int i = 0; i < 4; ++i) res[i] = __builtin_amdgcn_raw_buffer_load_b128(r_rs, rs_off, static_cast<int>((mb * 32 + i * 8) * r_pitch), 0);
        }
#pragma unroll
        for (int nb = 0; nb < NB; ++nb) {
            const int n0 = n_wave0 + nb * 32;                                                  // wave-uniform
            float v[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) v[r] = acc[nb][mb][r];
            if (s_bias) {
                // (inline asm like the writes below: a visible ds_read would make hipcc drain the load ring first)
                typedef unsigned u2 __attribute__((ext_vector_type(2)));
                const unsigned b_lds = static_cast<unsigned>(reinterpret_cast<uintptr_t>((lds_ptr_t)(const_cast<uint16_t *>(s_bias))));
                u2 bq[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) asm volatile("ds_read_b64 %0, %1" : "=v"(bq[q]) : "v"(b_lds + (n0 + 4 * kh + q * 8) * 2));
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3]));
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    v[q * 4 + 0] += __uint_as_float(bq[q].x << 16);
                    v[q * 4 + 1] += __uint_as_float(bq[q].x & 0xffff0000u);
                    v[q * 4 + 2] += __uint_as_float(bq[q].y << 16);
                    v[q * 4 + 3] += __uint_as_float(bq[q].y & 0xffff0000u);
                }
            }
            if (n0 >= silu_from_col) {                                                         // (silu_from_col % 32 == 0)
#pragma unroll
                for (int r = 0; r < 16; ++r) v[r] = silu(v[r]);
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                uint2 pk;
                pk.x = static_cast<uint32_t>(from_float<BF16>(v[q * 4])) | (static_cast<uint32_t>(from_float<BF16>(v[q * 4 + 1])) << 16);
                pk.y = static_cast<uint32_t>(from_float<BF16>(v[q * 4 + 2])) | (static_cast<uint32_t>(from_float<BF16>(v[q * 4 + 3])) << 16);
                // (inline asm: a ds_write the compiler can see makes it drain vmcnt — i.e. the whole load ring — first,
                // because an LDS-DMA in flight might target the same bytes; it cannot: the scratch is outside the ring)
                typedef unsigned u2 __attribute__((ext_vector_type(2)));
                const u2 pkv = {pk.x, pk.y};
                asm volatile("ds_write_b64 %0, %1" ::"v"(scr_lds + wr_off + (((nb * 4 + q) ^ wr_sw) << 4)), "v"(pkv) : "memory");
            }
        }
        // wave-private scratch: writes and reads of one wave execute in order in the LDS queue; the reads' data is waited for
        // explicitly (inline asm again: no drain of the load ring in front of them)
        u4 row[4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
            asm volatile("ds_read_b128 %0, %1" : "=v"(row[i]) : "v"(scr_lds + i * 1024 + rd_tok * 128 + ((rd_slot ^ ((i & 1) << 2)) << 4)));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(row[0]), "+v"(row[1]), "+v"(row[2]), "+v"(row[3]));
        if constexpr (RES) {
            const unsigned gq[4] = {gate8.x, gate8.y, gate8.z, gate8.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float lo = __builtin_fmaf(__uint_as_float(gq[e] << 16), __uint_as_float(row[i][e] << 16), __uint_as_float(res[i][e] << 16));
                    const float hi = __builtin_fmaf(__uint_as_float(gq[e] & 0xffff0000u), __uint_as_float(row[i][e] & 0xffff0000u),
                                                    __uint_as_float(res[i][e] & 0xffff0000u));
                    row[i][e] = static_cast<uint32_t>(from_float<BF16>(lo)) | (static_cast<uint32_t>(from_float<BF16>(hi)) << 16);
                }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
            __builtin_amdgcn_raw_buffer_store_b128(row[i], o_rs, st_off, static_cast<int>((mb * 32 + i * 8) * o_pitch), 0);
    }
}

// s_waitcnt vmcnt(n) for the handful of counts the pipeline uses (the instruction takes an immediate)
__device__ __forceinline__ void wait_vm(int n) {
    switch (n) {
        case 0: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
        case 6: asm volatile("s_waitcnt vmcnt(6)" ::: "memory"); break;
        case 8: asm volatile("s_waitcnt vmcnt(8)" ::: "memory"); break;
        case 14: asm volatile("s_waitcnt vmcnt(14)" ::: "memory"); break;
        case 16: asm volatile("s_waitcnt vmcnt(16)" ::: "memory"); break;
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

// WN_ = waves along the features: 4 -> 256 x 256 tile (wave tile 64 x 128), 2 -> 256 x 128 tile (wave tile 64 x 64).
// NST = LDS stages of one k-step each: the k-steps of ALL tiles of the workgroup form one pipeline, loads run NST - 1 k-steps
// ahead (two stages of 64 KB for the wide tile; the narrow tile's 48 KB stages leave room for a third, which is what makes
// out_proj insensitive to where its operand lives — inside the forward its input was just written by the scan).
// Synchronisation is explicit: ONE raw s_barrier per k-step behind a COUNTED s_waitcnt vmcnt.  VM_CNT retires in issue order
// for loads and stores alike on gfx9-class targets (the compiler's own wait insertion relies on it), so the count is "what
// was issued after the batch this k-step needs": the younger load batch (NST == 3) and, during the first NST - 1 k-steps
// after an epilogue, that epilogue's stores — the store acknowledgements are never waited for on the critical path
// (__syncthreads() would drain them: measured ~2 us per tile).
template <int WN_, int NST, bool HAS_BIAS, bool RES = false>
__global__ __launch_bounds__(512, 2) void linear_tn_kernel(const zigma_linear_params_t p, const int tiles_m, const int tiles_n) {
    constexpr int BM = kLinBM, BN = 64 * WN_, WM_ = 8 / WN_, MB = BM / WM_ / 32, NB = 2;
    constexpr int ROWS = BN + BM, STAGE = ROWS * 128;            // bytes per stage: W rows first, then token rows
    constexpr int NLD = ROWS / 64;                                 // direct-to-LDS loads per wave per stage (8 rows each)
    constexpr int NSTORE = MB * 4;                                 // 16-byte stores per wave per epilogue
    constexpr int BIAS_BYTES = HAS_BIAS ? 8192 : 0;                // n <= 4096
    static_assert(NST * STAGE + BIAS_BYTES <= 160 * 1024, "LDS");
    // ONE LDS object (a second one makes hipcc drain vmcnt in front of every fragment read, cdna_hip_programming.md §5)
    __shared__ __attribute__((aligned(1024))) unsigned char smem[NST * STAGE + BIAS_BYTES];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave % WN_, wm = wave / WN_;
    const int j = lane & 31, kh = lane >> 5;
    const int nk = p.k / kLinBK;
    const unsigned char *xb = reinterpret_cast<const unsigned char *>(p.x);
    const unsigned char *wb = reinterpret_cast<const unsigned char *>(p.w);
    const int64_t x_pitch = p.x_row_stride * 2, w_pitch = p.w_row_stride * 2;
    const int dbg = p.flags;

    // ---- tile schedule: XCD x owns the contiguous raster chunk [x * chunk, (x + 1) * chunk); its workgroups take it round-robin
    const int n_tiles = tiles_m * tiles_n;
    const int xcd = blockIdx.x & 7, slot_in_xcd = blockIdx.x >> 3, wg_per_xcd = gridDim.x >> 3;
    const int chunk = (n_tiles + 7) >> 3;
    const int chunk_end = (xcd + 1) * chunk < n_tiles ? (xcd + 1) * chunk : n_tiles;
    const int tile0 = xcd * chunk + slot_in_xcd;
    if (tile0 >= chunk_end) return;
    const int my_tiles = (chunk_end - tile0 + wg_per_xcd - 1) / wg_per_xcd;
    const int g_total = my_tiles * nk;                             // k-steps of this workgroup's whole run

    // staging pattern.  Direct-to-LDS instruction i of wave w fills the 8 rows q*8 .. q*8+7, q = i * 8 + w, of the stage: rows
    // [0, BN) are W rows, [BN, BN + 256) token rows, so the operand of instruction i is known at compile time (i < BN / 64).
    // Per lane the source address is
    //   operand base (SGPR) + [tile row0 + i' * 64] * pitch + kt * 128                     (wave-uniform: scalar unit)
    //   + (w * 8 + (lane >> 3)) * pitch + piece * 16,  piece = (lane & 7) ^ ((row >> 1) & 7)  (per lane, ONE value per operand)
    // with row = w * 8 + (lane >> 3) (mod 16): the swizzle term does not depend on i.
    const int srow = wave * 8 + (lane >> 3);
    const unsigned piece = ((lane & 7) ^ ((srow >> 1) & 7)) << 4;
    const unsigned lane_off_w = static_cast<unsigned>(srow * w_pitch) + piece, lane_off_x = static_cast<unsigned>(srow * x_pitch) + piece;
    int is_tile = tile0, is_kt = 0, is_g = 0;                      // issue cursor: next (tile, k-step) to fetch, stage is_g % NST
    auto issue = [&]() {
        const int mt = is_tile / tiles_n, nt = is_tile - mt * tiles_n;
        const unsigned char *wbase = wb + static_cast<int64_t>(nt) * BN * w_pitch + is_kt * (kLinBK * 2);
        const unsigned char *xbase = xb + static_cast<int64_t>(mt) * BM * x_pitch + is_kt * (kLinBK * 2);
        const int64_t rows_left = p.m - static_cast<int64_t>(mt) * BM;      // m % 8 == 0: a group of 8 rows exists or does not
        unsigned char *dst = smem + (is_g % NST) * STAGE;
#pragma unroll
        for (int i = 0; i < NLD; ++i) {
            const unsigned char *src;
            if (i < BN / 64) {
                src = wbase + static_cast<int64_t>(i * 64) * w_pitch + lane_off_w;
            } else {
                const int r0 = (i - BN / 64) * 64;
                src = xbase + (r0 + wave * 8 < rows_left ? static_cast<int64_t>(r0) * x_pitch : -static_cast<int64_t>(wave * 8) * x_pitch) + lane_off_x;
            }
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src), (lds_ptr_t)(dst) + (i * 8 + wave) * 1024, 16, 0, 0);
        }
        ++is_g;
        if (++is_kt == nk) { is_kt = 0; is_tile += wg_per_xcd; }
    };
    // the same batch in pieces, for the main loop: the direct-to-LDS instructions cost 60-180 cycles of issue each and, fired back to
    // back at the top of a k-step by every wave, leave the matrix pipe idle for their whole burst; spread between the MFMAs of the
    // first two sub-steps they cost the wave what they cost, and the pipe keeps running on its partner's MFMAs
    const unsigned char *ip_w = nullptr, *ip_x = nullptr;
    unsigned char *ip_dst = nullptr;
    int64_t ip_rows_left = 0;
    auto issue_prep = [&]() {
        const int mt = is_tile / tiles_n, nt = is_tile - mt * tiles_n;
        ip_w = wb + static_cast<int64_t>(nt) * BN * w_pitch + is_kt * (kLinBK * 2);
        ip_x = xb + static_cast<int64_t>(mt) * BM * x_pitch + is_kt * (kLinBK * 2);
        ip_rows_left = p.m - static_cast<int64_t>(mt) * BM;
        ip_dst = smem + (is_g % NST) * STAGE;
        ++is_g;
        if (++is_kt == nk) { is_kt = 0; is_tile += wg_per_xcd; }
    };
    auto issue_piece = [&](const int i) {                          // i: compile-time constant at the call sites
        const unsigned char *src;
        if (i < BN / 64) {
            src = ip_w + static_cast<int64_t>(i * 64) * w_pitch + lane_off_w;
        } else {
            const int r0 = (i - BN / 64) * 64;
            src = ip_x + (r0 + wave * 8 < ip_rows_left ? static_cast<int64_t>(r0) * x_pitch : -static_cast<int64_t>(wave * 8) * x_pitch) + lane_off_x;
        }
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src), (lds_ptr_t)(ip_dst) + (i * 8 + wave) * 1024, 16, 0, 0);
    };

    // fragment read offsets: row * 128 + (((ks << 1) | kh) ^ ((row >> 1) & 7)) * 16, row = base (multiple of 32) + j
    const int sw = (j >> 1) & 7;
    const int a_row0 = (wn * 64 + j) * 128, b_row0 = (BN + wm * (BM / WM_) + j) * 128;

    if (HAS_BIAS) {                                                // plain loads + LDS writes, retired before the pipeline starts counting
        for (int i = tid; i < p.n / 2; i += 512)
            reinterpret_cast<uint32_t *>(smem + NST * STAGE)[i] = reinterpret_cast<const uint32_t *>(p.bias)[i];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
#pragma unroll 1
    for (int g0 = 0; g0 < NST - 1 && g0 < g_total; ++g0) issue();

    int g = 0, tile = tile0;
#pragma unroll 1
    for (int ti = 0; ti < my_tiles; ++ti, tile += wg_per_xcd) {
        f32x16 acc[NB][MB];
#pragma unroll
        for (int nb = 0; nb < NB; ++nb)
#pragma unroll
            for (int mb = 0; mb < MB; ++mb) acc[nb][mb] = f32x16{};
#pragma unroll 1
        for (int kt = 0; kt < nk; ++kt, ++g) {
            // batch g has landed when only what was issued after it is still outstanding: the younger batches (NST - 2 of
            // them, fewer at the very end) and, right after an epilogue, its stores
            const int younger = (g_total - 1 - g) < (NST - 2) ? (g_total - 1 - g) : (NST - 2);
            wait_vm(NLD * younger + ((ti > 0 && kt < NST - 1 && !(dbg & 0x400)) ? NSTORE : 0));
            __builtin_amdgcn_s_barrier();                       // ... for every wave; and every wave is done with stage (g - 1) % NST
            const bool do_issue = g + NST - 1 < g_total && !(dbg & 0x200);
            // SPREAD (the 256 x 128 tile: three stages, loads two k-steps ahead): the pieces of the batch go out between the MFMAs of
            // the first two sub-steps instead of as a burst at the top: out_proj 131 -> 120 us, to_out 61 -> 57.  The 256 x 256 tile
            // (two stages: every cycle of lead counts, 228 instead of 192 registers) measured slower that way (254 vs 241) and keeps the burst.
            constexpr bool SPREAD = WN_ == 2;
            constexpr int PPS = (NLD + 1) / 2;                      // pieces per sub-step, sub-steps 0 and 1
            if (do_issue) {
                if constexpr (SPREAD) issue_prep(); else issue();
            }
            const unsigned char *sb = smem + (g % NST) * STAGE;
            // fragments one k-substep ahead of the MFMAs that use them (the LDS latency hides under the previous 8 MFMAs)
            bf16x8 a[2][NB], b[2][MB];
            auto frags = [&](int ks, bf16x8 (&fa)[NB], bf16x8 (&fb)[MB]) {
                const int off = ((((ks << 1) | kh) ^ sw) << 4);
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
                    fa[nb] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(sb + a_row0 + nb * 32 * 128 + off));
#pragma unroll
                for (int mb = 0; mb < MB; ++mb)
                    fb[mb] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(sb + b_row0 + mb * 32 * 128 + off));
            };
            if (dbg & 0x100) continue;
            frags(0, a[0], b[0]);
#pragma unroll
            for (int ks = 0; ks < kLinBK / 16; ++ks) {
                if (ks + 1 < kLinBK / 16) frags(ks + 1, a[(ks + 1) & 1], b[(ks + 1) & 1]);
                if constexpr (SPREAD) {
                    if (ks < 2 && do_issue) {
#pragma unroll
                        for (int i = ks * PPS; i < (ks + 1) * PPS && i < NLD; ++i) issue_piece(i);
                    }
                }
#pragma unroll
                for (int nb = 0; nb < NB; ++nb)
#pragma unroll
                    for (int mb = 0; mb < MB; ++mb)
                        acc[nb][mb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ks & 1][nb], b[ks & 1][mb], acc[nb][mb], 0, 0, 0);
            }
            // Pin the order hipcc would otherwise collapse to [all reads of a sub-step -> wait -> its MFMAs]: the NB + MB reads of
            // sub-step ks + 1 are issued one by one between the first MFMAs of sub-step ks (masks: 0x100 DS read, 0x008 MFMA).
            {
                __builtin_amdgcn_sched_group_barrier(0x100, NB + MB, 0);
#pragma unroll
                for (int ks = 0; ks + 1 < kLinBK / 16; ++ks) {
#pragma unroll
                    for (int r = 0; r < (NB + MB < NB * MB ? NB + MB : NB * MB); ++r) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                    if (NB * MB > NB + MB) __builtin_amdgcn_sched_group_barrier(0x008, NB * MB - (NB + MB), 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, NB * MB, 0);
            }
        }
        // ---- epilogue: transposed through the stage the main loop has just finished with (the next issue into it happens
        // behind the next k-step's barrier) ---------------------------------------------------------------------------------
        if (!(dbg & 0x400)) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();                                                     // every wave is done with the fragments of this stage
            const int mt = tile / tiles_n, nt = tile - mt * tiles_n;
            const int64_t m_tile = static_cast<int64_t>(mt) * BM + wm * (BM / WM_);           // first token of this wave's tile
            const int64_t rows_here = p.m - m_tile;                                            // tokens of it that exist
            const int64_t o_pitch = p.out_row_stride * 2;
            // rows as buffer descriptor (base = the wave tile's first row and first column, extent = its valid rows): stores of
            // tokens beyond m fall outside the extent and are dropped by the hardware
            const rsrc_t o_rs = make_rsrc(reinterpret_cast<unsigned char *>(p.out) + m_tile * o_pitch + (nt * BN + wn * 64) * 2,
                                          rows_here > 0 ? (rows_here < BM / WM_ ? rows_here : BM / WM_) * o_pitch - (nt * BN + wn * 64) * 2 : 0);
            if constexpr (RES) {
                // the wave tile (BM / WM_ tokens) lies inside one sample (rows_per_batch % 256 == 0): one gate row, 8 values per lane
                const int64_t r_pitch = p.res_row_stride * 2;
                const int n_w = nt * BN + wn * 64;
                const rsrc_t r_rs = make_rsrc(reinterpret_cast<const unsigned char *>(p.residual) + m_tile * r_pitch + n_w * 2,
                                              rows_here > 0 ? (rows_here < BM / WM_ ? rows_here : BM / WM_) * r_pitch - n_w * 2 : 0);
                const int64_t bsm = m_tile / p.rows_per_batch;
                const uint4 gate8 = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint16_t *>(p.gate) + bsm * p.gate_batch_stride + n_w + (lane & 7) * 8);
                linear_epilogue<MB, NB, true>(acc, smem + ((g - 1) % NST) * STAGE + wave * 4096,
                                              HAS_BIAS ? reinterpret_cast<const uint16_t *>(smem + NST * STAGE) : nullptr, o_rs, o_pitch, n_w,
                                              p.silu_from_col, lane, r_rs, r_pitch, gate8);
            } else {
                linear_epilogue<MB, NB>(acc, smem + ((g - 1) % NST) * STAGE + wave * 4096,
                                        HAS_BIAS ? reinterpret_cast<const uint16_t *>(smem + NST * STAGE) : nullptr, o_rs, o_pitch, nt * BN + wn * 64,
                                        p.silu_from_col, lane, o_rs, 0, uint4{});      // (no residual: the last three are not read)
            }
        }
    }
}

// csrc/linear4w.hip: the one-wave-per-SIMD kernel for the wide, epilogue-free projections (in_proj, to_q)
bool linear4w_eligible(const zigma_linear_params_t &p);
int launch_linear4w(const zigma_linear_params_t &p, hipStream_t stream);
// csrc/linear_ws.hip: weights stationary in registers, only the tokens stream (k <= 640, n % 256 == 0)
bool linear_ws_eligible(const zigma_linear_params_t &p);
int launch_linear_ws(const zigma_linear_params_t &p, hipStream_t stream);
// csrc/linear_sm.hip: few tokens — tiles of 128 tokens x n / 4 features (n % 160 == 0 or n % 192 == 0), one per workgroup
bool linear_sm_eligible(const zigma_linear_params_t &p);
int launch_linear_sm(const zigma_linear_params_t &p, hipStream_t stream);

}  // namespace zigma

using namespace zigma;

extern "C" int zigma_linear_fwd(const zigma_linear_params_t *pp, void *stream_) {
    if (!pp) return ZIGMA_ERR_NULL;
    (void)hipGetLastError();
    const zigma_linear_params_t &p = *pp;
    if (p.m < 0 || p.n < 1 || p.k < 1) return ZIGMA_ERR_SHAPE;
    if (p.flags & ~0xf7ff00) return ZIGMA_ERR_UNSUPPORTED;     // 0x4000: the weight-stationary kernel; 0x8000: the few-token kernel; 0x100 ... 0x1000: timing / A-B probes (tools/linear_probe.py); 0x2000: the 8-wave kernel; 0x10000 .. 0x50000: probes of the 4-wave kernel (probe builds only)
    if (p.m == 0) return ZIGMA_OK;
    if (!p.x || !p.w || !p.out) return ZIGMA_ERR_NULL;
    if (p.dtype != ZIGMA_BF16) return ZIGMA_ERR_DTYPE;
    if (p.k % kLinBK != 0 || p.n % 128 != 0 || p.m % 8 != 0) return ZIGMA_ERR_SHAPE;
    if (p.m * p.x_row_stride * 2 > 0x7fffffff || static_cast<int64_t>(p.n) * p.w_row_stride * 2 > 0x7fffffff ||
        256 * p.out_row_stride * 2 > 0x7fffffff)
        return ZIGMA_ERR_SHAPE;                       // 32-bit lane offsets inside an operand tile / a wave's output rows
    if (p.silu_from_col < 0 || p.silu_from_col % 32 != 0) return ZIGMA_ERR_SHAPE;
    if (p.x_row_stride % 8 != 0 || p.w_row_stride % 8 != 0 || p.out_row_stride % 4 != 0 ||
        (reinterpret_cast<uintptr_t>(p.x) | reinterpret_cast<uintptr_t>(p.w)) % 16 != 0 || reinterpret_cast<uintptr_t>(p.out) % 8 != 0)
        return ZIGMA_ERR_STRIDE;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int tiles_m = static_cast<int>((p.m + kLinBM - 1) / kLinBM);
    if (p.residual) {                     // gated residual epilogue
        if (!p.gate || p.rows_per_batch < 1 || p.rows_per_batch % 256 != 0 || p.m % p.rows_per_batch != 0) return ZIGMA_ERR_SHAPE;
        if (p.res_row_stride % 8 != 0 || p.gate_batch_stride % 8 != 0 || reinterpret_cast<uintptr_t>(p.residual) % 16 != 0 ||
            reinterpret_cast<uintptr_t>(p.gate) % 16 != 0 || 256 * p.res_row_stride * 2 > 0x7fffffff)
            return ZIGMA_ERR_STRIDE;
    }
    if (p.bias && (p.n > 4096 || reinterpret_cast<uintptr_t>(p.bias) % 4 != 0)) return ZIGMA_ERR_SHAPE;
    if (p.flags & 0x4000) return linear_ws_eligible(p) ? launch_linear_ws(p, stream) : ZIGMA_ERR_UNSUPPORTED;
    if (p.flags & 0x8000) return linear_sm_eligible(p) ? launch_linear_sm(p, stream) : ZIGMA_ERR_UNSUPPORTED;
    if (linear4w_eligible(p)) return launch_linear4w(p, stream);        // (needs flags == 0: any probe flag pins the 8-wave kernel)
    const bool wide = p.n % 256 == 0 && !(p.flags & 0x1000) && !p.residual;      // 0x1000: force the 256 x 128 tile (probe)
    const int tiles_n = p.n / (wide ? 256 : 128);
    const int64_t n_tiles = static_cast<int64_t>(tiles_m) * tiles_n;
    if (n_tiles > 0x7fffffff) return ZIGMA_ERR_SHAPE;
    int grid = 256;                                  // one persistent workgroup per CU; multiples of 8 keep the XCD map
    if (n_tiles < grid) grid = static_cast<int>((n_tiles + 7) / 8 * 8);
#define ZIGMA_LIN(W_, S_, B_) hipLaunchKernelGGL((linear_tn_kernel<W_, S_, B_>), dim3(grid), dim3(512), 0, stream, p, tiles_m, tiles_n)
    if (wide) { if (p.bias) ZIGMA_LIN(4, 2, true); else ZIGMA_LIN(4, 2, false); }
    else if (p.flags & 0x800) { if (p.bias) ZIGMA_LIN(2, 2, true); else ZIGMA_LIN(2, 2, false); }      // 0x800: two stages (probe)
    else if (p.residual) {
        if (p.bias) hipLaunchKernelGGL((linear_tn_kernel<2, 3, true, true>), dim3(grid), dim3(512), 0, stream, p, tiles_m, tiles_n);
        else hipLaunchKernelGGL((linear_tn_kernel<2, 3, false, true>), dim3(grid), dim3(512), 0, stream, p, tiles_m, tiles_n);
    }
    else { if (p.bias) ZIGMA_LIN(2, 3, true); else ZIGMA_LIN(2, 3, false); }
#undef ZIGMA_LIN
    set_last_kernel(wide ? "linear_tn_256x256" : "linear_tn_256x128");
    return check_launch();
}
