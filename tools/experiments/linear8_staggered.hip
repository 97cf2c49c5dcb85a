// Dense projection out = x @ w^T (+ bias) (+ SiLU on a column range) on the matrix cores, gfx950 — the staggered schedule.
// Entered through zigma_linear_fwd (linear.hip) for m % 256 == 0; same ABI, same results layout as linear_tn_kernel.
//
// What the one-barrier-per-k-step kernel of linear.hip cannot do: its eight waves read fragments and run MFMAs at the same
// moments, so the two waves that share a SIMD leave the matrix pipe idle together (ds_read + MFMA alone, no global traffic:
// 58 % of the MFMA rate on the in_proj shape, profiles/r02_linear_probe.jsonl).  Here the waves of a workgroup form two groups
// (waves 0-3 / 4-7 = the two waves of every SIMD) that run the SAME phase sequence one barrier apart:
//
//     group 0:  L0 | M0 | L1 | M1 | L2 | ...            L = wait for landed data, issue the prefetch, ds_read a register subtile
//     group 1:     | L0 | M0 | L1 | M1 | ...            M = 8 x v_mfma_f32_32x32x16_bf16 on that subtile      | = s_barrier
//
// so that in every interval one wave of each SIMD is in its MFMA section while its partner fetches.
//
// Tile 256 tokens x 256 features, BK = 64; wave (wm, wn) = 128 tokens x 64 features (fp32 accumulators: 128 registers).
// A K-tile is consumed as four QUADRANT phases of the wave tile, (token half th, feature half fh) = (0,0) (0,1) (1,1) (1,0):
// a phase needs 64 tokens x 64 k of x (8 ds_read_b128) and/or 32 features x 64 k of w (4), kept in registers across the two
// phases that share them.  LDS holds two K-tiles as eight 16 KB HALF-TILES (X0 X1: token half th of both wm; W0 W1: feature half
// fh of all four wn; 128 rows x 128 B, 16-byte slots swizzled by (row >> 1) & 7).  One half-tile is restaged per phase (two
// global_load_lds_dwordx4 per wave), six phases ahead of its first read and two or more phases after the last read of the
// half-tile it replaces; a counted s_waitcnt vmcnt (6 = the three younger half-tiles) retires it one phase BEFORE it is read —
// the barrier that separates wait and read is what covers the other waves' loads.  All k-steps of all tiles of the (persistent,
// XCD-aware) workgroup form one pipeline; an epilogue's stores are never waited for on the critical path.
//
// The epilogue (+ bias, SiLU, bf16) transposes through four 1 KB chunks per wave that are exactly the chunks this wave restages
// next (the W1 / X1 half-tiles of the K-tile just finished), so no other wave's load can land in them while they are in use.
#include "scan_helpers.h"

namespace zigma {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) unsigned char *lds_ptr_t;
typedef const __attribute__((address_space(1))) void *gbl_ptr_t;

constexpr int kL8HT = 128 * 128;                 // one half-tile: 128 rows x 128 B
constexpr int kL8Ring = 8 * kL8HT;               // two K-tiles x {X0, W0, W1, X1}
constexpr int kL8X0 = 0, kL8W0 = 1, kL8W1 = 2, kL8X1 = 3;      // order of issue (= order of first use) inside a K-tile
constexpr int kL8Stores = 16;                    // 16-byte store instructions per wave per epilogue

__device__ __forceinline__ void l8_rd(u32x4 &d, unsigned addr) { asm volatile("ds_read_b128 %0, %1" : "=v"(d) : "v"(addr)); }

__device__ __forceinline__ void l8_wait_vm(int n) {
    switch (n) {
#define ZIGMA_VM_CASE(N_) case N_: asm volatile("s_waitcnt vmcnt(" #N_ ")" ::: "memory"); break;
        ZIGMA_VM_CASE(2) ZIGMA_VM_CASE(4) ZIGMA_VM_CASE(6) ZIGMA_VM_CASE(8) ZIGMA_VM_CASE(10) ZIGMA_VM_CASE(16) ZIGMA_VM_CASE(18)
        ZIGMA_VM_CASE(20) ZIGMA_VM_CASE(22)
#undef ZIGMA_VM_CASE
        default: asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); break;
    }
}

template <bool HAS_BIAS>
__global__ __launch_bounds__(512) void linear8_kernel(const zigma_linear_params_t p, const int tiles_m, const int tiles_n) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[kL8Ring + (HAS_BIAS ? 8192 : 0)];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave & 3, wm = wave >> 2;                       // wm is also the group: waves w and w + 4 share a SIMD
    const int j = lane & 31, kh = lane >> 5;
    const int nk = p.k / 64;
    const unsigned char *xb = reinterpret_cast<const unsigned char *>(p.x);
    const unsigned char *wb = reinterpret_cast<const unsigned char *>(p.w);
    const int64_t x_pitch = p.x_row_stride * 2, w_pitch = p.w_row_stride * 2;
    const unsigned smem_lds = static_cast<unsigned>(reinterpret_cast<uintptr_t>((lds_ptr_t)(smem)));
    const int dbg = p.flags;

    // ---- tile schedule: XCD x owns the contiguous raster chunk [x * chunk, (x + 1) * chunk); its workgroups take it round-robin
    const int n_tiles = tiles_m * tiles_n;
    const int xcd = blockIdx.x & 7, slot_in_xcd = blockIdx.x >> 3, wg_per_xcd = gridDim.x >> 3;
    const int chunk = (n_tiles + 7) >> 3;
    const int chunk_end = (xcd + 1) * chunk < n_tiles ? (xcd + 1) * chunk : n_tiles;
    const int tile0 = xcd * chunk + slot_in_xcd;
    if (tile0 >= chunk_end) return;
    const int my_tiles = (chunk_end - tile0 + wg_per_xcd - 1) / wg_per_xcd;
    const int total = my_tiles * nk * 4;                           // phases = half-tile issues of this workgroup's whole run

    // ---- staging: instruction i (0, 1) of wave w fills half-tile rows q * 8 .. q * 8 + 7, q = i * 8 + w.  Half-tile row r of
    // X(th) is token (r >> 6) * 128 + th * 64 + (r & 63) of the tile, of W(fh) feature (r >> 5) * 64 + fh * 32 + (r & 31).
    // 16-byte slot of piece c in row r: c ^ ((r >> 1) & 7), and (r >> 1) & 7 does not depend on i.
    const unsigned piece = ((lane & 7) ^ (((lane >> 4) + 4 * (wave & 1)) & 7)) << 4;
    unsigned off_x[2];
    int feat_l[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int r = (i * 8 + wave) * 8 + (lane >> 3);
        off_x[i] = static_cast<unsigned>(((r >> 6) * 128 + (r & 63)) * x_pitch) + piece;
        feat_l[i] = (r >> 5) * 64 + (r & 31);
    }
    // cursor of the next half-tile to stage: tile -> (x rows base, first feature), K-tile, ring parity; half-tiles issued so far
    int is_tile = tile0, is_kt = 0, is_T = 0, issued = 0;
    const unsigned char *is_xbase = xb + static_cast<int64_t>(tile0 / tiles_n) * 256 * x_pitch;
    int is_f0 = (tile0 % tiles_n) * 256;
    const bool n_full = p.n % 256 == 0;                            // no partial feature tile: no clamp
    auto issue = [&](const int which) {                            // `which` is a compile-time constant at every call site
        unsigned char *dst = smem + ((is_T & 1) * 4 + which) * kL8HT;
        if (which == kL8X0 || which == kL8X1) {
            const unsigned char *base = is_xbase + (which == kL8X1 ? 64 : 0) * x_pitch + is_kt * 128;
#pragma unroll
            for (int i = 0; i < 2; ++i)
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)(base + off_x[i]), (lds_ptr_t)(dst) + (i * 8 + wave) * 1024, 16, 0, 0);
        } else {
            const unsigned char *base = wb + is_kt * 128 + piece;
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                int f = is_f0 + feat_l[i] + (which == kL8W1 ? 32 : 0);
                if (!n_full) f = f < p.n ? f : p.n - 1;            // features beyond n: a valid row, results never stored
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)(base + static_cast<unsigned>(f * static_cast<int>(w_pitch))),
                                                 (lds_ptr_t)(dst) + (i * 8 + wave) * 1024, 16, 0, 0);
            }
        }
        ++issued;
        if (which == kL8X1) {                                      // last half-tile of the K-tile: advance the cursor
            ++is_T;
            if (++is_kt == nk) {
                is_kt = 0;
                is_tile += wg_per_xcd;
                const int mt = is_tile / tiles_n;
                is_xbase = xb + static_cast<int64_t>(mt) * 256 * x_pitch;
                is_f0 = (is_tile - mt * tiles_n) * 256;
            }
        }
    };

    // fragment addresses inside a half-tile (k-substep ks: ^ (ks << 5))
    const unsigned swz = (kh ^ ((j >> 1) & 7)) << 4;
    const unsigned xr_off = (wm * 64 + j) * 128 + swz;             // + mbh * 4096
    const unsigned wr_off = (wn * 32 + j) * 128 + swz;

    if (HAS_BIAS) {
        for (int i = tid; i < p.n / 2; i += 512)
            reinterpret_cast<uint32_t *>(smem + kL8Ring)[i] = reinterpret_cast<const uint32_t *>(p.bias)[i];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    // ---- prologue: six half-tiles in flight, the first two landed for everybody before phase 0 reads them
    if (0 < total) issue(kL8X0);
    if (1 < total) issue(kL8W0);
    if (2 < total) issue(kL8W1);
    if (3 < total) issue(kL8X1);
    if (4 < total) issue(kL8X0);
    if (5 < total) issue(kL8W0);
    l8_wait_vm(2 * (issued - 2));
    __builtin_amdgcn_s_barrier();
    if (wm == 1) __builtin_amdgcn_s_barrier();                     // group 1 runs one barrier behind group 0 from here on

    int P = 0, P_epi = -100, tile = tile0;
    u32x4 Xr[2][4], Wr[2][4];                                      // register subtiles: [mb in half][ks], [fh][ks]
    f32x16 acc[2][4];

    // LOAD section of phase P (ph = P & 3 is a compile-time constant at the call sites)
    auto load_section = [&](const int ph, const int which_next) {
        // (1) this phase's register subtile: what it reads was retired by the previous phase's wait and barrier, so the reads go
        // first and have the whole interval to come back
        const unsigned ring = smem_lds + ((P >> 2) & 1) * 4 * kL8HT;
        if (ph == 0) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) l8_rd(Wr[0][ks], (ring + kL8W0 * kL8HT + wr_off) ^ (ks << 5));
#pragma unroll
            for (int mbh = 0; mbh < 2; ++mbh)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) l8_rd(Xr[mbh][ks], (ring + kL8X0 * kL8HT + xr_off + mbh * 4096) ^ (ks << 5));
        } else if (ph == 1) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) l8_rd(Wr[1][ks], (ring + kL8W1 * kL8HT + wr_off) ^ (ks << 5));
        } else if (ph == 2) {
#pragma unroll
            for (int mbh = 0; mbh < 2; ++mbh)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) l8_rd(Xr[mbh][ks], (ring + kL8X1 * kL8HT + xr_off + mbh * 4096) ^ (ks << 5));
        }
        // (2) retire every half-tile with issue index <= P + 2 (read in phase P + 1, behind this phase's barrier).  VM_CNT retires
        // in issue order: what may stay in flight is what was issued after it — younger half-tiles, and an epilogue's stores while
        // they are younger than index P + 2 (they were issued behind the loads of phase P_epi, i.e. behind index P_epi + 6)
        {
            const int younger = issued - 1 - (P + 2);
            const int n = 2 * (younger > 0 ? younger : 0) + ((P - P_epi >= 1 && P - P_epi <= 4) ? kL8Stores : 0);
            if (n == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");          // the steady state
            else l8_wait_vm(n);
        }
        // (3) restage one half-tile
        if (issued < total && !(dbg & 0x200)) issue(which_next);
        else if (issued < total) ++issued;
    };
    // MFMA section: quadrant (th, fh) of the wave tile, K = 64
    auto mfma_section = [&](const int th, const int fh) {
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(Xr[0][0]), "+v"(Xr[0][1]), "+v"(Xr[0][2]), "+v"(Xr[0][3]), "+v"(Xr[1][0]), "+v"(Xr[1][1]), "+v"(Xr[1][2]),
                       "+v"(Xr[1][3]), "+v"(Wr[fh][0]), "+v"(Wr[fh][1]), "+v"(Wr[fh][2]), "+v"(Wr[fh][3]));
        if (dbg & 0x100) return;
        __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks)
#pragma unroll
            for (int mbh = 0; mbh < 2; ++mbh)
                acc[fh][th * 2 + mbh] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, Wr[fh][ks]), __builtin_bit_cast(bf16x8, Xr[mbh][ks]),
                                                                              acc[fh][th * 2 + mbh], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
#define ZIGMA_L8_PHASE(PH_, NEXT_, TH_, FH_)              \
    load_section(PH_, NEXT_);                             \
    __builtin_amdgcn_sched_barrier(0);                    \
    __builtin_amdgcn_s_barrier();                         \
    mfma_section(TH_, FH_);                               \
    __builtin_amdgcn_sched_barrier(0);

#pragma unroll 1
    for (int ti = 0; ti < my_tiles; ++ti, tile += wg_per_xcd) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
#pragma unroll
            for (int mb = 0; mb < 4; ++mb) acc[nb][mb] = f32x16{};
#pragma unroll 1
        for (int kt = 0; kt < nk; ++kt) {
            // phase P issues half-tile index P + 6: ph 0 -> W1, 1 -> X1, 2 -> X0, 3 -> W0 (of a later K-tile)
            ZIGMA_L8_PHASE(0, kL8W1, 0, 0)
            __builtin_amdgcn_s_barrier(); ++P;
            ZIGMA_L8_PHASE(1, kL8X1, 0, 1)
            __builtin_amdgcn_s_barrier(); ++P;
            ZIGMA_L8_PHASE(2, kL8X0, 1, 1)
            __builtin_amdgcn_s_barrier(); ++P;
            ZIGMA_L8_PHASE(3, kL8W0, 1, 0)
            if (kt + 1 < nk) { __builtin_amdgcn_s_barrier(); ++P; }
        }
        // ---- epilogue of this wave's 128 x 64 tile, in the MFMA interval of the tile's last phase -------------------------------
        if (!(dbg & 0x400)) {
            const int mt = tile / tiles_n, nt = tile - mt * tiles_n;
            const int64_t o_pitch = p.out_row_stride * 2;
            const int n_wave0 = nt * 256 + wn * 64;
            if (n_wave0 < p.n) {                                                           // (n % 64 == 0: a wave's 64 features exist or do not)
                const int64_t m_tile = static_cast<int64_t>(mt) * 256 + wm * 128;
                const rsrc_t o_rs = make_rsrc(reinterpret_cast<unsigned char *>(p.out) + m_tile * o_pitch + n_wave0 * 2, 128 * o_pitch - n_wave0 * 2);
                // scratch: token row t of the 32 x 64 transposition tile lives in chunk t >> 3: chunks 0, 1 in W1, 2, 3 in X1 of the
                // K-tile just finished, at this wave's own staging slots (i * 8 + wave) * 1024
                const unsigned ring = smem_lds + ((P >> 2) & 1) * 4 * kL8HT;
                const unsigned wr_row = ring + kL8W1 * kL8HT + wave * 1024 + ((j >> 3) & 1) * 8192 + (j >> 4) * kL8HT + (j & 7) * 128 + kh * 8;
                const unsigned wr_sw = (j >> 1) & 7;
                const int rd_tok = lane >> 3;
                const unsigned rd_base = ring + kL8W1 * kL8HT + wave * 1024 + rd_tok * 128;
                const unsigned rd_slot = (lane & 7) ^ (rd_tok >> 1);                          // row = i * 8 + rd_tok: (row >> 1) & 7 = (rd_tok >> 1) ^ ((i & 1) << 2)
                const unsigned st_off = static_cast<unsigned>(rd_tok * o_pitch + (lane & 7) * 16);
                const unsigned bias_lds = smem_lds + kL8Ring;
#pragma unroll
                for (int mb = 0; mb < 4; ++mb) {
#pragma unroll
                    for (int nb = 0; nb < 2; ++nb) {
                        const int n0 = n_wave0 + nb * 32;
                        float v[16];
#pragma unroll
                        for (int r = 0; r < 16; ++r) v[r] = acc[nb][mb][r];
                        if (HAS_BIAS) {
                            u32x2 bq[4];
#pragma unroll
                            for (int q = 0; q < 4; ++q) asm volatile("ds_read_b64 %0, %1" : "=v"(bq[q]) : "v"(bias_lds + (n0 + 4 * kh + q * 8) * 2));
                            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(bq[0]), "+v"(bq[1]), "+v"(bq[2]), "+v"(bq[3]));
#pragma unroll
                            for (int q = 0; q < 4; ++q) {
                                v[q * 4 + 0] += __uint_as_float(bq[q].x << 16);
                                v[q * 4 + 1] += __uint_as_float(bq[q].x & 0xffff0000u);
                                v[q * 4 + 2] += __uint_as_float(bq[q].y << 16);
                                v[q * 4 + 3] += __uint_as_float(bq[q].y & 0xffff0000u);
                            }
                        }
                        if (n0 >= p.silu_from_col) {
#pragma unroll
                            for (int r = 0; r < 16; ++r) v[r] = silu(v[r]);
                        }
#pragma unroll
                        for (int q = 0; q < 4; ++q) {
                            u32x2 pk;
                            pk.x = static_cast<uint32_t>(from_float<BF16>(v[q * 4])) | (static_cast<uint32_t>(from_float<BF16>(v[q * 4 + 1])) << 16);
                            pk.y = static_cast<uint32_t>(from_float<BF16>(v[q * 4 + 2])) | (static_cast<uint32_t>(from_float<BF16>(v[q * 4 + 3])) << 16);
                            asm volatile("ds_write_b64 %0, %1" ::"v"(wr_row + (((nb * 4 + q) ^ wr_sw) << 4)), "v"(pk) : "memory");
                        }
                    }
                    u32x4 row[4];
#pragma unroll
                    for (int i = 0; i < 4; ++i)                                                // chunk i: + (i & 1) * 8 KB, + (i >> 1) half-tiles
                        l8_rd(row[i], rd_base + (i & 1) * 8192 + (i >> 1) * kL8HT + ((rd_slot ^ ((i & 1) << 2)) << 4));
                    asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(row[0]), "+v"(row[1]), "+v"(row[2]), "+v"(row[3]));
#pragma unroll
                    for (int i = 0; i < 4; ++i)
                        __builtin_amdgcn_raw_buffer_store_b128(row[i], o_rs, st_off, static_cast<int>((mb * 32 + i * 8) * o_pitch), 0);
                }
                P_epi = P;
            }
        }
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier(); ++P;
    }
#undef ZIGMA_L8_PHASE
    if (wm == 0) __builtin_amdgcn_s_barrier();                     // pairs with group 1's last barrier
}

int launch_linear8(const zigma_linear_params_t &p, hipStream_t stream) {
    const int tiles_m = static_cast<int>(p.m / 256), tiles_n = (p.n + 255) / 256;
    const int64_t n_tiles = static_cast<int64_t>(tiles_m) * tiles_n;
    if (n_tiles > 0x7fffffff) return ZIGMA_ERR_SHAPE;
    int grid = 256;                                  // one persistent workgroup per CU; multiples of 8 keep the XCD map
    if (n_tiles < grid) grid = static_cast<int>((n_tiles + 7) / 8 * 8);
    if (p.bias) hipLaunchKernelGGL((linear8_kernel<true>), dim3(grid), dim3(512), 0, stream, p, tiles_m, tiles_n);
    else hipLaunchKernelGGL((linear8_kernel<false>), dim3(grid), dim3(512), 0, stream, p, tiles_m, tiles_n);
    set_last_kernel("linear8_256x256");
    return check_launch();
}

}  // namespace zigma
