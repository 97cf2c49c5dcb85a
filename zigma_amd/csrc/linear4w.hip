// linear4w: dense projections out = x @ W^T (bf16 in, fp32 accumulate, bf16 out) with ONE wave per SIMD — in_proj, out_proj
// (mamba_simple.py:290-294, selective_scan_interface.py:365) and to_q / to_out (model_zigma.py:104-135) of the ZigMa block,
// reference F.linear; with the block's gated branch add `residual + gate * (x W^T + bias)` (model_zigma.py:441-449) in the epilogue.
//
//   workgroup = 4 waves (one per SIMD, 512 registers per lane: 256 accumulators in AGPRs), one per CU, persistent over an XCD-aware
//   tile list; tile = 256 tokens x 256 features (x 128 for the remainder of n % 256 == 128), wave tile 128 x 128 (16 blocks of
//   v_mfma_f32_32x32x16_bf16), BK = 64; two 64 KB LDS stages filled by global_load_lds_dwordx4 one k-step ahead (source-side bank
//   swizzle), one barrier per k-step placed before its LAST sub-step so that the next k-step's fragments are already in flight when
//   it starts; the last k-step of a tile runs block-pair-major and carries the epilogue of the previous pair in its MFMA gaps:
//   accumulators -> LDS in fp32 straight from the AGPRs -> 32-byte row pieces (+ residual rows, gate) -> bf16 -> 16-byte stores
//   (8 rows x 128 B each); the bias enters as a rank-1 MFMA (bias x ones) on the accumulators.
//
// The whole loop is ONE asm statement per variant, generated (and simulated, on the CPU) by csrc/gen/linear4w_gen.py: with a single
// wave per SIMD every instruction has to be placed between the MFMAs by hand, see the header of the generator.  This file only
// computes the statement's operands.  Limits (zigma_linear_fwd falls back to linear_tn_kernel otherwise): m % 256 == 0,
// n % 128 == 0, k % 64 == 0, k >= 192, at least 256 tiles; gated residual: residual rows in the output's pitch (a multiple of 128
// elements), samples of 2^i >= 128 rows; a bias only together with the gated residual (to_out), n <= 8192.
#include "zigma_common.h"
#include "linear4w_body.inc"

namespace zigma {

typedef __attribute__((address_space(3))) unsigned char *lds4w_ptr_t;

// EPI: 0 = 256-wide tiles only, no epilogue operands (in_proj, to_q); 1 = + narrow tiles; 2 = + gated residual; 3 = + bias.
// VARIANT > 0: timing probes of tools/linear4w_probe.py (only in a library built with -DZIGMA_LINEAR4W_PROBES, EPI 0)
template <int EPI, int VARIANT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void linear4w_kernel(const zigma_linear_params_t p, const int tiles_n, const int n_wide, const int n_tiles) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[163840];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // tile list of this workgroup: XCD x (blockIdx % 8) owns the raster chunk [x * chunk, (x + 1) * chunk), its workgroups walk it
    // round-robin — a few 256-token activation panels x all weight panels stay in that XCD's L2
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, wg_per_xcd = gridDim.x >> 3;
    const int chunk = (n_tiles + 7) >> 3;
    const int chunk_end = (xcd + 1) * chunk < n_tiles ? (xcd + 1) * chunk : n_tiles;
    const int tile0i = xcd * chunk + slot;
    if (tile0i >= chunk_end) return;
    const int my_tiles = (chunk_end - tile0i + wg_per_xcd - 1) / wg_per_xcd;
#ifdef ZIGMA_LINEAR4W_PROBES
    // start skew (probe): every CU finishes its tiles — and fires its 128 KB of stores — at the same moment otherwise
    for (int i = 0, n = (static_cast<int>(blockIdx.x) * ((p.flags >> 20) & 15)) >> 3; i < n; ++i) __builtin_amdgcn_s_sleep(16);
#endif
    const unsigned lds_base = static_cast<unsigned>(reinterpret_cast<uintptr_t>((lds4w_ptr_t)(smem)));
    const unsigned w_pitch = static_cast<unsigned>(p.w_row_stride * 2), x_pitch = static_cast<unsigned>(p.x_row_stride * 2),
                   o_pitch = static_cast<unsigned>(p.out_row_stride * 2);
    // scalars that never change, packed (an asm statement takes at most 30 operands)
    const unsigned dims = static_cast<unsigned>(p.k / 64) | (static_cast<unsigned>(tiles_n) << 12) | (static_cast<unsigned>(n_wide) << 22);
    const unsigned steps = static_cast<unsigned>(wg_per_xcd / tiles_n) | (static_cast<unsigned>(wg_per_xcd % tiles_n) << 20);
    const unsigned tile0 = static_cast<unsigned>(tile0i / tiles_n) | (static_cast<unsigned>(tile0i % tiles_n) << 20);
    const unsigned wave_lds = lds_base + static_cast<unsigned>(wave);
    // per-lane constants (the same expressions as lane_operands() in csrc/gen/linear4w_sim.py)
    const unsigned j = lane & 31, kh = lane >> 5, wn = wave & 1, wm = wave >> 1;
    const unsigned piece = (lane & 7) ^ (((wave & 1) << 2) | (lane >> 4));
    const unsigned srow = wave * 8 + (lane >> 3);
    const unsigned sw = (j >> 1) & 7, u = lane & 7, t8 = lane >> 3;
    const unsigned voffw0 = srow * w_pitch + piece * 16, voffx0 = srow * x_pitch + piece * 16;
    const unsigned a_base = lds_base + (wn * 128 + j) * 128, b_base = lds_base + (256 + wm * 128 + j) * 128;
    const unsigned t_xor = kh ^ sw;
    const unsigned scrw_base = lds_base + 2 * 65536 + wave * 8192 + j * 256 + kh * 16, j7 = j & 7;
    const unsigned scrr = lds_base + 2 * 65536 + wave * 8192 + t8 * 256 + ((u ^ t8) << 5);
    const unsigned stoff = t8 * o_pitch + u * 16;
    const unsigned bias_voff = lane < 32 ? j * 2 : 0x7fff0000u, ones0 = lane < 32 ? 0x3f80u : 0u;
    const void *w_ptr = p.w, *x_ptr = p.x;
    void *out_ptr = p.out;
    const uint64_t res_a = reinterpret_cast<uint64_t>(p.residual), gate_a = reinterpret_cast<uint64_t>(p.gate), bias_a = reinterpret_cast<uint64_t>(p.bias);
    const unsigned res_lo = static_cast<unsigned>(res_a), res_hi = static_cast<unsigned>(res_a >> 32);
    const unsigned gate_lo = static_cast<unsigned>(gate_a), gate_hi = static_cast<unsigned>(gate_a >> 32);
    const unsigned bias_lo = static_cast<unsigned>(bias_a), bias_hi = static_cast<unsigned>(bias_a >> 32);
    const unsigned gate_bstride = static_cast<unsigned>(p.gate_batch_stride * 2);
    const unsigned rpb_shift = p.rows_per_batch > 0 ? static_cast<unsigned>(31 - __builtin_clz(static_cast<unsigned>(p.rows_per_batch))) : 0u;
#define ZIGMA_L4W_ASM(BODY_)                                                                                                                    \
    asm volatile(BODY_                                                                                                                          \
                 :                                                                                                                              \
                 : ZIGMA_LINEAR4W_OPERANDS(voffw0, voffx0, a_base, b_base, t_xor, scrw_base, j7, scrr, stoff, bias_voff, ones0, w_ptr, x_ptr,  \
                                           out_ptr, w_pitch, x_pitch, o_pitch, dims, my_tiles, steps, tile0, wave_lds, res_lo, res_hi,        \
                                           gate_lo, gate_hi, gate_bstride, rpb_shift, bias_lo, bias_hi)                                       \
                 : ZIGMA_LINEAR4W_CLOBBERS)
    if constexpr (EPI == 1) { ZIGMA_L4W_ASM(ZIGMA_LINEAR4W_BODY_N); }
    else if constexpr (EPI == 2) { ZIGMA_L4W_ASM(ZIGMA_LINEAR4W_BODY_NR); }
    else if constexpr (EPI == 3) { ZIGMA_L4W_ASM(ZIGMA_LINEAR4W_BODY_NRB); }
    else if constexpr (VARIANT == 0) { ZIGMA_L4W_ASM(ZIGMA_LINEAR4W_BODY); }
#ifdef ZIGMA_LINEAR4W_PROBES
    else if constexpr (VARIANT == 1) { ZIGMA_L4W_ASM(ZIGMA_LINEAR4W_BODY_NOMFMA); }
    else if constexpr (VARIANT == 2) { ZIGMA_L4W_ASM(ZIGMA_LINEAR4W_BODY_LOADS); }
    else if constexpr (VARIANT == 3) { ZIGMA_L4W_ASM(ZIGMA_LINEAR4W_BODY_NOGLDS); }
    else if constexpr (VARIANT == 4) { ZIGMA_L4W_ASM(ZIGMA_LINEAR4W_BODY_NOSTORE); }
    else if constexpr (VARIANT == 5) { ZIGMA_L4W_ASM(ZIGMA_LINEAR4W_BODY_MFMAONLY); }
    else if constexpr (VARIANT == 6) { ZIGMA_L4W_ASM(ZIGMA_LINEAR4W_BODY_NOGLDS_LAX); }
    else if constexpr (VARIANT == 7) { ZIGMA_L4W_ASM(ZIGMA_LINEAR4W_BODY_MFMA_NOEPI); }
#endif
#undef ZIGMA_L4W_ASM
}

// which variant serves the call, or -1
static int linear4w_variant(const zigma_linear_params_t &p) {
#ifdef ZIGMA_LINEAR4W_PROBES
    if (p.flags & ~0xf70000) return -1;                    // 0x10000 .. 0x70000: probe variant 1 .. 7; 0x100000 * k: start skew
#else
    if (p.flags) return -1;
#endif
    if (p.silu_from_col < p.n) return -1;
    if (p.m % 256 != 0 || p.n % 128 != 0 || p.k % 64 != 0 || p.k < 192 || p.k / 64 > 4095) return -1;
    if (p.out_row_stride % 8 != 0 || reinterpret_cast<uintptr_t>(p.out) % 16 != 0) return -1;               // 16-byte stores
    if (p.m * p.out_row_stride * 2 > 0xffffffffll) return -1;                                                // 32-bit tile offsets
    const int64_t tiles_n = p.n / 256 + (p.n % 256 != 0), n_tiles = (p.m / 256) * tiles_n;
    if (n_tiles < 256 || n_tiles > 0x7fffffff || tiles_n > 1023 || p.m / 256 > 0xfffff) return -1;   // >= one tile per CU (smaller: the 8-wave kernel)
    int epi = p.n % 256 != 0 ? 1 : 0;
    if (p.residual) {
        if (!p.gate || p.res_row_stride != p.out_row_stride || p.out_row_stride % 128 != 0) return -1;
        if (p.rows_per_batch < 128 || (p.rows_per_batch & (p.rows_per_batch - 1)) != 0 || p.m % p.rows_per_batch != 0) return -1;
        if (reinterpret_cast<uintptr_t>(p.residual) % 16 != 0 || reinterpret_cast<uintptr_t>(p.gate) % 16 != 0 || p.gate_batch_stride % 8 != 0) return -1;
        epi = 2;
    }
    if (p.bias) {
        if (!p.residual || p.n > 8192 || reinterpret_cast<uintptr_t>(p.bias) % 2 != 0) return -1;
        epi = 3;
    }
    if (epi != 0 && (p.flags >> 16)) return -1;
    return epi;
}

bool linear4w_eligible(const zigma_linear_params_t &p) { return linear4w_variant(p) >= 0; }

int launch_linear4w(const zigma_linear_params_t &p, hipStream_t stream) {
    const int n_wide = p.n / 256, tiles_n = n_wide + (p.n % 256 != 0);
    const int n_tiles = static_cast<int>((p.m / 256) * tiles_n);
    const dim3 grid(256), block(256);
#define ZIGMA_L4W(E_, V_) hipLaunchKernelGGL((linear4w_kernel<E_, V_>), grid, block, 0, stream, p, tiles_n, n_wide, n_tiles)
    switch (linear4w_variant(p)) {
        case 1: ZIGMA_L4W(1, 0); break;
        case 2: ZIGMA_L4W(2, 0); break;
        case 3: ZIGMA_L4W(3, 0); break;
        default:
            switch ((p.flags >> 16) & 7) {
#ifdef ZIGMA_LINEAR4W_PROBES
                case 1: ZIGMA_L4W(0, 1); break;
                case 2: ZIGMA_L4W(0, 2); break;
                case 3: ZIGMA_L4W(0, 3); break;
                case 4: ZIGMA_L4W(0, 4); break;
                case 5: ZIGMA_L4W(0, 5); break;
                case 6: ZIGMA_L4W(0, 6); break;
                case 7: ZIGMA_L4W(0, 7); break;
#endif
                default: ZIGMA_L4W(0, 0);
            }
    }
#undef ZIGMA_L4W
    set_last_kernel(p.n % 256 ? "linear4w_256x256+128" : "linear4w_256x256");
    return check_launch();
}

}  // namespace zigma
