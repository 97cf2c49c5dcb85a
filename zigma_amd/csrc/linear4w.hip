// linear4w: dense projection out = x @ W^T (bf16 in, fp32 accumulate, bf16 out) with ONE wave per SIMD — the arrangement for the
// wide projections of the ZigMa block, in_proj (mamba_simple.py:290-294) and to_q (model_zigma.py:104-112), reference F.linear.
//
//   workgroup = 4 waves (one per SIMD, 512 registers per lane: 256 accumulators in AGPRs), one per CU, persistent over an XCD-aware
//   tile list; tile = 256 tokens x 256 features, wave tile 128 x 128 (16 blocks of v_mfma_f32_32x32x16_bf16), BK = 64;
//   two 64 KB LDS stages filled by global_load_lds_dwordx4 one k-step ahead (source-side bank swizzle), one barrier per k-step
//   placed before its LAST sub-step so that the next k-step's fragments are already in flight when it starts;
//   the last k-step of a tile runs block-pair-major and carries the epilogue of the previous pair in its MFMA gaps:
//   accumulators -> LDS in fp32 straight from the AGPRs -> 32-byte row pieces -> bf16 -> 16-byte stores (8 rows x 128 B each).
//
// The whole loop is ONE asm statement generated (and simulated, on the CPU) by csrc/gen/linear4w_gen.py: with a single wave per
// SIMD every instruction has to be placed between the MFMAs by hand, see the header of the generator.  This file only computes the
// statement's operands.  Limits (zigma_linear_fwd falls back to linear_tn_kernel otherwise): m % 256 == 0, n % 256 == 0,
// k % 64 == 0, k >= 192, no bias / SiLU range / residual.
#include "zigma_common.h"
#include "linear4w_body.inc"

namespace zigma {

typedef __attribute__((address_space(3))) unsigned char *lds4w_ptr_t;

// VARIANT: 0 = the kernel; 1 .. 5 = timing probes of tools/linear4w_probe.py (only in a library built with -DZIGMA_LINEAR4W_PROBES)
template <int VARIANT>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 1)))
void linear4w_kernel(const zigma_linear_params_t p, const int tiles_n, const int n_tiles) {
    __shared__ __attribute__((aligned(1024))) unsigned char smem[163840];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // tile list of this workgroup: XCD x (blockIdx % 8) owns the raster chunk [x * chunk, (x + 1) * chunk), its workgroups walk it
    // round-robin — a few 256-token activation panels x all weight panels stay in that XCD's L2
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, wg_per_xcd = gridDim.x >> 3;
    const int chunk = (n_tiles + 7) >> 3;
    const int chunk_end = (xcd + 1) * chunk < n_tiles ? (xcd + 1) * chunk : n_tiles;
    const int tile0 = xcd * chunk + slot;
    if (tile0 >= chunk_end) return;
    const int my_tiles = (chunk_end - tile0 + wg_per_xcd - 1) / wg_per_xcd;
    const int step_m = wg_per_xcd / tiles_n, step_n = wg_per_xcd % tiles_n;
    const int mt0 = tile0 / tiles_n, nt0 = tile0 % tiles_n;

#ifdef ZIGMA_LINEAR4W_PROBES
    // start skew (probe): every CU finishes its tiles — and fires its 128 KB of stores — at the same moment otherwise
    for (int i = 0, n = (static_cast<int>(blockIdx.x) * ((p.flags >> 20) & 15)) >> 3; i < n; ++i) __builtin_amdgcn_s_sleep(16);
#endif
    const unsigned lds_base = static_cast<unsigned>(reinterpret_cast<uintptr_t>((lds4w_ptr_t)(smem)));
    const unsigned w_pitch = static_cast<unsigned>(p.w_row_stride * 2), x_pitch = static_cast<unsigned>(p.x_row_stride * 2),
                   o_pitch = static_cast<unsigned>(p.out_row_stride * 2);
    const int nk = p.k / 64;
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
    const void *w_ptr = p.w, *x_ptr = p.x;
    void *out_ptr = p.out;
#define ZIGMA_L4W_ASM(BODY_)                                                                                                                    \
    asm volatile(BODY_                                                                                                                          \
                 :                                                                                                                              \
                 : ZIGMA_LINEAR4W_OPERANDS(voffw0, voffx0, a_base, b_base, t_xor, scrw_base, j7, scrr, stoff, w_ptr, x_ptr, out_ptr, w_pitch,  \
                                           x_pitch, o_pitch, nk, tiles_n, my_tiles, step_m, step_n, mt0, nt0, wave, lds_base)                  \
                 : ZIGMA_LINEAR4W_CLOBBERS)
    if constexpr (VARIANT == 0) { ZIGMA_L4W_ASM(ZIGMA_LINEAR4W_BODY); }
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

bool linear4w_eligible(const zigma_linear_params_t &p) {
#ifdef ZIGMA_LINEAR4W_PROBES
    if (p.flags & ~0xf70000) return false;                 // 0x10000 .. 0x50000: probe variant 1 .. 5; 0x100000 * k: start skew
#else
    if (p.flags) return false;
#endif
    if (p.bias || p.residual || p.silu_from_col < p.n) return false;
    if (p.m % 256 != 0 || p.n % 256 != 0 || p.k % 64 != 0 || p.k < 192) return false;
    if (p.out_row_stride % 8 != 0 || reinterpret_cast<uintptr_t>(p.out) % 16 != 0) return false;               // 16-byte stores
    if (p.m * p.out_row_stride * 2 > 0xffffffffll) return false;                                                // 32-bit tile offsets
    const int64_t n_tiles = (p.m / 256) * (p.n / 256);
    return n_tiles >= 256 && n_tiles <= 0x7fffffff;          // at least one tile per CU (smaller problems: the 8-wave kernel)
}

int launch_linear4w(const zigma_linear_params_t &p, hipStream_t stream) {
    const int tiles_n = p.n / 256;
    const int n_tiles = static_cast<int>((p.m / 256) * tiles_n);
    switch ((p.flags >> 16) & 7) {
#ifdef ZIGMA_LINEAR4W_PROBES
        case 1: hipLaunchKernelGGL(linear4w_kernel<1>, dim3(256), dim3(256), 0, stream, p, tiles_n, n_tiles); break;
        case 2: hipLaunchKernelGGL(linear4w_kernel<2>, dim3(256), dim3(256), 0, stream, p, tiles_n, n_tiles); break;
        case 3: hipLaunchKernelGGL(linear4w_kernel<3>, dim3(256), dim3(256), 0, stream, p, tiles_n, n_tiles); break;
        case 4: hipLaunchKernelGGL(linear4w_kernel<4>, dim3(256), dim3(256), 0, stream, p, tiles_n, n_tiles); break;
        case 5: hipLaunchKernelGGL(linear4w_kernel<5>, dim3(256), dim3(256), 0, stream, p, tiles_n, n_tiles); break;
        case 6: hipLaunchKernelGGL(linear4w_kernel<6>, dim3(256), dim3(256), 0, stream, p, tiles_n, n_tiles); break;
        case 7: hipLaunchKernelGGL(linear4w_kernel<7>, dim3(256), dim3(256), 0, stream, p, tiles_n, n_tiles); break;
#endif
        default: hipLaunchKernelGGL(linear4w_kernel<0>, dim3(256), dim3(256), 0, stream, p, tiles_n, n_tiles);
    }
    set_last_kernel("linear4w_256x256");
    return check_launch();
}

}  // namespace zigma
