// Micro-benchmark (not product code): can the 4x4x1 f32 MFMA take the rank-1 product B*du and the C.h contraction off the VALU in the
// scan core?  A stripped replica of scan_tok2_kernel's tile loop (operands from LDS, 4 waves x 4 states, lane = channel, y hand-over
// through LDS, two barriers per tile; no global loads inside the loop) in four forms:
//   0: the shipped VALU core (6 packed + 4 exp + 4 plain per step)     1: MFMA form (4 packed + 4 exp + 5 v_mfma_f32_4x4x1_16b_f32)
//   2: form 1 without the y products                                    3: form 0 without the y chain
//   4: form 0, y summed in registers (no hand-over)   5: form 0, hand-over as one 16-byte write per 4 steps (the shipped form)
//   6: form 5 with C_l from scalar loads (global fp32 table) instead of LDS broadcast reads   7: form 6 with B_l too
//   8: form 7 from a 16-bit table (2 x s_load_dwordx2 + 8 SALU conversions per step): the deployable form
//   10 / 11: forms 7 / 8 with the hand-over of form 0 (one 4-byte write per step)
//   9: functional probe of the 4x4x1 operand / result layout
//   12 (round 6): the BARRIER-FREE decomposition — a wave = 16 channels x 4 state-quads (lane = (row r, quad q, j): channel 4 r + j, states
//       4 q .. 4 q + 3), y reduced across the four lanes of a channel by two DPP adds per step (row_ror:8, row_ror:4), the second one bank-masked so that
//       lane q keeps the steps s = q mod 4 (the per-element work stays 4 elements per lane); no s_y hand-over, no workgroup barrier
//   13: form 4 (y summed in registers) without the barriers = what form 0 would cost with no hand-over and no synchronisation at all
//   15: two waves x 8 states per slab (128-thread workgroups), hand-over and barriers as in form 0      16: form 15 with y summed in registers, no barriers (its floor)
//   14: form 12 with the hazard no-ops of its DPP adds left to luck (NOT valid code: a lower bound of what scheduling them away could reach)
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }

template <int MODE>
__global__ __launch_bounds__(256, 5) void core(float *__restrict__ out, const float *__restrict__ in, int tiles) {
    __shared__ __attribute__((aligned(16))) float s_dtdu[16][64][2];
    __shared__ __attribute__((aligned(16))) float s_bc[16][2][16];
    __shared__ __attribute__((aligned(16))) float s_y[4][16][64];
    float (*s_y4)[4][64][4] = reinterpret_cast<float (*)[4][64][4]>(&s_y[0][0][0]);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), n0 = wave * 4;
    if constexpr (MODE == 9) {
        const v4f d = __builtin_amdgcn_mfma_f32_4x4x1f32(static_cast<float>(lane), 100.f + lane, v4f{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        if (blockIdx.x == 0 && wave == 0) { out[lane * 4] = d.x; out[lane * 4 + 1] = d.y; out[lane * 4 + 2] = d.z; out[lane * 4 + 3] = d.w; }
        return;
    }
    for (int i = tid; i < 16 * 64; i += 256) {
        s_dtdu[i >> 6][i & 63][0] = 0.01f + 0.2f * in[(i * 7) & 1023];
        s_dtdu[i >> 6][i & 63][1] = in[(i * 3 + 1) & 1023] - 0.5f;
    }
    for (int i = tid; i < 512; i += 256) s_bc[i >> 5][(i >> 4) & 1][i & 15] = in[(i + 17) & 1023] - 0.5f;
    if constexpr (MODE == 12 || MODE == 14) {
        const int q = (lane >> 2) & 3, ch = wave * 16 + 4 * (lane >> 4) + (lane & 3), nq = q * 4;
        v2f a2A = {-1.f - in[ch + nq], -2.f - in[ch + nq + 64]}, a2B = {-3.f - in[ch + nq + 128], -4.f - in[ch + nq + 192]};
        v2f hA = {0.f, 0.f}, hB = {0.f, 0.f};
        float acc = 0.f;
        __syncthreads();
#pragma unroll 1
        for (int t = 0; t < tiles; ++t) {
            float keep[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int s = 0; s < 16; ++s) {
                const v2f dc = *reinterpret_cast<const v2f *>(&s_dtdu[s][ch][0]);        // 16 distinct addresses, each read by the 4 lanes of a channel
                const v4f Bc = *reinterpret_cast<const v4f *>(&s_bc[s][0][nq]);          // 4 distinct addresses
                const v4f Cc = *reinterpret_cast<const v4f *>(&s_bc[s][1][nq]);
                const v2f dtv = {dc.x, dc.x}, duv = {dc.y, dc.y};
                const v2f dA = a2A * dtv, dB = a2B * dtv;
                const v2f eA = {ex2(dA.x), ex2(dA.y)}, eB = {ex2(dB.x), ex2(dB.y)};
                const v2f bA = v2f{Bc.x, Bc.y} * duv, bB = v2f{Bc.z, Bc.w} * duv;
                hA = __builtin_elementwise_fma(eA, hA, bA);
                hB = __builtin_elementwise_fma(eB, hB, bB);
                const float y = __builtin_fmaf(Cc.w, hB.y, __builtin_fmaf(Cc.z, hB.x, __builtin_fmaf(Cc.y, hA.y, Cc.x * hA.x)));
                float y1;
                if constexpr (MODE == 12) {      // (VALU write -> DPP read of the same register: 2 wait states, which the compiler cannot see inside the asm)
                    asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "=v"(y1) : "v"(y));
                    asm("s_nop 1\n\tv_add_f32_dpp %0, %1, %1 row_ror:4 row_mask:0xf bank_mask:%2" : "+v"(keep[s >> 2]) : "v"(y1), "n"(1 << (s & 3)));
                } else {
                    asm("v_add_f32_dpp %0, %1, %1 row_ror:8 row_mask:0xf bank_mask:0xf" : "=v"(y1) : "v"(y));
                    asm("v_add_f32_dpp %0, %1, %1 row_ror:4 row_mask:0xf bank_mask:%2" : "+v"(keep[s >> 2]) : "v"(y1), "n"(1 << (s & 3)));
                }
            }
            acc += keep[0] + keep[1] + keep[2] + keep[3];
        }
        out[blockIdx.x * 256 + tid] = acc + hA.x + hA.y + hB.x + hB.y;
        return;
    }
    v2f a2A = {-1.f - in[lane], -2.f - in[lane + 64]}, a2B = {-3.f - in[lane + 128], -4.f - in[lane + 192]};
    v2f hA = {0.f, 0.f}, hB = {0.f, 0.f};
    float acc = 0.f;
    __syncthreads();
#pragma unroll 1
    for (int t = 0; t < tiles; ++t) {
        float yg[4];
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const v2f dc = *reinterpret_cast<const v2f *>(&s_dtdu[s][lane][0]);
            const v4f Cc = *reinterpret_cast<const v4f *>(&s_bc[s][1][n0]);
            const v2f dtv = {dc.x, dc.x}, duv = {dc.y, dc.y};
            const v2f dA = a2A * dtv, dB = a2B * dtv;
            const v2f eA = {ex2(dA.x), ex2(dA.y)}, eB = {ex2(dB.x), ex2(dB.y)};
            if constexpr (MODE == 0 || MODE == 3 || MODE >= 4) {      // (12 / 14 returned above)
                v4f Bc = *reinterpret_cast<const v4f *>(&s_bc[s][0][n0]), Cs = Cc;
                if constexpr (MODE >= 6) Cs = *reinterpret_cast<const v4f *>(in + ((t & 1) * 512 + s * 32 + 16 + n0));     // wave-uniform address
                if constexpr (MODE == 7 || MODE == 10) Bc = *reinterpret_cast<const v4f *>(in + ((t & 1) * 512 + s * 32 + n0));
                if constexpr (MODE == 8 || MODE == 11) {
                    const uint2 bw = *reinterpret_cast<const uint2 *>(in + ((t & 1) * 256 + s * 16 + (n0 >> 1)));
                    const uint2 cw = *reinterpret_cast<const uint2 *>(in + ((t & 1) * 256 + s * 16 + 8 + (n0 >> 1)));
                    Bc = v4f{__uint_as_float(bw.x << 16), __uint_as_float(bw.x & 0xffff0000u), __uint_as_float(bw.y << 16), __uint_as_float(bw.y & 0xffff0000u)};
                    Cs = v4f{__uint_as_float(cw.x << 16), __uint_as_float(cw.x & 0xffff0000u), __uint_as_float(cw.y << 16), __uint_as_float(cw.y & 0xffff0000u)};
                }
                const v2f bA = v2f{Bc.x, Bc.y} * duv, bB = v2f{Bc.z, Bc.w} * duv;
                hA = __builtin_elementwise_fma(eA, hA, bA);
                hB = __builtin_elementwise_fma(eB, hB, bB);
                if constexpr (MODE != 3) {
                    const float y = __builtin_fmaf(Cs.w, hB.y, __builtin_fmaf(Cs.z, hB.x, __builtin_fmaf(Cs.y, hA.y, Cs.x * hA.x)));
                    if constexpr (MODE == 0 || MODE >= 10) s_y[wave][s][lane] = y;
                    else if constexpr (MODE == 4 || MODE == 13) acc += y;
                    else {
                        yg[s & 3] = y;
                        if ((s & 3) == 3) *reinterpret_cast<v4f *>(&s_y4[wave][s >> 2][lane][0]) = v4f{yg[0], yg[1], yg[2], yg[3]};
                    }
                }
            } else {
                const float Bsel = s_bc[s][0][n0 + (lane & 3)];
                const v4f tt = __builtin_amdgcn_mfma_f32_4x4x1f32(Bsel, dc.y, v4f{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                hA = __builtin_elementwise_fma(eA, hA, v2f{tt.x, tt.y});
                hB = __builtin_elementwise_fma(eB, hB, v2f{tt.z, tt.w});
                if constexpr (MODE == 1) {
                    v4f d = __builtin_amdgcn_mfma_f32_4x4x1f32(hA.x, Cc.x, v4f{0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_4x4x1f32(hA.y, Cc.y, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_4x4x1f32(hB.x, Cc.z, d, 0, 0, 0);
                    d = __builtin_amdgcn_mfma_f32_4x4x1f32(hB.y, Cc.w, d, 0, 0, 0);
                    if ((lane & 3) == 0) *reinterpret_cast<v4f *>(&s_y[wave][s][lane]) = d;      // channels lane .. lane + 3 of step s
                }
            }
        }
        if (MODE != 13) __syncthreads();
        if (MODE >= 5 && MODE <= 8) {
            v4f ys = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int w = 0; w < 4; ++w) ys += *reinterpret_cast<const v4f *>(&s_y4[w][wave][lane][0]);
            acc += ys.x + ys.y + ys.z + ys.w;
        }
        if (MODE == 0 || MODE == 1 || MODE >= 10) {
#pragma unroll
            for (int e = 0; e < 4; ++e) acc += s_y[0][wave * 4 + e][lane] + s_y[1][wave * 4 + e][lane] + s_y[2][wave * 4 + e][lane] + s_y[3][wave * 4 + e][lane];
        }
        if (MODE != 13) __syncthreads();
    }
    out[blockIdx.x * 256 + tid] = acc + hA.x + hA.y + hB.x + hB.y;
}

// form 15 (round 6): TWO waves x 8 states per (sample, slab) instead of four x 4 — half the (dt, dt u) reads and half the partial-y hand-over per slab, barriers
// couple two waves instead of four, 2.5 waves per SIMD with twice the independent chains each.  128-thread workgroups; tiles of 16 steps; hand-over as in form 0.
template <int MODE>
__global__ __launch_bounds__(128, 3) void core8(float *__restrict__ out, const float *__restrict__ in, int tiles) {
    __shared__ __attribute__((aligned(16))) float s_dtdu[16][64][2];
    __shared__ __attribute__((aligned(16))) float s_bc[16][2][16];
    __shared__ __attribute__((aligned(16))) float s_y[2][16][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), n0 = wave * 8;
    for (int i = tid; i < 16 * 64; i += 128) {
        s_dtdu[i >> 6][i & 63][0] = 0.01f + 0.2f * in[(i * 7) & 1023];
        s_dtdu[i >> 6][i & 63][1] = in[(i * 3 + 1) & 1023] - 0.5f;
    }
    for (int i = tid; i < 512; i += 128) s_bc[i >> 5][(i >> 4) & 1][i & 15] = in[(i + 17) & 1023] - 0.5f;
    v2f a2[4], h[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) { a2[k] = v2f{-1.f - k - in[lane + 64 * (k & 3)], -1.5f - k - in[lane + 32 * k]}; h[k] = v2f{0.f, 0.f}; }
    float acc = 0.f;
    __syncthreads();
#pragma unroll 1
    for (int t = 0; t < tiles; ++t) {
#pragma unroll
        for (int s = 0; s < 16; ++s) {
            const v2f dc = *reinterpret_cast<const v2f *>(&s_dtdu[s][lane][0]);
            const v4f B0 = *reinterpret_cast<const v4f *>(&s_bc[s][0][n0]), B1 = *reinterpret_cast<const v4f *>(&s_bc[s][0][n0 + 4]);
            const v4f C0 = *reinterpret_cast<const v4f *>(&s_bc[s][1][n0]), C1 = *reinterpret_cast<const v4f *>(&s_bc[s][1][n0 + 4]);
            const v2f dtv = {dc.x, dc.x}, duv = {dc.y, dc.y};
            const v2f Bv[4] = {v2f{B0.x, B0.y}, v2f{B0.z, B0.w}, v2f{B1.x, B1.y}, v2f{B1.z, B1.w}};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const v2f dA = a2[k] * dtv;
                const v2f e = {ex2(dA.x), ex2(dA.y)};
                h[k] = __builtin_elementwise_fma(e, h[k], Bv[k] * duv);
            }
            float y = C0.x * h[0].x;
            y = __builtin_fmaf(C0.y, h[0].y, y); y = __builtin_fmaf(C0.z, h[1].x, y); y = __builtin_fmaf(C0.w, h[1].y, y);
            y = __builtin_fmaf(C1.x, h[2].x, y); y = __builtin_fmaf(C1.y, h[2].y, y); y = __builtin_fmaf(C1.z, h[3].x, y); y = __builtin_fmaf(C1.w, h[3].y, y);
            if constexpr (MODE == 15) s_y[wave][s][lane] = y; else acc += y;
        }
        if constexpr (MODE == 15) {
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 8; ++e) acc += s_y[0][wave * 8 + e][lane] + s_y[1][wave * 8 + e][lane];
            __syncthreads();
        }
    }
    out[blockIdx.x * 128 + tid] = acc + h[0].x + h[1].y + h[2].x + h[3].y;
}

extern "C" int ubench3_launch(int mode, int blocks, int tiles, float *out, const float *in, void *stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    dim3 g(blocks), b(256);
    switch (mode) {
        case 0: hipLaunchKernelGGL(core<0>, g, b, 0, st, out, in, tiles); break;
        case 1: hipLaunchKernelGGL(core<1>, g, b, 0, st, out, in, tiles); break;
        case 2: hipLaunchKernelGGL(core<2>, g, b, 0, st, out, in, tiles); break;
        case 3: hipLaunchKernelGGL(core<3>, g, b, 0, st, out, in, tiles); break;
        case 4: hipLaunchKernelGGL(core<4>, g, b, 0, st, out, in, tiles); break;
        case 5: hipLaunchKernelGGL(core<5>, g, b, 0, st, out, in, tiles); break;
        case 6: hipLaunchKernelGGL(core<6>, g, b, 0, st, out, in, tiles); break;
        case 7: hipLaunchKernelGGL(core<7>, g, b, 0, st, out, in, tiles); break;
        case 8: hipLaunchKernelGGL(core<8>, g, b, 0, st, out, in, tiles); break;
        case 10: hipLaunchKernelGGL(core<10>, g, b, 0, st, out, in, tiles); break;
        case 11: hipLaunchKernelGGL(core<11>, g, b, 0, st, out, in, tiles); break;
        case 9: hipLaunchKernelGGL(core<9>, g, b, 0, st, out, in, tiles); break;
        case 12: hipLaunchKernelGGL(core<12>, g, b, 0, st, out, in, tiles); break;
        case 13: hipLaunchKernelGGL(core<13>, g, b, 0, st, out, in, tiles); break;
        case 14: hipLaunchKernelGGL(core<14>, g, b, 0, st, out, in, tiles); break;
        case 15: hipLaunchKernelGGL(core8<15>, g, dim3(128), 0, st, out, in, tiles); break;
        case 16: hipLaunchKernelGGL(core8<16>, g, dim3(128), 0, st, out, in, tiles); break;
        default: return -1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -5;
}
