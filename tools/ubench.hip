// Micro-benchmarks that decide the scan-kernel design on gfx950: issue rates of v_exp_f32, v_pk_fma_f32,
// their mix inside one wave, split across waves of one SIMD, and a packed-FMA software exp2.
// Not product code.  Built by tools/run_ubench.py into tools/libubench.so; C ABI: ubench_run(id, ...).
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef float v2f __attribute__((ext_vector_type(2)));

#define ITER_BODY_BEGIN for (int it = 0; it < iters; ++it) {
#define ITER_BODY_END }

template <int MODE>
__global__ __launch_bounds__(256) void k_rate(float *out, int iters, float seed) {
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    float x[8];
    v2f p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { x[i] = seed + 0.01f * i + 1e-3f * threadIdx.x; p[i] = v2f{x[i], x[i] + 0.5f}; }
    const v2f ca = {0.999f, 0.998f}, cb = {0.001f, 0.002f};
    if (MODE == 0) {  // exp only: 8 independent chains
        ITER_BODY_BEGIN
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = __builtin_amdgcn_exp2f(-x[i]);
        ITER_BODY_END
    } else if (MODE == 1) {  // packed fma only
        ITER_BODY_BEGIN
#pragma unroll
        for (int i = 0; i < 8; ++i) p[i] = p[i] * ca + cb;
        ITER_BODY_END
    } else if (MODE == 2) {  // scalar fma only
        ITER_BODY_BEGIN
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = x[i] * 0.999f + 0.001f;
        ITER_BODY_END
    } else if (MODE == 3) {  // mix inside one wave: per iteration 8 exp + 16 packed fma (the scan's 1 : 2 ratio)
        ITER_BODY_BEGIN
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            x[i] = __builtin_amdgcn_exp2f(-x[i]);
            p[i] = p[i] * ca + cb;
            p[i] = p[i] * cb + ca;
        }
        ITER_BODY_END
    } else if (MODE == 4) {  // wave-split: even waves exp (8/iter), odd waves packed fma (16/iter)
        if (wave & 1) {
            ITER_BODY_BEGIN
#pragma unroll
            for (int i = 0; i < 8; ++i) { p[i] = p[i] * ca + cb; p[i] = p[i] * cb + ca; }
            ITER_BODY_END
        } else {
            ITER_BODY_BEGIN
#pragma unroll
            for (int i = 0; i < 8; ++i) x[i] = __builtin_amdgcn_exp2f(-x[i]);
            ITER_BODY_END
        }
    } else if (MODE == 5) {  // software exp2 on the packed FMA pipe (magic-number round, degree-5 poly, exponent add)
        const v2f magic = {12582912.f, 12582912.f};
        const v2f c5 = {1.3333558e-3f, 1.3333558e-3f}, c4 = {9.6181291e-3f, 9.6181291e-3f}, c3 = {5.5504109e-2f, 5.5504109e-2f},
                  c2 = {2.4022651e-1f, 2.4022651e-1f}, c1 = {6.9314718e-1f, 6.9314718e-1f}, c0 = {1.f, 1.f};
        ITER_BODY_BEGIN
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            v2f xx = -p[i];
            xx.x = fmaxf(xx.x, -126.f); xx.y = fmaxf(xx.y, -126.f);
            v2f t = xx + magic;
            v2f f = xx - (t - magic);
            v2f q = c5 * f + c4;
            q = q * f + c3; q = q * f + c2; q = q * f + c1; q = q * f + c0;
            uint32_t e0 = __float_as_uint(t.x) << 23, e1 = __float_as_uint(t.y) << 23;
            p[i].x = __uint_as_float(__float_as_uint(q.x) + e0);
            p[i].y = __uint_as_float(__float_as_uint(q.y) + e1);
        }
        ITER_BODY_END
    } else if (MODE == 6) {  // v_log_f32
        ITER_BODY_BEGIN
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = __builtin_amdgcn_logf(x[i] + 2.f);
        ITER_BODY_END
    } else if (MODE == 7) {  // v_rcp_f32
        ITER_BODY_BEGIN
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = __builtin_amdgcn_rcpf(x[i] + 1.f);
        ITER_BODY_END
    } else if (MODE == 8) {  // exp + scalar (unpacked) fma mix 1:2
        ITER_BODY_BEGIN
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            x[i] = __builtin_amdgcn_exp2f(-x[i]);
            p[i].x = p[i].x * 0.999f + 0.001f;
            p[i].y = p[i].y * 0.998f + 0.002f;
        }
        ITER_BODY_END
    }
    else if (MODE == 9) {  // v_mul_f32 with a DPP row_newbcast source, 8 independent
        ITER_BODY_BEGIN
#pragma unroll
        for (int i = 0; i < 8; ++i)
            x[i] = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(x[i]), 0x150 + 3, 0xf, 0xf, true)) * 0.999f;
        ITER_BODY_END
    } else if (MODE == 10) {  // v_fmac_f32_dpp, 8 independent accumulators
        ITER_BODY_BEGIN
#pragma unroll
        for (int i = 0; i < 8; ++i)
            asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(x[i]) : "v"(p[i].x), "v"(p[i].y));
        ITER_BODY_END
    } else if (MODE == 11) {  // replica of the scan core: 2 steps x 4 states = 8 x (mul, exp, mul_dpp, fma, fmac_dpp)
        float hh[4] = {x[0], x[1], x[2], x[3]}, aa[4] = {-x[4], -x[5], -x[6], -x[7]};
        float Bf = p[0].x, Cf = p[0].y, dv = p[1].x, du = p[1].y, y = 0.f;
        ITER_BODY_BEGIN
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            float yy;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float e = __builtin_amdgcn_exp2f(dv * aa[j]);
                const float bq = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(Bf), 0x150 + 2, 0xf, 0xf, true)) * du;
                hh[j] = __builtin_fmaf(e, hh[j], bq);
                if (j == 0) yy = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(Cf), 0x150 + 1, 0xf, 0xf, true)) * hh[0];
                else asm volatile("v_fmac_f32_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(yy) : "v"(Cf), "v"(hh[j]));
            }
            y += yy;
            dv += 1e-9f;
        }
        ITER_BODY_END
        x[0] = hh[0] + y; x[1] = hh[1]; x[2] = hh[2]; x[3] = hh[3];
    }
    else if (MODE == 12) {  // v_exp_f16, 8 independent (is the 16-bit transcendental any faster?)
        _Float16 hx[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) hx[i] = static_cast<_Float16>(x[i]);
        ITER_BODY_BEGIN
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_exp_f16 %0, %0" : "+v"(hx[i]));
        ITER_BODY_END
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = static_cast<float>(hx[i]);
    } else if (MODE == 13) {  // v_rcp_f16
        _Float16 hx[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) hx[i] = static_cast<_Float16>(x[i]);
        ITER_BODY_BEGIN
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_rcp_f16 %0, %0" : "+v"(hx[i]));
        ITER_BODY_END
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = static_cast<float>(hx[i]);
    } else if (MODE == 14) {  // v_pk_mul_f16 (packed 16-bit pipe)
        typedef _Float16 h2 __attribute__((ext_vector_type(2)));
        h2 hp[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) hp[i] = h2{static_cast<_Float16>(x[i]), static_cast<_Float16>(x[i] + 0.5f)};
        const h2 hc = {static_cast<_Float16>(0.999f), static_cast<_Float16>(0.998f)};
        ITER_BODY_BEGIN
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f16 %0, %0, %1" : "+v"(hp[i]) : "v"(hc));
        ITER_BODY_END
#pragma unroll
        for (int i = 0; i < 8; ++i) x[i] = static_cast<float>(hp[i].x) + static_cast<float>(hp[i].y);
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += x[i] + p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

extern "C" int ubench_launch(int mode, int blocks, int iters, float *out, void *stream) {
    hipStream_t st = static_cast<hipStream_t>(stream);
    dim3 g(blocks), b(256);
    switch (mode) {
        case 0: hipLaunchKernelGGL(k_rate<0>, g, b, 0, st, out, iters, 0.5f); break;
        case 1: hipLaunchKernelGGL(k_rate<1>, g, b, 0, st, out, iters, 0.5f); break;
        case 2: hipLaunchKernelGGL(k_rate<2>, g, b, 0, st, out, iters, 0.5f); break;
        case 3: hipLaunchKernelGGL(k_rate<3>, g, b, 0, st, out, iters, 0.5f); break;
        case 4: hipLaunchKernelGGL(k_rate<4>, g, b, 0, st, out, iters, 0.5f); break;
        case 5: hipLaunchKernelGGL(k_rate<5>, g, b, 0, st, out, iters, 0.5f); break;
        case 6: hipLaunchKernelGGL(k_rate<6>, g, b, 0, st, out, iters, 0.5f); break;
        case 7: hipLaunchKernelGGL(k_rate<7>, g, b, 0, st, out, iters, 0.5f); break;
        case 8: hipLaunchKernelGGL(k_rate<8>, g, b, 0, st, out, iters, 0.5f); break;
        case 9: hipLaunchKernelGGL(k_rate<9>, g, b, 0, st, out, iters, 0.5f); break;
        case 10: hipLaunchKernelGGL(k_rate<10>, g, b, 0, st, out, iters, 0.5f); break;
        case 11: hipLaunchKernelGGL(k_rate<11>, g, b, 0, st, out, iters, 0.5f); break;
        case 12: hipLaunchKernelGGL(k_rate<12>, g, b, 0, st, out, iters, 0.5f); break;
        case 13: hipLaunchKernelGGL(k_rate<13>, g, b, 0, st, out, iters, 0.5f); break;
        case 14: hipLaunchKernelGGL(k_rate<14>, g, b, 0, st, out, iters, 0.5f); break;
        default: return -1;
    }
    return hipGetLastError() == hipSuccess ? 0 : -5;
}
