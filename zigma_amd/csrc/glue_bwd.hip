// Backward of the block's elementwise glue — modulate() and the gated branch add (reference model_zigma.py:53-54,441-458) — in one
// pass, gfx950.  C ABI: zigma_scale_reduce_bwd.
//
//   out[r, :] = dy[r, :] * s[b, :] (+ s_add)                                  b = r / rows_per_batch
//   r1[b, c, :] = sum over the rows of chunk c of sample b of dy[r, :] * a[r, :]
//   r2[b, c, :] = sum over the rows of chunk c of sample b of dy[r, :]                       (optional)
//
// modulate backward:  a = x, s = scale, s_add = 1  ->  out = dx, sum_c r1 = dscale, sum_c r2 = dshift;
// gated add backward: a = branch, s = gate         ->  out = dbranch, sum_c r1 = dgate.
// The autograd graph of these ops is 3-4 elementwise / reduction launches with 6-8 passes over (B, L, E) tensors; here dy and a
// are read once and out written once.  The per-chunk partial sums (chunks of 64 rows, fp32) are summed by the caller — a fixed
// order, bit-reproducible, no float atomics.
//
// One workgroup = 64 rows x all columns, walked in 128-column slabs: 16 groups of 16 lanes, a group takes 4 of the 64 rows, a lane
// 16 bytes of the slab; the 16 groups' partial sums of a slab meet in LDS.  bf16; cols % 128 == 0; rows_per_batch % 64 == 0; 16-byte aligned rows.
#include "zigma_common.h"

namespace zigma {

constexpr int kGbRows = 64, kGbMaxIters = 64;

template <bool HAS_R2>
__global__ __launch_bounds__(256) void scale_reduce_bwd_kernel(const zigma_glue_bwd_params_t p) {
    __shared__ float s_part[HAS_R2 ? 2 : 1][16][128];
    const int tid = threadIdx.x, grp = tid >> 4, l16 = tid & 15;
    const int64_t row0 = static_cast<int64_t>(blockIdx.x) * kGbRows;
    const int b = static_cast<int>(row0 / p.rows_per_batch);
    const uint16_t *dy = reinterpret_cast<const uint16_t *>(p.dy), *a = reinterpret_cast<const uint16_t *>(p.a);
    const uint16_t *s = reinterpret_cast<const uint16_t *>(p.s) + static_cast<int64_t>(b) * p.s_batch_stride;
    uint16_t *out = reinterpret_cast<uint16_t *>(p.out);
    const int chunk = static_cast<int>((row0 - static_cast<int64_t>(b) * p.rows_per_batch) / kGbRows);
    const int n_chunks = p.rows_per_batch / kGbRows;
    const int64_t r_out = (static_cast<int64_t>(b) * n_chunks + chunk) * p.cols;
    const int iters = p.cols / 128;
    // a pass per 128-column slab (16 lanes x 16 bytes): the 64 rows are walked 16 at a time, 4 rows per group, all loads of a group's
    // rows in flight before the first use; every element is read once
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        const int c0 = (it * 16 + l16) * 8;
        float sv[8], acc1[8], acc2[8];
        {
            const uint4 q = *reinterpret_cast<const uint4 *>(s + c0);
            const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                sv[2 * i] = __uint_as_float(w[i] << 16) + p.s_add;
                sv[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u) + p.s_add;
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) { acc1[i] = 0.f; acc2[i] = 0.f; }
        }
        uint4 qd[kGbRows / 16], qa[kGbRows / 16];
#pragma unroll
        for (int rr = 0; rr < kGbRows / 16; ++rr) {
            const int64_t r = row0 + rr * 16 + grp;
            qd[rr] = *reinterpret_cast<const uint4 *>(dy + r * p.dy_row_stride + c0);
            qa[rr] = *reinterpret_cast<const uint4 *>(a + r * p.a_row_stride + c0);
        }
#pragma unroll
        for (int rr = 0; rr < kGbRows / 16; ++rr) {
            const int64_t r = row0 + rr * 16 + grp;
            const uint32_t wd[4] = {qd[rr].x, qd[rr].y, qd[rr].z, qd[rr].w}, wa[4] = {qa[rr].x, qa[rr].y, qa[rr].z, qa[rr].w};
            uint32_t wo[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float d0 = __uint_as_float(wd[i] << 16), d1 = __uint_as_float(wd[i] & 0xffff0000u);
                const float a0 = __uint_as_float(wa[i] << 16), a1 = __uint_as_float(wa[i] & 0xffff0000u);
                acc1[2 * i] = __builtin_fmaf(d0, a0, acc1[2 * i]);
                acc1[2 * i + 1] = __builtin_fmaf(d1, a1, acc1[2 * i + 1]);
                if (HAS_R2) { acc2[2 * i] += d0; acc2[2 * i + 1] += d1; }
                wo[i] = static_cast<uint32_t>(from_float<BF16>(d0 * sv[2 * i])) | (static_cast<uint32_t>(from_float<BF16>(d1 * sv[2 * i + 1])) << 16);
            }
            if (out) *reinterpret_cast<uint4 *>(out + r * p.out_row_stride + c0) = make_uint4(wo[0], wo[1], wo[2], wo[3]);
        }
        // the 16 row groups meet in LDS; threads 0..127 own one column of the slab each
        __syncthreads();                                                   // (previous slab's sums have been read)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            s_part[0][grp][l16 * 8 + i] = acc1[i];
            if (HAS_R2) s_part[HAS_R2 ? 1 : 0][grp][l16 * 8 + i] = acc2[i];
        }
        __syncthreads();
        if (tid < 128) {
            float t1 = 0.f, t2 = 0.f;
#pragma unroll
            for (int g = 0; g < 16; ++g) {
                t1 += s_part[0][g][tid];
                if (HAS_R2) t2 += s_part[HAS_R2 ? 1 : 0][g][tid];
            }
            reinterpret_cast<float *>(p.r1)[r_out + it * 128 + tid] = t1;
            if (HAS_R2) reinterpret_cast<float *>(p.r2)[r_out + it * 128 + tid] = t2;
        }
    }
}

}  // namespace zigma

using namespace zigma;

extern "C" int zigma_scale_reduce_bwd(const zigma_glue_bwd_params_t *pp, void *stream_) {
    if (!pp) return ZIGMA_ERR_NULL;
    (void)hipGetLastError();
    const zigma_glue_bwd_params_t &p = *pp;
    if (p.rows < 0 || p.cols < 1 || p.rows_per_batch < 1) return ZIGMA_ERR_SHAPE;
    if (p.flags != 0) return ZIGMA_ERR_UNSUPPORTED;
    if (p.rows == 0) return ZIGMA_OK;
    if (!p.dy || !p.a || !p.s || !p.r1) return ZIGMA_ERR_NULL;
    if (p.dtype != ZIGMA_BF16) return ZIGMA_ERR_DTYPE;
    if (p.cols % 128 != 0 || p.cols > 128 * kGbMaxIters || p.rows_per_batch % kGbRows != 0 || p.rows % p.rows_per_batch != 0) return ZIGMA_ERR_SHAPE;
    auto bad = [](const void *q, int64_t st) { return q && (reinterpret_cast<uintptr_t>(q) % 16 != 0 || st % 8 != 0); };
    if (bad(p.dy, p.dy_row_stride) || bad(p.a, p.a_row_stride) || bad(p.out, p.out_row_stride) || bad(p.s, p.s_batch_stride)) return ZIGMA_ERR_STRIDE;
    const dim3 grid(static_cast<unsigned>(p.rows / kGbRows)), block(256);
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (p.r2) hipLaunchKernelGGL((scale_reduce_bwd_kernel<true>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((scale_reduce_bwd_kernel<false>), grid, block, 0, stream, p);
    set_last_kernel("scale_reduce_bwd");
    return check_launch();
}
