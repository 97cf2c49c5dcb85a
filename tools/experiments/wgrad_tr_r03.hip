// Weight gradient of a projection: dW = dY^T X, contraction over the TOKENS (rows of both row-major operands), on the matrix cores, gfx950.
// C ABI: zigma_wgrad_fwd (+ zigma_wgrad_workspace_bytes).
//
// What autograd's linear backward computes for the weights of in_proj / out_proj / to_q / to_out (reference: F.linear in
// mamba_simple.py:290-294, selective_scan_interface.py:365, model_zigma.py:104-135, differentiated by torch).  dY (m, n) and X (m, k) both
// have the contraction index as their ROW index: the MFMA operands (8 consecutive contraction values per lane) are columns of the tiles.
//   workgroup = 4 waves, output tile 256 (n) x 128 (k), wave tile 128 x 64 = 8 x 4 blocks of v_mfma_f32_16x16x32_bf16 (128 accumulators);
//   the token range of the workgroup (m / splits rows: the tokens are ALSO the parallel dimension — a 2560 x 640 output is only 50 tiles) is
//   walked in steps of 32 rows: the two row-major operand tiles [32][256] and [32][128] go to LDS as they lie in HBM (16-byte pieces,
//   256 / 512 contiguous bytes per row), double-buffered, next step's loads in flight during the products;
//   fragments come out of LDS TRANSPOSED by ds_read_b64_tr_b16: within a 16-lane group lane p supplies the address of 4 consecutive
//   16-bit elements, lane l receives element (l & 3) of the pieces 4 j + (l >> 2), j = 0..3 (probed: tools/ubench4) — with piece p at
//   tile[row 4 g' + (p >> 2)][col 4 (p & 3)] a lane gets 4 consecutive ROWS of its column: two reads = the 8 contraction values of a lane;
//   the row pitch is padded by 32 bytes (4 rows x 32 bytes of a group fall into distinct banks);
//   fp32 partial tiles per token split -> workspace; zigma_wgrad_finish adds the splits in a fixed order and rounds to bf16.
#include "zigma_common.h"

namespace zigma {

typedef __bf16 wg_bf16x8 __attribute__((ext_vector_type(8)));
typedef float wg_f32x4 __attribute__((ext_vector_type(4)));

constexpr int kWgTN = 256, kWgTK = 128, kWgBM = 32;
constexpr int kWgPitchA = kWgTN * 2 + 32, kWgPitchB = kWgTK * 2 + 32;         // bytes per tile row
constexpr int kWgStage = kWgBM * (kWgPitchA + kWgPitchB);

__device__ __forceinline__ uint2 lds_tr_read(unsigned addr) {
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(v) : "v"(addr));
    return v;
}

__global__ __launch_bounds__(256, 2) void wgrad_kernel(const zigma_wgrad_params_t p, float *ws) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * kWgStage];
    typedef __attribute__((address_space(3))) unsigned char *lds_ptr_t;
    const unsigned lds0 = static_cast<unsigned>(reinterpret_cast<uintptr_t>((lds_ptr_t)(smem)));
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wn = wave >> 1, wk = wave & 1;                       // wave tile: rows (n) 128 wn .., cols (k) 64 wk ..
    const int n0 = blockIdx.x * kWgTN, k0 = blockIdx.y * kWgTK;
    const int64_t rows_per_split = p.m / p.splits, m_begin = blockIdx.z * rows_per_split;
    const int steps = static_cast<int>(rows_per_split / kWgBM);
    const uint16_t *A = reinterpret_cast<const uint16_t *>(p.dy) + m_begin * p.dy_row_stride + n0;
    const uint16_t *B = reinterpret_cast<const uint16_t *>(p.x) + m_begin * p.x_row_stride + k0;
    // staging: A tile 32 rows x 32 pieces of 16 bytes = 1024 pieces (4 per thread), B tile 32 x 16 = 512 pieces (2 per thread)
    uint4 ra[4], rb[2];
    auto fetch = [&](int step) {
        const uint16_t *a = A + static_cast<int64_t>(step) * kWgBM * p.dy_row_stride, *b = B + static_cast<int64_t>(step) * kWgBM * p.x_row_stride;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int piece = q * 256 + tid, row = piece >> 5, pc = piece & 31;
            ra[q] = *reinterpret_cast<const uint4 *>(a + static_cast<int64_t>(row) * p.dy_row_stride + pc * 8);
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int piece = q * 256 + tid, row = piece >> 4, pc = piece & 15;
            rb[q] = *reinterpret_cast<const uint4 *>(b + static_cast<int64_t>(row) * p.x_row_stride + pc * 8);
        }
    };
    auto stash = [&](int buf) {
        unsigned char *sa = smem + buf * kWgStage, *sb = sa + kWgBM * kWgPitchA;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int piece = q * 256 + tid, row = piece >> 5, pc = piece & 31;
            *reinterpret_cast<uint4 *>(sa + row * kWgPitchA + pc * 16) = ra[q];
        }
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int piece = q * 256 + tid, row = piece >> 4, pc = piece & 15;
            *reinterpret_cast<uint4 *>(sb + row * kWgPitchB + pc * 16) = rb[q];
        }
    };
    // transposed fragment addresses: lane -> piece p = lane & 15 of its 16-lane group g = lane >> 4: row 8 g + 4 r + (p >> 2), col 4 (p & 3)
    const int pp = lane & 15, g = lane >> 4;
    const unsigned fa = static_cast<unsigned>((8 * g + (pp >> 2)) * kWgPitchA + (wn * 128 + 4 * (pp & 3)) * 2);
    const unsigned fb = static_cast<unsigned>(kWgBM * kWgPitchA + (8 * g + (pp >> 2)) * kWgPitchB + (wk * 64 + 4 * (pp & 3)) * 2);

    wg_f32x4 acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = wg_f32x4{0.f, 0.f, 0.f, 0.f};

    fetch(0);
    stash(0);
    __syncthreads();
#pragma unroll 1
    for (int s = 0; s < steps; ++s) {
        const int buf = s & 1;
        if (s + 1 < steps) fetch(s + 1);
        const unsigned base = lds0 + buf * kWgStage;
        // (the transposed reads are inline asm: hipcc does not count them — every use of their results is tied to an explicit wait)
        uint2 blo[4], bhi[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) { blo[j] = lds_tr_read(base + fb + j * 32); bhi[j] = lds_tr_read(base + fb + j * 32 + 4 * kWgPitchB); }
        uint2 lo = lds_tr_read(base + fa), hi = lds_tr_read(base + fa + 4 * kWgPitchA);
        asm volatile("s_waitcnt lgkmcnt(0)"
                     : "+v"(blo[0]), "+v"(bhi[0]), "+v"(blo[1]), "+v"(bhi[1]), "+v"(blo[2]), "+v"(bhi[2]), "+v"(blo[3]), "+v"(bhi[3]), "+v"(lo), "+v"(hi));
        wg_bf16x8 bf[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) bf[j] = __builtin_bit_cast(wg_bf16x8, make_uint4(blo[j].x, blo[j].y, bhi[j].x, bhi[j].y));
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            const wg_bf16x8 af = __builtin_bit_cast(wg_bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
            if (i + 1 < 8) {            // the next block's fragment is on its way during this block's 4 products
                lo = lds_tr_read(base + fa + (i + 1) * 32);
                hi = lds_tr_read(base + fa + (i + 1) * 32 + 4 * kWgPitchA);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bf[j], acc[i][j], 0, 0, 0);
            if (i + 1 < 8) asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(lo), "+v"(hi));
        }
        if (s + 1 < steps) stash(buf ^ 1);
        __syncthreads();
    }
    // fp32 partial tile of this token split: ws[split][n][k]; lane -> column k0 + 64 wk + 16 j + (lane & 15), rows n0 + 128 wn + 16 i + 4 g + r
    float *out = ws + (static_cast<int64_t>(blockIdx.z) * p.n + n0 + wn * 128) * p.k + k0 + wk * 64;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) out[static_cast<int64_t>(16 * i + 4 * g + r) * p.k + 16 * j + (lane & 15)] = acc[i][j][r];
}

__global__ void wgrad_finish_kernel(const zigma_wgrad_params_t p, const float *ws) {
    const int64_t i = (static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x) * 4, total = static_cast<int64_t>(p.n) * p.k;
    if (i >= total) return;
    wg_f32x4 acc = *reinterpret_cast<const wg_f32x4 *>(ws + i);
    for (int s = 1; s < p.splits; ++s) acc += *reinterpret_cast<const wg_f32x4 *>(ws + s * total + i);
    const int64_t row = i / p.k, col = i % p.k;
    if (p.flags & 1) {            // transposed result: dw (k, n) — the caller swapped the operands to meet the tile shape
        uint16_t *o = reinterpret_cast<uint16_t *>(p.dw) + col * p.dw_row_stride + row;
#pragma unroll
        for (int q = 0; q < 4; ++q) o[q * p.dw_row_stride] = from_float<BF16>(acc[q]);
        return;
    }
    uint16_t *o = reinterpret_cast<uint16_t *>(p.dw) + row * p.dw_row_stride + col;
    *reinterpret_cast<uint2 *>(o) = make_uint2(static_cast<uint32_t>(from_float<BF16>(acc[0])) | (static_cast<uint32_t>(from_float<BF16>(acc[1])) << 16),
                                               static_cast<uint32_t>(from_float<BF16>(acc[2])) | (static_cast<uint32_t>(from_float<BF16>(acc[3])) << 16));
}

}  // namespace zigma

using namespace zigma;

static int wgrad_check(const zigma_wgrad_params_t &p) {
    if (p.m < 0 || p.n < 0 || p.k < 0 || p.splits < 1) return ZIGMA_ERR_SHAPE;
    if (p.flags & ~1) return ZIGMA_ERR_UNSUPPORTED;
    if (p.dtype != ZIGMA_BF16) return ZIGMA_ERR_DTYPE;
    if (p.n % kWgTN != 0 || p.k % kWgTK != 0 || p.m % (static_cast<int64_t>(p.splits) * kWgBM) != 0 || p.splits > 65535) return ZIGMA_ERR_SHAPE;
    return ZIGMA_OK;
}

extern "C" int64_t zigma_wgrad_workspace_bytes(const zigma_wgrad_params_t *p) {
    if (!p || wgrad_check(*p) != ZIGMA_OK) return 0;
    return static_cast<int64_t>(p->splits) * p->n * p->k * static_cast<int64_t>(sizeof(float));
}

extern "C" int zigma_wgrad_fwd(const zigma_wgrad_params_t *pp, void *stream_) {
    if (!pp) return ZIGMA_ERR_NULL;
    (void)hipGetLastError();
    const zigma_wgrad_params_t &p = *pp;
    const int st = wgrad_check(p);
    if (st != ZIGMA_OK) return st;
    if (p.n == 0 || p.k == 0) return ZIGMA_OK;
    if (p.m == 0 || !p.dy || !p.x || !p.dw) return ZIGMA_ERR_NULL;
    if (!p.workspace || p.workspace_bytes < zigma_wgrad_workspace_bytes(pp) || reinterpret_cast<uintptr_t>(p.workspace) % 16 != 0) return ZIGMA_ERR_NULL;
    auto mis = [](const void *q, int64_t rs, int al) { return reinterpret_cast<uintptr_t>(q) % al != 0 || rs % (al / 2) != 0; };
    if (mis(p.dy, p.dy_row_stride, 16) || mis(p.x, p.x_row_stride, 16) || mis(p.dw, p.dw_row_stride, (p.flags & 1) ? 2 : 8)) return ZIGMA_ERR_STRIDE;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    float *ws = reinterpret_cast<float *>(p.workspace);
    hipLaunchKernelGGL(wgrad_kernel, dim3(p.n / kWgTN, p.k / kWgTK, p.splits), dim3(256), 0, stream, p, ws);
    const int64_t quads = static_cast<int64_t>(p.n) * p.k / 4;
    hipLaunchKernelGGL(wgrad_finish_kernel, dim3(static_cast<unsigned>((quads + 255) / 256)), dim3(256), 0, stream, p, ws);
    set_last_kernel("wgrad_mfma_tr");
    return check_launch();
}
