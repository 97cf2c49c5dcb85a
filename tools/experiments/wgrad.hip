// Experiment: weight-gradient GEMM  dW[N x K'] = dY[M x N]^T * X[M x K']  (reduction over the M = 65 536 tokens), bf16 in,
// fp32 accumulate.  The library runs this shape at 0.16-0.5 PF/s.  Split-K over token ranges; a workgroup (4 waves, one per
// SIMD) owns a 320 x 256 (or 256 x 320) output tile with the whole accumulator in registers (20 MFMA 32x32 blocks per wave)
// and walks its token range in chunks of 32 tokens staged through LDS (16-byte coalesced loads in, TRANSPOSED 2-byte reads
// out: both MFMA operands have the token index as their k dimension).  fp32 partial tiles -> workspace -> summed in a fixed
// order by a finishing kernel.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int CH = 32;            // tokens per LDS stage (2 MFMA k-steps)

template <int BI, int BJ>         // 32x32 blocks per wave along N (dY columns) and K' (X columns); wave grid 2 x 2
__global__ __launch_bounds__(256, 1) void wgrad_kernel(const uint16_t *dY, const uint16_t *X, float *part, int M, int N, int K,
                                                       int64_t dy_stride, int64_t x_stride, int tokens_per_split) {
    constexpr int TN = 2 * BI * 32, TK = 2 * BJ * 32;
    constexpr int PA = TN * 2 + 16, PB = TK * 2 + 16;                 // LDS row pitches (bytes)
    constexpr int STAGE = CH * (PA + PB);
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wi = wave >> 1, wj = wave & 1;
    const int j = lane & 31, kh = lane >> 5;
    const int n0 = blockIdx.x * TN, k0 = blockIdx.y * TK;
    const int64_t m_begin = static_cast<int64_t>(blockIdx.z) * tokens_per_split;
    const int n_chunks = tokens_per_split / CH;

    f32x16 acc[BI][BJ];
#pragma unroll
    for (int a = 0; a < BI; ++a)
#pragma unroll
        for (int b = 0; b < BJ; ++b) acc[a][b] = f32x16{};

    constexpr int PIECES_A = CH * TN / 8, PIECES_B = CH * TK / 8;     // 16-byte pieces per stage
    constexpr int NA = PIECES_A / 256, NB = PIECES_B / 256;
    uint4 ra[NA], rb[NB];
    auto g_load = [&](int64_t m0) {
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int piece = tid + 256 * i, row = piece / (TN / 8), pc = piece % (TN / 8);
            ra[i] = *reinterpret_cast<const uint4 *>(dY + (m0 + row) * dy_stride + n0 + pc * 8);
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int piece = tid + 256 * i, row = piece / (TK / 8), pc = piece % (TK / 8);
            rb[i] = *reinterpret_cast<const uint4 *>(X + (m0 + row) * x_stride + k0 + pc * 8);
        }
    };
    auto s_store = [&](int stage) {
        unsigned char *sa = smem + stage * STAGE, *sb = sa + CH * PA;
#pragma unroll
        for (int i = 0; i < NA; ++i) {
            const int piece = tid + 256 * i, row = piece / (TN / 8), pc = piece % (TN / 8);
            *reinterpret_cast<uint4 *>(sa + row * PA + pc * 16) = ra[i];
        }
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            const int piece = tid + 256 * i, row = piece / (TK / 8), pc = piece % (TK / 8);
            *reinterpret_cast<uint4 *>(sb + row * PB + pc * 16) = rb[i];
        }
    };
    // transposed fragments with the hardware transpose read.  ds_read_b64_tr_b16 (measured, tr_probe.hip): inside a 16-lane
    // group lane p supplies the address of 4 contiguous elements S_p[0..3]; lane i receives { S_{4e + (i >> 2)}[i & 3] }_e=0..3.
    // With lane p pointing at row t0 + (p >> 2), columns c0 + 4 (p & 3) .. +3 of the row-major [token][column] image, lane i
    // gets column c0 + i for tokens t0 .. t0+3: two reads = the 8 consecutive tokens of an MFMA A / B fragment.
    typedef unsigned u2v __attribute__((ext_vector_type(2)));
    const int p16 = lane & 15, cgrp = (lane >> 4) & 1;
    const unsigned lane_a = static_cast<unsigned>(((p16 >> 2) + 8 * kh) * PA + (16 * cgrp + 4 * (p16 & 3)) * 2);
    const unsigned lane_b = static_cast<unsigned>(((p16 >> 2) + 8 * kh) * PB + (16 * cgrp + 4 * (p16 & 3)) * 2);
    const unsigned smem_base = static_cast<unsigned>(reinterpret_cast<uintptr_t>(smem));
#define TR_READ(dst, addr, off) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(off) : "memory")
    g_load(m_begin);
    s_store(0);
    __syncthreads();
    if (n_chunks > 1) g_load(m_begin + CH);
#pragma unroll 1
    for (int c = 0; c < n_chunks; ++c) {
        const int stage = c & 1;
        const unsigned va = smem_base + stage * STAGE + lane_a + (wi * BI * 32) * 2;
        const unsigned vb = smem_base + stage * STAGE + CH * PA + lane_b + (wj * BJ * 32) * 2;
#pragma unroll
        for (int ks = 0; ks < CH / 16; ++ks) {
            u2v ar[BI][2], br[BJ][2];
#pragma unroll
            for (int a = 0; a < BI; ++a) {
                TR_READ(ar[a][0], va, a * 64 + (ks * 16) * PA);
                TR_READ(ar[a][1], va, a * 64 + (ks * 16 + 4) * PA);
            }
#pragma unroll
            for (int b = 0; b < BJ; ++b) {
                TR_READ(br[b][0], vb, b * 64 + (ks * 16) * PB);
                TR_READ(br[b][1], vb, b * 64 + (ks * 16 + 4) * PB);
            }
            // the compiler does not count inline-asm LDS reads: one explicit wait that every fragment register depends on
            if constexpr (BI == 5)
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ar[0][0]), "+v"(ar[0][1]), "+v"(ar[1][0]), "+v"(ar[1][1]), "+v"(ar[2][0]), "+v"(ar[2][1]),
                             "+v"(ar[3][0]), "+v"(ar[3][1]), "+v"(ar[4][0]), "+v"(ar[4][1]), "+v"(br[0][0]), "+v"(br[0][1]), "+v"(br[1][0]),
                             "+v"(br[1][1]), "+v"(br[2][0]), "+v"(br[2][1]), "+v"(br[3][0]), "+v"(br[3][1]));
            else
                asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ar[0][0]), "+v"(ar[0][1]), "+v"(ar[1][0]), "+v"(ar[1][1]), "+v"(ar[2][0]), "+v"(ar[2][1]),
                             "+v"(ar[3][0]), "+v"(ar[3][1]), "+v"(br[0][0]), "+v"(br[0][1]), "+v"(br[1][0]), "+v"(br[1][1]), "+v"(br[2][0]),
                             "+v"(br[2][1]), "+v"(br[3][0]), "+v"(br[3][1]), "+v"(br[4][0]), "+v"(br[4][1]));
            bf16x8 af[BI], bf[BJ];
#pragma unroll
            for (int a = 0; a < BI; ++a) af[a] = __builtin_bit_cast(bf16x8, make_uint4(ar[a][0][0], ar[a][0][1], ar[a][1][0], ar[a][1][1]));
#pragma unroll
            for (int b = 0; b < BJ; ++b) bf[b] = __builtin_bit_cast(bf16x8, make_uint4(br[b][0][0], br[b][0][1], br[b][1][0], br[b][1][1]));
#pragma unroll
            for (int a = 0; a < BI; ++a)
#pragma unroll
                for (int b = 0; b < BJ; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
        if (c + 1 < n_chunks) s_store(stage ^ 1);
        __syncthreads();
        if (c + 2 < n_chunks) g_load(m_begin + static_cast<int64_t>(c + 2) * CH);
    }
    // partial tile -> workspace [split][N][K]; D layout: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
    float *out = part + static_cast<int64_t>(blockIdx.z) * N * K;
#pragma unroll
    for (int a = 0; a < BI; ++a)
#pragma unroll
        for (int b = 0; b < BJ; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int n = n0 + (wi * BI + a) * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh;
                const int k = k0 + (wj * BJ + b) * 32 + j;
                out[static_cast<int64_t>(n) * K + k] = acc[a][b][r];
            }
}

__global__ void wgrad_finish(const float *part, float *dW, int total, int splits) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    float acc = 0.f;
    for (int s = 0; s < splits; ++s) acc += part[static_cast<int64_t>(s) * total + i];
    dW[i] = acc;
}

extern "C" int wgrad(const void *dY, const void *X, void *part, void *dW, int M, int N, int K, int64_t dy_stride, int64_t x_stride,
                     int splits, void *stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    const int tps = M / splits;
    if (N % 320 == 0 && K % 256 == 0) {
        constexpr int smem = 2 * CH * ((320 * 2 + 16) + (256 * 2 + 16));
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(wgrad_kernel<5, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        hipLaunchKernelGGL((wgrad_kernel<5, 4>), dim3(N / 320, K / 256, splits), dim3(256), smem, s, reinterpret_cast<const uint16_t *>(dY),
                           reinterpret_cast<const uint16_t *>(X), reinterpret_cast<float *>(part), M, N, K, dy_stride, x_stride, tps);
    } else if (N % 256 == 0 && K % 320 == 0) {
        constexpr int smem = 2 * CH * ((256 * 2 + 16) + (320 * 2 + 16));
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(wgrad_kernel<4, 5>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
        hipLaunchKernelGGL((wgrad_kernel<4, 5>), dim3(N / 256, K / 320, splits), dim3(256), smem, s, reinterpret_cast<const uint16_t *>(dY),
                           reinterpret_cast<const uint16_t *>(X), reinterpret_cast<float *>(part), M, N, K, dy_stride, x_stride, tps);
    } else return -2;
    hipLaunchKernelGGL(wgrad_finish, dim3((N * K + 255) / 256), dim3(256), 0, s, reinterpret_cast<const float *>(part), reinterpret_cast<float *>(dW), N * K, splits);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
