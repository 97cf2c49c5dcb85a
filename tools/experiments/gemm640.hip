// Feasibility probe: C[M x 640] = A[M x K] * W[640 x K]^T, bf16 in / fp32 accumulate / bf16 out, full-row tiles
// (128 x 640 per workgroup) so that a row-wise epilogue (residual add + LayerNorm + modulate) could be fused later.
// 8 waves (2 x 4), wave tile 64 x 160 = 2 x 5 MFMA 32x32x16 blocks, BK = 32, LDS double buffer.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128, BN = 640, BK = 32, PITCH = BK * 2 + 16;   // 80 B per LDS row (16 B skew)
constexpr int A_BYTES = BM * PITCH, B_BYTES = BN * PITCH, STAGE = A_BYTES + B_BYTES;

__global__ __launch_bounds__(512) void gemm640_kernel(const uint16_t *A, const uint16_t *W, uint16_t *C, int M, int K,
                                                      int64_t a_stride, int64_t w_stride, int64_t c_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 2, wn = wave & 3;                 // wave tile origin: rows 64 wm, cols 160 wn
    const int m0 = blockIdx.x * BM;
    const int j = lane & 31, kh = lane >> 5;

    f32x16 acc[2][5];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 5; ++b) acc[a][b] = f32x16{};

    // global -> register staging: A 512 pieces (1 / thread), B 2560 pieces (5 / thread); piece = (row, 16-byte column)
    uint4 ra, rb[5];
    auto g_load = [&](int k0) {
        {
            const int row = tid >> 2, pc = tid & 3;
            int mr = m0 + row; mr = mr < M ? mr : M - 1;
            ra = *reinterpret_cast<const uint4 *>(A + mr * a_stride + k0 + pc * 8);
        }
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int piece = tid + 512 * i, row = piece >> 2, pc = piece & 3;
            rb[i] = *reinterpret_cast<const uint4 *>(W + row * w_stride + k0 + pc * 8);
        }
    };
    auto s_store = [&](int stage) {
        unsigned char *sa = smem + stage * STAGE, *sb = sa + A_BYTES;
        *reinterpret_cast<uint4 *>(sa + (tid >> 2) * PITCH + (tid & 3) * 16) = ra;
#pragma unroll
        for (int i = 0; i < 5; ++i) {
            const int piece = tid + 512 * i;
            *reinterpret_cast<uint4 *>(sb + (piece >> 2) * PITCH + (piece & 3) * 16) = rb[i];
        }
    };
    const int n_k = K / BK;
    g_load(0);
    s_store(0);
    __syncthreads();
    for (int kt = 0; kt < n_k; ++kt) {
        const int stage = kt & 1;
        if (kt + 1 < n_k) g_load((kt + 1) * BK);
        const unsigned char *sa = smem + stage * STAGE, *sb = sa + A_BYTES;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            bf16x8 af[2], bf[5];
#pragma unroll
            for (int a = 0; a < 2; ++a)
                af[a] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(sa + (wm * 64 + a * 32 + j) * PITCH + ks * 32 + kh * 16));
#pragma unroll
            for (int b = 0; b < 5; ++b)
                bf[b] = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(sb + (wn * 160 + b * 32 + j) * PITCH + ks * 32 + kh * 16));
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < 5; ++b) acc[a][b] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[a], bf[b], acc[a][b], 0, 0, 0);
        }
        if (kt + 1 < n_k) s_store(stage ^ 1);
        __syncthreads();
    }
    // plain epilogue: C layout col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 5; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * 64 + a * 32 + (r & 3) + 8 * (r >> 2) + 4 * kh, col = wn * 160 + b * 32 + j;
                if (row < M) {
                    const __bf16 h = static_cast<__bf16>(acc[a][b][r]);
                    uint16_t v; __builtin_memcpy(&v, &h, 2);
                    C[row * c_stride + col] = v;
                }
            }
}

extern "C" int gemm640(const void *A, const void *W, void *C, int M, int K, int64_t a_stride, int64_t w_stride, int64_t c_stride, void *stream) {
    static bool once = false;
    if (!once) { hipFuncSetAttribute(reinterpret_cast<const void *>(gemm640_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE); once = true; }
    hipLaunchKernelGGL(gemm640_kernel, dim3((M + BM - 1) / BM), dim3(512), 2 * STAGE, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const uint16_t *>(A), reinterpret_cast<const uint16_t *>(W), reinterpret_cast<uint16_t *>(C), M, K,
                       a_stride, w_stride, c_stride);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
