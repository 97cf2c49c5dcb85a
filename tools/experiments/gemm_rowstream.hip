// Experiment: C[M x 640] = A[M x K] * W[640 x K]^T (+ bias), bf16 / fp32 accumulate, "row-streaming" tiling:
//   workgroup = 4 waves = 128 tokens, ONE wave per SIMD with the whole 32 x 640 accumulator in registers (320 AGPR/VGPR),
//   A fragments streamed per wave straight from HBM (each used for 20 MFMAs), W staged through LDS in K-chunks of 64
//   (shared by the 4 waves), next chunk's W in flight in registers during the MFMAs.
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int N = 640, BK = 32, PITCH = BK * 2 + 16, WAVES = 4, TOK = 32, NB = N / 32, ADEPTH = 2;
constexpr int STG = N * (BK / 8) / (64 * WAVES);     // 16-byte pieces of a W chunk per thread: 640 * 4 / 256 = 10
constexpr int STAGE = N * PITCH;                     // 51 200 B, double-buffered

__global__ __launch_bounds__(64 * WAVES, 1) void gemm_rowstream_kernel(const uint16_t *A, const uint16_t *W, const float *bias, uint16_t *C,
                                                                      int M, int K, int64_t a_stride, int64_t w_stride, int64_t c_stride) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_w[];     // N * PITCH = 92 160 B
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, kh = lane >> 5;
    const int64_t m0 = (static_cast<int64_t>(blockIdx.x) * WAVES + wave) * TOK;
    int64_t mr = m0 + j; mr = mr < M ? mr : M - 1;
    const uint16_t *arow = A + mr * a_stride + kh * 8;

    f32x16 acc[NB];
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) acc[nb] = f32x16{};
    const int n_steps = K / 16;
    uint4 aq[ADEPTH];
#pragma unroll
    for (int d = 0; d < ADEPTH; ++d) aq[d] = *reinterpret_cast<const uint4 *>(arow + d * 16);
    uint4 stg[STG];
    auto w_load = [&](int c0) {
#pragma unroll
        for (int i = 0; i < STG; ++i) {
            const int piece = tid + 64 * WAVES * i, row = piece >> 2, pc = piece & 3;
            stg[i] = *reinterpret_cast<const uint4 *>(W + static_cast<int64_t>(row) * w_stride + c0 + pc * 8);
        }
    };
    auto w_store = [&](int stage) {
#pragma unroll
        for (int i = 0; i < STG; ++i) {
            const int piece = tid + 64 * WAVES * i, row = piece >> 2, pc = piece & 3;
            *reinterpret_cast<uint4 *>(s_w + stage * STAGE + row * PITCH + pc * 16) = stg[i];
        }
    };
    w_load(0);
    w_store(0);
    __syncthreads();
    if (BK < K) w_load(BK);
#pragma unroll 1
    for (int c0 = 0; c0 < K; c0 += BK) {
        const int stage = (c0 / BK) & 1;
        const unsigned char *sw = s_w + stage * STAGE;
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const int s = c0 / 16 + ks;
            const bf16x8 a = __builtin_bit_cast(bf16x8, aq[ks % ADEPTH]);
            if (s + ADEPTH < n_steps) aq[ks % ADEPTH] = *reinterpret_cast<const uint4 *>(arow + (s + ADEPTH) * 16);
#pragma unroll
            for (int nb = 0; nb < NB; ++nb) {
                const bf16x8 b = __builtin_bit_cast(bf16x8, *reinterpret_cast<const uint4 *>(sw + (nb * 32 + j) * PITCH + ks * 32 + kh * 16));
                acc[nb] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[nb], 0, 0, 0);
            }
        }
        if (c0 + BK < K) w_store(stage ^ 1);       // chunk c+1 (in registers since the last barrier) -> the idle buffer
        __syncthreads();
        if (c0 + 2 * BK < K) w_load(c0 + 2 * BK);
    }
#pragma unroll
    for (int nb = 0; nb < NB; ++nb) {
        const int n = nb * 32 + j;
        const float bv = bias ? bias[n] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t m = m0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (m < M) {
                const __bf16 h = static_cast<__bf16>(acc[nb][r] + bv);
                uint16_t v; __builtin_memcpy(&v, &h, 2);
                C[m * c_stride + n] = v;
            }
        }
    }
}

extern "C" int gemm_rowstream(const void *A, const void *W, const void *bias, void *C, int M, int K, int64_t a_stride, int64_t w_stride,
                              int64_t c_stride, void *stream) {
    static bool once = false;
    if (!once) { (void)hipFuncSetAttribute(reinterpret_cast<const void *>(gemm_rowstream_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 2 * STAGE); once = true; }
    hipLaunchKernelGGL(gemm_rowstream_kernel, dim3((M + WAVES * TOK - 1) / (WAVES * TOK)), dim3(64 * WAVES), 2 * STAGE, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const uint16_t *>(A), reinterpret_cast<const uint16_t *>(W), reinterpret_cast<const float *>(bias),
                       reinterpret_cast<uint16_t *>(C), M, K, a_stride, w_stride, c_stride);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
