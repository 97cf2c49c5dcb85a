// How fast can 168 MB of (65536 x 1280) bf16 be WRITTEN with the store patterns of the dt_proj kernels?
#include <hip/hip_runtime.h>
#include <stdint.h>
constexpr int M = 65536, N = 1280;
// P0: linear 16-byte stores
__global__ void p0(uint16_t *o) {
    const int64_t i = (static_cast<int64_t>(blockIdx.x) * 256 + threadIdx.x) * 8;
    if (i < static_cast<int64_t>(M) * N) *reinterpret_cast<uint4 *>(o + i) = make_uint4(1, 2, 3, 4);
}
// P1: wave = 32 tokens x 64 channels, 4-byte stores, 16 instructions (2 rows x 128 B each), 4 iterations per wave
__global__ void p1(uint16_t *o) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, j = lane & 31, kh = lane >> 5, d0 = blockIdx.x * 64;
    for (int it = 0; it < 4; ++it) {
        const int64_t m0 = (static_cast<int64_t>(blockIdx.y) * 16 + it * 4 + wave) * 32;
        for (int r = 0; r < 16; ++r) {
            const int64_t m = m0 + (r & 3) + 8 * (r >> 2) + 4 * kh;
            *reinterpret_cast<uint32_t *>(o + m * N + d0 + 2 * j) = 0x3f803f80u;
        }
    }
}
// P2: wave = 16 tokens x 128 channels, 16-byte stores, 4 instructions (4 rows x 256 B each), 4 iterations per wave
__global__ void p2(uint16_t *o) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, i16 = lane & 15, g = lane >> 4, c0 = blockIdx.x * 128 + 8 * i16;
    for (int it = 0; it < 4; ++it) {
        const int64_t m0 = (static_cast<int64_t>(blockIdx.y) * 16 + it * 4 + wave) * 16;
        for (int r = 0; r < 4; ++r) *reinterpret_cast<uint4 *>(o + (m0 + 4 * g + r) * N + c0) = make_uint4(1, 2, 3, 4);
    }
}
// P3: wave = 4 whole rows per step (lane -> 16-byte piece of a row; 160 pieces per row), 16 steps
__global__ void p3(uint16_t *o) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t row0 = (static_cast<int64_t>(blockIdx.x) * 4 + wave) * 64;
    for (int r = 0; r < 64; ++r)
        for (int pc = lane; pc < 160; pc += 64) *reinterpret_cast<uint4 *>(o + (row0 + r) * N + pc * 8) = make_uint4(1, 2, 3, 4);
}
extern "C" void run(int which, void *o, void *stream) {
    hipStream_t s = static_cast<hipStream_t>(stream);
    uint16_t *p = reinterpret_cast<uint16_t *>(o);
    if (which == 0) hipLaunchKernelGGL(p0, dim3(M * (N / 8) / 256), dim3(256), 0, s, p);
    if (which == 1) hipLaunchKernelGGL(p1, dim3(N / 64, M / (32 * 16)), dim3(256), 0, s, p);
    if (which == 2) hipLaunchKernelGGL(p2, dim3(N / 128, M / (16 * 16)), dim3(256), 0, s, p);
    if (which == 3) hipLaunchKernelGGL(p3, dim3(M / 256), dim3(256), 0, s, p);
}
