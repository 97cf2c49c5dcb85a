// Probe (not product code): what does ds_read_b64_tr_b16 return?  LDS holds lds[i] = i (16-bit); every lane passes its own byte address;
// the 4 x 16 bits each lane receives are written out.  Address patterns: 0: lane * 8;  1: (lane & 15) * 32 + (lane >> 4) * 8
// (16 rows of 32 bytes, lane group -> 8-byte column);  2: (lane >> 4) * 128 + (lane & 15) * 8
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ void probe(uint16_t *out, int pattern) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[2048];
    for (int i = threadIdx.x; i < 2048; i += 64) lds[i] = static_cast<uint16_t>(i);
    __syncthreads();
    const int lane = threadIdx.x;
    unsigned addr = pattern == 0 ? lane * 8 : pattern == 1 ? (lane & 15) * 32 + (lane >> 4) * 8 : (lane >> 4) * 128 + (lane & 15) * 8;
    addr += static_cast<unsigned>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) uint16_t *)lds));
    uint2 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr) : "memory");
    out[lane * 4 + 0] = v.x & 0xffff; out[lane * 4 + 1] = v.x >> 16; out[lane * 4 + 2] = v.y & 0xffff; out[lane * 4 + 3] = v.y >> 16;
}
extern "C" int tr_probe(uint16_t *out, int pattern, void *stream) {
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), out, pattern);
    return hipGetLastError() == hipSuccess ? 0 : -1;
}
