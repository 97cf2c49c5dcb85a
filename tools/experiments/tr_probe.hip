// What does ds_read_b64_tr_b16 return?  Every lane passes its own 8-byte-aligned LDS address holding 4 distinct 16-bit
// values; LDS element at byte address a holds the value a/2 (its own element index).  Output: 4 values per lane.
#include <hip/hip_runtime.h>
#include <stdint.h>
__global__ void tr_probe(const int *lane_addr, uint16_t *out) {
    __shared__ __attribute__((aligned(16))) uint16_t lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = static_cast<uint16_t>(i);
    __syncthreads();
    const unsigned addr = static_cast<unsigned>(reinterpret_cast<uintptr_t>(lds)) + lane_addr[threadIdx.x];
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    u2 r;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr) : "memory");
    out[threadIdx.x * 4 + 0] = r[0] & 0xffff; out[threadIdx.x * 4 + 1] = r[0] >> 16;
    out[threadIdx.x * 4 + 2] = r[1] & 0xffff; out[threadIdx.x * 4 + 3] = r[1] >> 16;
}
extern "C" void run(const int *lane_addr, uint16_t *out, void *stream) {
    hipLaunchKernelGGL(tr_probe, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), lane_addr, out);
}
