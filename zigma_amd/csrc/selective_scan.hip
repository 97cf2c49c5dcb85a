// Selective-scan forward for gfx950 (MI355X).  C ABI: zigma_selective_scan_fwd (include/zigma_hip.h).
//
// Replaces the reference's selective_scan_fwd_kernel (dis_mamba/csrc/selective_scan/
// selective_scan_fwd_kernel.cuh:67-303).  The reference gives one 64-thread block a whole (b, d)
// row, lays L across the threads and runs `dstate` CUB block scans of (a, b) pairs.  That shape is
// wrong for this machine: the recurrence is VALU / transcendental bound on CDNA4 (16 exp2 per
// element against 8 B of HBM traffic), so every cross-lane combine is pure overhead, and B/C would
// be re-read through L2 by every one of the 1280 channel rows of a sample.
//
// Two kernels:
//
//  scan_tok_kernel   — the hot path.  Token-major operands (channel contiguous), one LANE per
//      channel, time runs sequentially inside the lane, so there is NO scan and no cross-lane
//      traffic: 2 packed FMA-pipe ops + 1 v_exp_f32 per (element, state).  A workgroup owns a
//      64-channel slab of one sample; its NW waves split the dstate dimension (SPW states each), so
//      the B_l / C_l values a wave needs are wave-uniform and travel through the SCALAR cache into
//      SGPRs (zero VGPR / LDS cost).  Per-element work (softplus, D*u, SiLU gate) is done once per
//      element by a cooperative prologue / epilogue around each LT-step tile and shared through LDS.
//      The zigzag reordering is two row-index tables applied to whole 128-byte rows (z gather,
//      out_z scatter): coalesced by construction.
//
//  scan_generic_kernel — any strides / constant or grouped B,C / any dstate <= 256: the reference's
//      full call surface (selective_scan.cpp:233-305).  One row per NS lanes (one lane per state),
//      butterfly reduction for y.  Compatibility path, not tuned.
#include "zigma_common.h"

namespace zigma {

// =================================================================================================
// generic kernel
// =================================================================================================
template <typename IO, typename BCT, int NS, int SPL>
__global__ __launch_bounds__(64) void scan_generic_kernel(const zigma_scan_params_t p) {
    constexpr int RPW = 64 / NS;  // rows per wave
    const int lane = threadIdx.x;
    const int sub = lane % NS;
    const int64_t nrows = static_cast<int64_t>(p.batch) * p.dim;
    int64_t row = static_cast<int64_t>(blockIdx.x) * RPW + lane / NS;
    const bool row_ok = row < nrows;
    if (!row_ok) row = nrows - 1;  // keep the lane in the shuffles
    const int b = static_cast<int>(row / p.dim);
    const int d = static_cast<int>(row % p.dim);
    const int g = d / (p.dim / p.n_groups);
    const int N = p.dstate;
    const int chunk_len = p.chunk_len > 0 ? p.chunk_len : 2048;
    const int n_chunks = (p.seqlen + chunk_len - 1) / chunk_len;

    float a2[SPL], h[SPL], bc_const_b[SPL], bc_const_c[SPL];
    bool n_ok[SPL];
#pragma unroll
    for (int j = 0; j < SPL; ++j) {
        const int n = sub + j * NS;
        n_ok[j] = n < N;
        const int nn = n_ok[j] ? n : 0;
        a2[j] = reinterpret_cast<const float *>(p.A)[d * p.A_d_stride + nn * p.A_dstate_stride] * kLog2e;
        h[j] = 0.f;
        bc_const_b[j] = p.is_variable_B ? 0.f
                        : reinterpret_cast<const float *>(p.B)[d * p.B_d_stride + nn * p.B_dstate_stride];
        bc_const_c[j] = p.is_variable_C ? 0.f
                        : reinterpret_cast<const float *>(p.C)[d * p.C_d_stride + nn * p.C_dstate_stride];
    }
    const float Dv = p.D ? reinterpret_cast<const float *>(p.D)[d] : 0.f;
    const float bias = p.delta_bias ? reinterpret_cast<const float *>(p.delta_bias)[d] : 0.f;
    const int64_t u_off = b * p.u_batch_stride + d * p.u_d_stride;
    const int64_t dl_off = b * p.delta_batch_stride + d * p.delta_d_stride;
    const int64_t z_off = b * p.z_batch_stride + d * p.z_d_stride;
    const int64_t o_off = b * p.out_batch_stride + d * p.out_d_stride;
    const int64_t oz_off = b * p.out_z_batch_stride + d * p.out_z_d_stride;
    const int64_t B_off = b * p.B_batch_stride + g * p.B_group_stride;
    const int64_t C_off = b * p.C_batch_stride + g * p.C_group_stride;
    float cum = 0.f;

    for (int l = 0; l < p.seqlen; ++l) {
        const float uv = ld<IO>(p.u, u_off + l * p.u_l_stride);
        float dv = ld<IO>(p.delta, dl_off + l * p.delta_l_stride) + bias;
        if (p.delta_softplus) dv = softplus20(dv);
        const float du = dv * uv;
        cum += dv;
        float acc = 0.f;
#pragma unroll
        for (int j = 0; j < SPL; ++j) {
            const int n = sub + j * NS;
            if (n_ok[j]) {
                const float Bv = p.is_variable_B ? ld<BCT>(p.B, B_off + n * p.B_dstate_stride + l * p.B_l_stride)
                                                 : bc_const_b[j];
                const float Cv = p.is_variable_C ? ld<BCT>(p.C, C_off + n * p.C_dstate_stride + l * p.C_l_stride)
                                                 : bc_const_c[j];
                h[j] = fast_exp2(dv * a2[j]) * h[j] + du * Bv;
                acc += h[j] * Cv;
            }
        }
#pragma unroll
        for (int s = NS / 2; s > 0; s >>= 1) acc += __shfl_xor(acc, s, 64);
        if (sub == 0 && row_ok) {
            const float y = acc + Dv * uv;
            const int64_t orow = p.out_row_index ? p.out_row_index[l] : l;
            if (p.out) st<IO>(p.out, o_off + orow * p.out_l_stride, y);
            if (p.z) {
                const int64_t zrow = p.z_row_index ? p.z_row_index[l] : l;
                const float zv = ld<IO>(p.z, z_off + zrow * p.z_l_stride);
                st<IO>(p.out_z, oz_off + orow * p.out_z_l_stride, y * silu(zv));
            }
        }
        if (p.x && row_ok && ((l + 1) % chunk_len == 0 || l == p.seqlen - 1)) {
            const int chunk = l / chunk_len;
            float *xr = reinterpret_cast<float *>(p.x) + (row * n_chunks + chunk) * 2 * N;
#pragma unroll
            for (int j = 0; j < SPL; ++j) {
                const int n = sub + j * NS;
                if (n_ok[j]) {
                    xr[2 * n] = fast_exp2(cum * a2[j]);
                    xr[2 * n + 1] = h[j];
                }
            }
        }
    }
}

// =================================================================================================
// token-major kernel
// =================================================================================================

// SPW wave-uniform B (or C) values of one time step -> SGPRs.  `base` and `off` are wave-uniform, so
// the loads select to s_load_dword{,x2,x4} and the bf16 widening to SALU shifts.
template <typename BCT, int SPW>
__device__ __forceinline__ void load_bc_uniform(const void *base, int64_t off, float (&v)[SPW]) {
    if constexpr (BCT::id == ZIGMA_F32) {
        const float *f = reinterpret_cast<const float *>(base) + off;
#pragma unroll
        for (int j = 0; j < SPW; ++j) v[j] = f[j];
    } else {
        static_assert(BCT::id == ZIGMA_BF16, "fast path carries B/C as f32 or bf16");
        const uint32_t *w = reinterpret_cast<const uint32_t *>(reinterpret_cast<const uint16_t *>(base) + off);
#pragma unroll
        for (int j = 0; j < SPW / 2; ++j) {
            const uint32_t x = w[j];
            v[2 * j] = __uint_as_float(x << 16);
            v[2 * j + 1] = __uint_as_float(x & 0xffff0000u);
        }
    }
}

template <typename IO, typename BCT, int SPW, int NW, int LT, bool HAS_Z, bool HAS_X>
__global__ __launch_bounds__(64 * NW) void scan_tok_kernel(const zigma_scan_params_t p) {
    static_assert(LT % NW == 0 && LT % 2 == 0, "tile rows split evenly over the waves");
    constexpr int RPT = LT / NW;  // tile rows handled by one wave in the cooperative phases
    // [buf][step pair][channel][sp0, du0, sp1, du1]  -> one ds_read_b128 feeds two steps
    __shared__ __attribute__((aligned(16))) float s_spdu[2][LT / 2][64][4];
    __shared__ __attribute__((aligned(16))) float s_y[NW][LT][64];  // per-wave partial y

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.y;
    const int c = blockIdx.x * 64 + lane;  // channel of this lane
    const int L = p.seqlen;
    const int n0 = wave * SPW;

    float a2[SPW], h[SPW];
#pragma unroll
    for (int j = 0; j < SPW; ++j) {
        a2[j] = reinterpret_cast<const float *>(p.A)[c * p.A_d_stride + (n0 + j) * p.A_dstate_stride] * kLog2e;
        h[j] = 0.f;
    }
    const float Dv = p.D ? reinterpret_cast<const float *>(p.D)[c] : 0.f;
    const float bias = p.delta_bias ? reinterpret_cast<const float *>(p.delta_bias)[c] : 0.f;
    const bool sp_on = p.delta_softplus != 0;

    const int64_t u_off = b * p.u_batch_stride + c;
    const int64_t dl_off = b * p.delta_batch_stride + c;
    const int64_t z_off = b * p.z_batch_stride + c;
    const int64_t o_off = b * p.out_batch_stride + c;
    const int64_t oz_off = b * p.out_z_batch_stride + c;
    const int64_t B_off = b * p.B_batch_stride + n0;
    const int64_t C_off = b * p.C_batch_stride + n0;
    const int chunk_len = p.chunk_len > 0 ? p.chunk_len : 2048;
    const int n_chunks = (L + chunk_len - 1) / chunk_len;
    float cum = 0.f;

    struct Rows { float u[RPT], d[RPT], z[RPT]; };  // this wave's rows of one tile, in registers
    Rows ra, rb;

    auto issue_loads = [&](int t, Rows &rw) {
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int k = t * LT + wave * RPT + i;
            const bool ok = k < L;
            const int kk = ok ? k : L - 1;
            rw.u[i] = ld<IO>(p.u, u_off + kk * p.u_l_stride);
            rw.d[i] = ld<IO>(p.delta, dl_off + kk * p.delta_l_stride);
            if constexpr (HAS_Z) {
                const int64_t zrow = p.z_row_index ? p.z_row_index[kk] : kk;
                rw.z[i] = ld<IO>(p.z, z_off + zrow * p.z_l_stride);
            }
            if (!ok) { rw.u[i] = 0.f; rw.d[i] = 0.f; }
        }
    };
    // cooperative prologue of tile t: softplus, delta*u -> LDS
    auto stage = [&](int t, const Rows &rw) {
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int r = wave * RPT + i;
            const int k = t * LT + r;
            float dv = rw.d[i] + bias;
            if (sp_on) dv = softplus20(dv);
            if (k >= L) dv = 0.f;  // identity step: a = 1, b = 0
            const float du = dv * rw.u[i];
            v2f w = {dv, du};
            *reinterpret_cast<v2f *>(&s_spdu[t & 1][r >> 1][lane][(r & 1) * 2]) = w;
        }
    };

    const int n_tiles = (L + LT - 1) / LT;

    // one tile: prefetch rows of t+1 -> recurrence over t -> stage t+1 -> barrier -> epilogue of t
    auto tile = [&](int t, Rows &cur, Rows &nxt) {
        const int s = t & 1;
        if (t + 1 < n_tiles) issue_loads(t + 1, nxt);

        // ---- recurrence over the tile: this wave's SPW states of 64 channels ------------------
        const int64_t kb = static_cast<int64_t>(t) * LT;
#pragma unroll
        for (int l2 = 0; l2 < LT / 2; ++l2) {
            const v4f q = *reinterpret_cast<const v4f *>(&s_spdu[s][l2][lane][0]);
#pragma unroll
            for (int e = 0; e < 2; ++e) {
                const int l = l2 * 2 + e;
                const float dv = e ? q.z : q.x;
                const float du = e ? q.w : q.y;
                int64_t k = kb + l;
                if (k >= L) k = L - 1;  // B/C of padded steps are multiplied by du = 0
                float Bv[SPW], Cv[SPW];
                load_bc_uniform<BCT, SPW>(p.B, B_off + k * p.B_l_stride, Bv);
                load_bc_uniform<BCT, SPW>(p.C, C_off + k * p.C_l_stride, Cv);
                float y = 0.f;
#pragma unroll
                for (int j = 0; j < SPW; ++j) {
                    const float a = fast_exp2(dv * a2[j]);
                    h[j] = a * h[j] + du * Bv[j];
                    y += h[j] * Cv[j];
                }
                s_y[wave][l][lane] = y;
                if constexpr (HAS_X) {
                    cum += dv;
                    const int64_t kk = kb + l;
                    if (kk < L && ((kk + 1) % chunk_len == 0 || kk == L - 1)) {
                        float *xr = reinterpret_cast<float *>(p.x) +
                                    ((static_cast<int64_t>(b) * p.dim + c) * n_chunks + kk / chunk_len) * 2 * p.dstate;
#pragma unroll
                        for (int j = 0; j < SPW; ++j) {
                            xr[2 * (n0 + j)] = fast_exp2(cum * a2[j]);
                            xr[2 * (n0 + j) + 1] = h[j];
                        }
                    }
                }
            }
        }
        if (t + 1 < n_tiles) stage(t + 1, nxt);
        __syncthreads();

        // ---- cooperative epilogue of tile t: sum the partial y, skip term, gate, store --------
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int r = wave * RPT + i;
            const int k = t * LT + r;
            if (k < L) {
                float y = Dv * cur.u[i];
#pragma unroll
                for (int w = 0; w < NW; ++w) y += s_y[w][r][lane];
                const int64_t orow = p.out_row_index ? p.out_row_index[k] : k;
                if (p.out) st<IO>(p.out, o_off + orow * p.out_l_stride, y);
                if constexpr (HAS_Z) st<IO>(p.out_z, oz_off + orow * p.out_z_l_stride, y * silu(cur.z[i]));
            }
        }
        __syncthreads();
    };

    issue_loads(0, ra);
    stage(0, ra);
    __syncthreads();
    for (int t = 0; t < n_tiles; t += 2) {
        tile(t, ra, rb);
        if (t + 1 < n_tiles) tile(t + 1, rb, ra);
    }
}

// =================================================================================================
// host dispatch
// =================================================================================================
static bool aligned(const void *ptr, size_t a) { return (reinterpret_cast<uintptr_t>(ptr) % a) == 0; }

template <typename IO, typename BCT, int SPW, int NW, int LT>
static int launch_tok(const zigma_scan_params_t &p, hipStream_t stream, const char *name) {
    dim3 grid(p.dim / 64, p.batch), block(64 * NW);
    const bool has_z = p.z != nullptr, has_x = p.x != nullptr;
    if (has_z && has_x) hipLaunchKernelGGL((scan_tok_kernel<IO, BCT, SPW, NW, LT, true, true>), grid, block, 0, stream, p);
    else if (has_z) hipLaunchKernelGGL((scan_tok_kernel<IO, BCT, SPW, NW, LT, true, false>), grid, block, 0, stream, p);
    else if (has_x) hipLaunchKernelGGL((scan_tok_kernel<IO, BCT, SPW, NW, LT, false, true>), grid, block, 0, stream, p);
    else hipLaunchKernelGGL((scan_tok_kernel<IO, BCT, SPW, NW, LT, false, false>), grid, block, 0, stream, p);
    set_last_kernel(name);
    return check_launch();
}

template <typename IO, typename BCT>
static int launch_generic(const zigma_scan_params_t &p, hipStream_t stream) {
    const int N = p.dstate;
    const int64_t nrows = static_cast<int64_t>(p.batch) * p.dim;
#define ZIGMA_GEN(NS_, SPL_)                                                                           \
    {                                                                                                  \
        const int rpw = 64 / NS_;                                                                      \
        dim3 grid(static_cast<unsigned>((nrows + rpw - 1) / rpw)), block(64);                          \
        hipLaunchKernelGGL((scan_generic_kernel<IO, BCT, NS_, SPL_>), grid, block, 0, stream, p);      \
    }
    if (N <= 1) ZIGMA_GEN(1, 1)
    else if (N <= 2) ZIGMA_GEN(2, 1)
    else if (N <= 4) ZIGMA_GEN(4, 1)
    else if (N <= 8) ZIGMA_GEN(8, 1)
    else if (N <= 16) ZIGMA_GEN(16, 1)
    else if (N <= 32) ZIGMA_GEN(32, 1)
    else if (N <= 64) ZIGMA_GEN(64, 1)
    else if (N <= 128) ZIGMA_GEN(64, 2)
    else ZIGMA_GEN(64, 4)
#undef ZIGMA_GEN
    set_last_kernel("scan_generic");
    return check_launch();
}

// token-major fast path applies when channels are contiguous everywhere and B/C rows hold the
// dstate values of one token contiguously (x_dbl rows of the fused block).
static bool tok_eligible(const zigma_scan_params_t &p) {
    if (!p.is_variable_B || !p.is_variable_C || p.n_groups != 1) return false;
    if (p.dim % 64 != 0 || p.dstate != 16) return false;
    if (p.u_d_stride != 1 || p.delta_d_stride != 1) return false;
    if (p.z && (p.z_d_stride != 1 || p.out_z_d_stride != 1)) return false;
    if (p.out && p.out_d_stride != 1) return false;
    if (!p.z && !p.out) return false;
    if (p.B_dstate_stride != 1 || p.C_dstate_stride != 1) return false;
    if (p.A_dstate_stride != 1) return false;
    if (p.bc_dtype == ZIGMA_F16) return false;
    const size_t es = p.bc_dtype == ZIGMA_F32 ? 4 : 2;
    // every SPW-group of B/C must start on a dword (scalar loads): 4 states * es bytes per group
    if (!aligned(p.B, 4) || !aligned(p.C, 4)) return false;
    if ((p.B_l_stride * es) % 4 || (p.C_l_stride * es) % 4 || (p.B_batch_stride * es) % 4 || (p.C_batch_stride * es) % 4)
        return false;
    return true;
}

}  // namespace zigma

using namespace zigma;

extern "C" int zigma_selective_scan_fwd(const zigma_scan_params_t *pp, void *stream_) {
    if (!pp) return ZIGMA_ERR_NULL;
    const zigma_scan_params_t &p = *pp;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (!p.u || !p.delta || !p.A || !p.B || !p.C) return ZIGMA_ERR_NULL;
    if (p.z && !p.out_z) return ZIGMA_ERR_NULL;
    if (!p.z && !p.out) return ZIGMA_ERR_NULL;
    if (p.batch < 0 || p.dim < 0 || p.seqlen < 0 || p.dstate < 1 || p.dstate > 256) return ZIGMA_ERR_SHAPE;  // MAX_DSTATE
    if (p.n_groups < 1 || p.dim % p.n_groups != 0) return ZIGMA_ERR_SHAPE;
    if (p.flags != 0) return ZIGMA_ERR_UNSUPPORTED;
    if (p.batch == 0 || p.dim == 0 || p.seqlen == 0) return ZIGMA_OK;  // empty: nothing to launch

    if (tok_eligible(p)) {
        // states-per-wave split: enough workgroups to fill 256 CUs x 4 SIMDs, else keep whole rows per wave
        const int64_t slabs = static_cast<int64_t>(p.dim / 64) * p.batch;
        const int forced = 0;
        (void)forced;
        if (p.bc_dtype == ZIGMA_BF16) {
            ZIGMA_DISPATCH_DTYPE(p.io_dtype, IO, {
                if (slabs >= 4096) return launch_tok<IO, BF16, 16, 1, 16>(p, stream, "scan_tok_s16w1");
                if (slabs >= 2048) return launch_tok<IO, BF16, 8, 2, 16>(p, stream, "scan_tok_s8w2");
                return launch_tok<IO, BF16, 4, 4, 16>(p, stream, "scan_tok_s4w4");
            })
        } else {
            ZIGMA_DISPATCH_DTYPE(p.io_dtype, IO, {
                if (slabs >= 2048) return launch_tok<IO, F32, 8, 2, 8>(p, stream, "scan_tok_s8w2");
                return launch_tok<IO, F32, 4, 4, 8>(p, stream, "scan_tok_s4w4");
            })
        }
    }
    ZIGMA_DISPATCH_DTYPE(p.io_dtype, IO, {
        ZIGMA_DISPATCH_DTYPE(p.bc_dtype, BCT, { return launch_generic<IO, BCT>(p, stream); })
    })
    return ZIGMA_ERR_DTYPE;
}
