// Selective-scan forward for gfx950 (MI355X).  C ABI: zigma_selective_scan_fwd (include/zigma_hip.h).
//
// Replaces the reference's selective_scan_fwd_kernel (dis_mamba/csrc/selective_scan/
// selective_scan_fwd_kernel.cuh:67-303).  The reference gives one 64-thread block a whole (b, d)
// row, lays L across the threads and runs `dstate` CUB block scans of (a, b) pairs.  That shape is
// wrong for this machine: the recurrence is VALU / transcendental bound on CDNA4 (16 exp2 per
// element against 8 B of HBM traffic; measured v_exp_f32 ~ 1.8 v_fma_f32 issue slots), so every
// cross-lane combine of a parallel scan is pure overhead, and B/C would be re-read through L2 by every
// one of the 1280 channel rows of a sample.
//
// Two kernels:
//
//  scan_tok_kernel   — the hot path.  Token-major operands (channel contiguous), one LANE per
//      channel, time runs sequentially inside the lane, so there is NO scan and no cross-lane
//      combine: per (element, state) exactly v_mul, v_exp, v_mul, v_fma, v_fmac.  A workgroup owns a
//      64-channel slab of one sample; its NW waves split the dstate dimension (4 states each).  The
//      B_l / C_l values of a 4-step group sit in ONE VGPR per operand (lane -> (step, state), the same
//      16 values in every row of 16 lanes) and reach the FMAs as DPP row_newbcast operands: no SGPR
//      traffic, no LDS traffic, no extra instruction.  Per-element work (softplus, D*u, SiLU gate) is
//      done once per element by a cooperative prologue / epilogue around each LT-step tile and shared
//      through LDS.  The zigzag reordering is two row-index tables applied to whole 128-byte rows
//      (z gather, out_z scatter): coalesced by construction.
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

// token-major kernel: scan_tok.inc, instantiated per I/O element type in scan_tok_{bf16,f16,f32}.hip
int launch_scan_tok_bf16(const zigma_scan_params_t &p, hipStream_t stream);
int launch_scan_tok_bf16_dtp(const zigma_scan_params_t &p, hipStream_t stream);
int launch_scan_tok_f16(const zigma_scan_params_t &p, hipStream_t stream);
int launch_scan_tok_f16_dtp(const zigma_scan_params_t &p, hipStream_t stream);
int launch_scan_tok_f32(const zigma_scan_params_t &p, hipStream_t stream);

// =================================================================================================
// host dispatch
// =================================================================================================
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
    if (p.info) { p.info[0] = ZIGMA_SCAN_KERNEL_GENERIC; p.info[1] = 0; }
    return check_launch();
}

// token-major fast path: channels contiguous in u / delta / z / out, input-dependent B and C (any
// strides), dstate 16 (4 waves x 4 states) or 8 (2 waves), dim a multiple of the 64-channel slab.
static bool tok_eligible(const zigma_scan_params_t &p) {
    if (!p.is_variable_B || !p.is_variable_C || p.n_groups != 1) return false;
    if (p.dim % 64 != 0 || (p.dstate != 16 && p.dstate != 8)) return false;
    if (p.u_d_stride != 1 || p.delta_d_stride != 1) return false;
    if (p.z && (p.z_d_stride != 1 || p.out_z_d_stride != 1)) return false;
    if (p.out && p.out_d_stride != 1) return false;
    const int chunk_len = p.chunk_len > 0 ? p.chunk_len : 2048;
    if (p.x && chunk_len % 16 != 0) return false;  // carries are stored at tile ends
    if (p.bc_dtype != p.io_dtype) return false;    // instantiation set: B/C in the activation dtype
    // in-sample offsets are 32-bit in the kernel
    const int64_t lim = ((int64_t(1) << 31) - 1) / 4, Lm = p.seqlen;  // byte offsets, up to 4-byte elements
    const int64_t ls[] = {p.u_l_stride, p.delta_l_stride, p.z ? p.z_l_stride : 0, p.out ? p.out_l_stride : 0,
                          p.z ? p.out_z_l_stride : 0, p.B_l_stride, p.C_l_stride};
    for (int64_t s : ls)
        if (s < 0 || s * Lm > lim) return false;
    return true;
}

}  // namespace zigma

using namespace zigma;

extern "C" int zigma_selective_scan_fwd(const zigma_scan_params_t *pp, void *stream_) {
    if (!pp) return ZIGMA_ERR_NULL;
    (void)hipGetLastError();  // a stale error of an unrelated earlier call is not ours to report
    const zigma_scan_params_t &p = *pp;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    if (p.batch < 0 || p.dim < 0 || p.seqlen < 0 || p.dstate < 1 || p.dstate > 256) return ZIGMA_ERR_SHAPE;  // MAX_DSTATE
    if (p.n_groups < 1 || p.dim % p.n_groups != 0) return ZIGMA_ERR_SHAPE;
#ifdef ZIGMA_SCAN_PROBES
    constexpr int kProbeBits = 0x7000;          // timing probes of scan_tok2.inc (probe library of tools/ only)
#else
    constexpr int kProbeBits = 0;
#endif
    if (p.flags & ~(ZIGMA_SCAN_Z_PREACTIVATED | ZIGMA_SCAN_ACCUMULATE | ZIGMA_SCAN_PROBE_V1 | (1 << ZIGMA_SCAN_PROBE_PRIO_SHIFT) | (1 << ZIGMA_SCAN_PROBE_R5_SHIFT) | kProbeBits)) return ZIGMA_ERR_UNSUPPORTED;
    if (p.batch == 0 || p.dim == 0 || p.seqlen == 0) return ZIGMA_OK;  // empty (pointers may be NULL): nothing to launch
    if (p.dt_x) {       // ABI 9: dt_proj inside the token-major hot kernel; `delta` is not read (the layout checks below see u's strides)
        if (!p.u || !p.dt_w || !p.A || !p.B || !p.C || !p.z || !p.out_z) return ZIGMA_ERR_NULL;
        if (p.reset_period < 0 || p.reset_period % 16 != 0) return ZIGMA_ERR_SHAPE;
        zigma_scan_params_t q = p;
        if (p.x) {      // sequence split (ABI 10): `delta` is a WORKSPACE of u's shape the first pass fills with softplus(dt_proj + bias) for the second
            if (!p.delta || p.reset_period != 0) return p.delta ? ZIGMA_ERR_SHAPE : ZIGMA_ERR_NULL;
        } else {
            q.delta = p.u;
            q.delta_batch_stride = p.u_batch_stride; q.delta_d_stride = p.u_d_stride; q.delta_l_stride = p.u_l_stride;
        }
        if ((p.io_dtype != ZIGMA_BF16 && p.io_dtype != ZIGMA_F16) || !tok_eligible(q) || p.batch > 65535) return ZIGMA_ERR_UNSUPPORTED;
        return p.io_dtype == ZIGMA_BF16 ? launch_scan_tok_bf16_dtp(q, stream) : launch_scan_tok_f16_dtp(q, stream);
    }
    if (p.flags & ZIGMA_SCAN_ACCUMULATE) return ZIGMA_ERR_UNSUPPORTED;       // (only the in-kernel dt_proj form above adds to out_z)
    if (!p.u || !p.delta || !p.A || !p.B || !p.C) return ZIGMA_ERR_NULL;
    if (p.z && !p.out_z) return ZIGMA_ERR_NULL;
    if (!p.z && !p.out) return ZIGMA_ERR_NULL;

    if (p.reset_period < 0 || p.reset_period % 16 != 0 || (p.reset_period > 0 && p.x)) return ZIGMA_ERR_SHAPE;
    if (p.reset_period > 0 && !tok_eligible(p)) return ZIGMA_ERR_STRIDE;   // only the token-major kernel restarts sequences
    if ((p.flags & ZIGMA_SCAN_Z_PREACTIVATED) && !(tok_eligible(p) && p.io_dtype != ZIGMA_F32)) return ZIGMA_ERR_UNSUPPORTED;
    if (tok_eligible(p) && p.batch > 65535) {
        // the first-generation token-major kernel carries the batch in gridDim.y: larger batches (video temporal layers:
        // batch x tokens-per-frame rows) run in slices.  checkpoints are per (batch, slab): sliced alike.
        const size_t es = p.io_dtype == ZIGMA_F32 ? 4 : 2;
        const int chunk_len = p.chunk_len > 0 ? p.chunk_len : 2048;
        const int64_t n_chunks = (p.seqlen + chunk_len - 1) / chunk_len, n_tiles = (p.seqlen + 15) / 16;
        for (int b0 = 0; b0 < p.batch; b0 += 65535) {
            zigma_scan_params_t q = p;
            q.batch = p.batch - b0 < 65535 ? p.batch - b0 : 65535;
            auto adv = [&](const void *ptr, int64_t stride_elems, size_t esz) -> const void * {
                return ptr ? reinterpret_cast<const char *>(ptr) + static_cast<int64_t>(b0) * stride_elems * static_cast<int64_t>(esz) : nullptr;
            };
            q.u = adv(p.u, p.u_batch_stride, es);
            q.delta = adv(p.delta, p.delta_batch_stride, es);
            q.z = adv(p.z, p.z_batch_stride, es);
            q.out = const_cast<void *>(adv(p.out, p.out_batch_stride, es));
            q.out_z = const_cast<void *>(adv(p.out_z, p.out_z_batch_stride, es));
            q.B = adv(p.B, p.B_batch_stride, es);
            q.C = adv(p.C, p.C_batch_stride, es);
            q.x = const_cast<void *>(adv(p.x, static_cast<int64_t>(p.dim) * n_chunks * 2 * p.dstate, 4));
            q.checkpoints = reinterpret_cast<float *>(const_cast<void *>(
                adv(p.checkpoints, static_cast<int64_t>(p.dim / 64) * n_tiles * p.dstate * 64, 4)));
            const int rc = zigma_selective_scan_fwd(&q, stream_);
            if (rc != ZIGMA_OK) return rc;
        }
        return ZIGMA_OK;
    }
    if (tok_eligible(p)) {
        switch (p.io_dtype) {
            case ZIGMA_BF16: return launch_scan_tok_bf16(p, stream);
            case ZIGMA_F16: return launch_scan_tok_f16(p, stream);
            case ZIGMA_F32: return launch_scan_tok_f32(p, stream);
            default: return ZIGMA_ERR_DTYPE;
        }
    }
    ZIGMA_DISPATCH_DTYPE(p.io_dtype, IO, {
        ZIGMA_DISPATCH_DTYPE(p.bc_dtype, BCT, { return launch_generic<IO, BCT>(p, stream); })
    })
    return ZIGMA_ERR_DTYPE;
}
