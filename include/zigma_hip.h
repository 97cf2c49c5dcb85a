/*
 * zigma_hip.h — C ABI of libzigma_hip.so, the MI355X (gfx950) native kernels behind the
 * ZigMa denoiser-forward / ODE-sampling hot path.
 *
 * Drop-in boundary.  Each entry point replaces one native entry of the
 * reference (CompVis/zigma); the parameter blocks mirror the PODs the reference already hands to
 * its CUDA kernels, widened so that token-major (channel-contiguous) layouts and the zigzag row
 * tables can be expressed without a separate index_select pass.
 *
 * Conventions (all entry points):
 *   - plain pointers and sizes only; every pointer is a DEVICE pointer on the current device;
 *   - the caller owns all memory and allocates the outputs; the library never allocates,
 *     never synchronises, keeps no state a caller depends on and is re-entrant (which kernel variant served a
 *     call is reported through the parameter block's optional `info` out-field; zigma_last_kernel() is a
 *     thread-local DIAGNOSTIC for tests and logs only);
 *   - `stream` is a hipStream_t passed as void*; one call = one or more kernel launches on it;
 *   - strides are in ELEMENTS of the tensor's own dtype;
 *   - returns ZIGMA_OK (0) or a negative zigma_status_t; zigma_strerror() names it.  The Python
 *     shim turns a non-zero status into RuntimeError, as TORCH_CHECK does in the reference.
 */
#ifndef ZIGMA_HIP_H_
#define ZIGMA_HIP_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ZIGMA_ABI_VERSION 10  /* 2: parameter blocks grew (row tables in the backward, checkpoints, reset_period)
                               * 3: scan block: `info` out-field, ZIGMA_SCAN_Z_PREACTIVATED flag; zigma_linear_fwd
                               * 4: zigma_linear_params_t grew (gated residual epilogue); zigma_conv_x_proj_fwd, zigma_q_attn_fwd
                               * 5: pruned — zigma_q_attn_fwd and the dt product of zigma_conv_xproj_params_t removed (measured no faster,
                               *    archived under tools/experiments/)
                               * 6: zigma_cross_attn_bwd / zigma_cross_attn_bwd_chunks added
                               * 7: reset_period in the two backward blocks (zigma_scan_bwd_params_t reuses its padding, zigma_conv_bwd_params_t grew)
                               * 8: zigma_patch_embed_fwd, zigma_timestep_embed_fwd, zigma_final_layer_fwd, zigma_skinny_linear_fwd added
                               * 9: zigma_scan_params_t grew: dt_x / dt_w (dt_proj + softplus inside the scan kernel)
                               * 10: zigma_calib_launch (bench.py's box calibration) added; new ZIGMA_LINEAR_* kernel selectors of round 6 */

/* zigma_scan_params_t.flags */
#define ZIGMA_SCAN_Z_PREACTIVATED 2   /* z already holds silu(z) (the in_proj GEMM epilogue applied it): out_z = y * z */
#define ZIGMA_SCAN_ACCUMULATE 4        /* out_z += y * silu(z) instead of out_z = ...: the second sweep of the bidirectional `v2` scan type
                                       * (mamba_simple.py:335-339) adds itself to the first one's result; served with dt_x / dt_w only (ABI 10) */
#define ZIGMA_SCAN_PROBE_V1 0x100     /* A/B probe (tools/scan_ab.py): pin the first-generation token-major kernel */
#define ZIGMA_SCAN_PROBE_PRIO_SHIFT 9 /* A/B probe: bit 9 = scan_tok2_kernel WITHOUT its wave-priority rotation */
#define ZIGMA_SCAN_PROBE_R5_SHIFT 10  /* A/B probe: bit 10 = never the six-resident-workgroups form of scan_tok2_kernel */

/* zigma_scan_params_t.info[0]: which kernel family served the call */
#define ZIGMA_SCAN_KERNEL_GENERIC 1
#define ZIGMA_SCAN_KERNEL_TOK 2    /* scan_tok_kernel: token-major, all features (carries, checkpoints, reset_period) */
#define ZIGMA_SCAN_KERNEL_TOK2 3   /* scan_tok2_kernel: token-major hot kernel (16-bit I/O, dstate 16, gate only)      */

typedef enum zigma_status {
    ZIGMA_OK = 0,
    ZIGMA_ERR_NULL = -1,        /* required pointer is NULL                                  */
    ZIGMA_ERR_SHAPE = -2,       /* size out of the supported range                           */
    ZIGMA_ERR_DTYPE = -3,       /* unsupported element type                                  */
    ZIGMA_ERR_STRIDE = -4,      /* layout not supported by any kernel                        */
    ZIGMA_ERR_LAUNCH = -5,      /* hipLaunchKernel / hipGetLastError failed                  */
    ZIGMA_ERR_UNSUPPORTED = -6  /* feature of the reference that is out of scope (complex A) */
} zigma_status_t;

typedef enum zigma_dtype { ZIGMA_F32 = 0, ZIGMA_F16 = 1, ZIGMA_BF16 = 2 } zigma_dtype_t;

/* ------------------------------------------------------------------------------------------
 * Selective scan forward.
 * Replaces  selective_scan_cuda.fwd  (reference dis_mamba/csrc/selective_scan/selective_scan.cpp:226-336,
 * kernel selective_scan_fwd_kernel.cuh:67-303); parameter block mirrors SSMParamsBase
 * (selective_scan.h:26-69).  Real A only (the reference's complex variant is never reached by ZigMa,
 * mamba_simple.py:298).
 *
 *   delta' = delta + delta_bias[d];  if delta_softplus: delta' = delta' <= 20 ? log1p(exp(delta')) : delta'
 *   h_n[l] = exp(delta'[l] * A[d,n]) * h_n[l-1] + delta'[l] * B[n,l] * u[l]
 *   out[l] = sum_n C[n,l] * h_n[l] + D[d] * u[l];      out_z[l] = out[l] * z[l] / (1 + exp(-z[l]))
 *
 * u, delta, z, out, out_z are logically (batch, dim, seqlen) with arbitrary batch/d/l strides:
 *   reference layout  = l_stride 1 (the reference REQUIRES this, selective_scan.cpp:252-253);
 *   token-major layout = d_stride 1, l_stride = row pitch (what the fused ZigMa block uses).
 * Variable B/C are (batch, n_groups, dstate, seqlen); constant B/C are (dim, dstate) float32.
 * x receives the running prefix at every `chunk_len` boundary, (batch, dim, n_chunks, 2*dstate) f32,
 * contiguous: x[b,d,c,2n] = prod_{l<=end(c)} exp(delta' A), x[b,d,c,2n+1] = h_n[end(c)]
 * (selective_scan_fwd_kernel.cuh:251-254); last_state = x[:, :, -1, 1::2].  May be NULL.
 *
 * z_row_index / out_row_index (optional, int32[seqlen]) fuse the zigzag reordering of
 * mamba_simple.py:55-61,362-395 into the kernel's own loads/stores: scan position k reads its gate
 * from row z_row_index[k] of z and writes out/out_z to row out_row_index[k].
 * ------------------------------------------------------------------------------------------ */
typedef struct zigma_scan_params {
    int32_t batch, dim, seqlen, dstate, n_groups;
    int32_t is_variable_B, is_variable_C, delta_softplus;
    int32_t io_dtype;   /* zigma_dtype_t of u, delta, z, out, out_z                                */
    int32_t bc_dtype;   /* zigma_dtype_t of VARIABLE B / C (reference: == io_dtype)               */
    int32_t chunk_len;  /* carry spacing for x; 0 -> 2048 (reference: selective_scan.cpp:307)      */
    int32_t flags;      /* 0 or ZIGMA_SCAN_Z_PREACTIVATED (+ ZIGMA_SCAN_PROBE_* bits; others must be 0)      */

    int64_t u_batch_stride, u_d_stride, u_l_stride;
    int64_t delta_batch_stride, delta_d_stride, delta_l_stride;
    int64_t z_batch_stride, z_d_stride, z_l_stride;
    int64_t out_batch_stride, out_d_stride, out_l_stride;
    int64_t out_z_batch_stride, out_z_d_stride, out_z_l_stride;
    int64_t A_d_stride, A_dstate_stride;
    int64_t B_batch_stride, B_group_stride, B_d_stride, B_dstate_stride, B_l_stride;
    int64_t C_batch_stride, C_group_stride, C_d_stride, C_dstate_stride, C_l_stride;

    const void *u, *delta, *A, *B, *C;
    const void *D;           /* float32 (dim) or NULL  */
    const void *delta_bias;  /* float32 (dim) or NULL  */
    const void *z;           /* or NULL: no gating      */
    void *out;               /* ungated y; may be NULL when z != NULL (only backward needs it)     */
    void *out_z;             /* required iff z != NULL  */
    void *x;                 /* float32 or NULL         */
    const int32_t *z_row_index;
    const int32_t *out_row_index;
    /* optional float32 [batch][dim/64][ceil(seqlen/16)][dstate][64]: the state h BEFORE every 16-step tile, written by the
     * token-major kernel when both out and out_z are requested (the training forward); zigma_selective_scan_bwd takes it
     * back as `checkpoints` and skips its own forward phase.  Ignored (left untouched) by every other kernel variant:
     * pass `info` and check info[1] == 1 before trusting it. */
    float *checkpoints;
    /* > 0: the sequence is a concatenation of independent sequences of this many steps (a multiple of 16): the state is
     * reset to 0 at every multiple.  Lets the video temporal layers (b (t k) c tokens, scan over t for every (b, k)) run as
     * batch = k, seqlen = b * t on strided views with no transposing copy.  Token-major kernel only; x must be NULL. */
    int32_t reset_period;
    int32_t pad2_;
    /* optional HOST pointer to int32[2], written by the call before it returns (never by a kernel):
     * info[0] = ZIGMA_SCAN_KERNEL_* that was launched, info[1] = 1 iff `checkpoints` is being written. */
    int32_t *info;
    /* ABI 9 — dt_proj INSIDE the scan (token-major hot kernel, bf16 / fp16, whole-sequence mode only): when dt_x != NULL, `delta` is ignored and
     *   delta'[b, l, d] = softplus( sum_{r < dt_rank} dt_x[b, l, r] * dt_w[d, r] + delta_bias[d] )
     * is formed by the workgroup itself (v_mfma_f32_16x16x32_bf16 in the tile prologue) from the dt columns of the x_dbl rows it
     * already fetches B_l / C_l from — the (batch, seqlen, dim) delta tensor (reference selective_scan_interface.py:323) is
     * neither written nor read.  dt_x: rows of scan position l (row pitch dt_x_l_stride, 16-byte aligned), dt_w: (dim, dt_rank)
     * rows of pitch dt_w_row_stride (16-byte aligned); 32 <= dt_rank <= 64, dt_rank % 8 == 0, dt_x rows at least 64 wide.
     * delta_softplus must be 1.  ZIGMA_SCAN_Z_PREACTIVATED may be combined with it (round 5). */
    const void *dt_x, *dt_w;
    int64_t dt_x_batch_stride, dt_x_l_stride, dt_w_row_stride;
    int32_t dt_rank, pad3_;
} zigma_scan_params_t;

int zigma_selective_scan_fwd(const zigma_scan_params_t *p, void *stream);

/* ------------------------------------------------------------------------------------------
 * Depthwise causal conv1d (+ bias, + SiLU) forward.
 * Replaces  causal_conv1d_cuda.causal_conv1d_fwd  (reference dis_causal_conv1d/csrc/causal_conv1d.cpp:130-189,
 * kernels causal_conv1d_fwd.cu:39-130,193-298); parameter block mirrors ConvParamsBase
 * (causal_conv1d.h:9-35).
 *
 *   out[b,c,l] = act( bias[c] + sum_{w<width} weight[c,w] * x[b,c,l-(width-1-w)] ),  x[<0] = 0
 *
 * x/out logically (batch, dim, seqlen), arbitrary strides (channel-first, channel-last, views).
 * x_row_index (optional, int32[seqlen]): the conv runs over the REORDERED sequence
 * x'[k] = x[x_row_index[k]]  (zigzag gather of mamba_simple.py:362-370 fused into the loads).
 * ------------------------------------------------------------------------------------------ */
typedef struct zigma_conv_params {
    int32_t batch, dim, seqlen, width;
    int32_t silu_activation;
    int32_t io_dtype;  /* x, out */
    int32_t w_dtype;   /* weight, bias */
    int32_t flags;     /* reserved, must be 0 */
    int64_t x_batch_stride, x_c_stride, x_l_stride;
    int64_t weight_c_stride, weight_width_stride;
    int64_t out_batch_stride, out_c_stride, out_l_stride;
    const void *x, *weight;
    const void *bias;  /* or NULL */
    void *out;
    const int32_t *x_row_index;
    /* > 0: independent sequences of this many positions (a multiple of 16) concatenated along seqlen: the causal window
     * does not reach across a multiple (zero padding restarts there).  Token-major kernel only. */
    int32_t reset_period;
    int32_t pad2_;
} zigma_conv_params_t;

int zigma_causal_conv1d_fwd(const zigma_conv_params_t *p, void *stream);

/* ------------------------------------------------------------------------------------------
 * Fused (gated-branch add) + residual add + RMSNorm / LayerNorm (+ adaLN modulate) forward.
 * Replaces the Triton kernel `_layer_norm_fwd_1pass_kernel` and its host `_layer_norm_fwd`
 * (reference dis_mamba/mamba_ssm/ops/triton/layernorm.py:65-120,123-177) and, when the optional
 * pointers are given, the elementwise glue of Block.forward around it (model_zigma.py:53-54,388-460).
 *
 * Per row r (batch index b = r / rows_per_batch):
 *   xe  = x[r]  (+ gate[b] * branch[r]          if branch != NULL;  xe is stored to x_out if != NULL)
 *   res = xe (+ residual[r]);  residual_out[r] = res            (float statistics, layernorm.py:98-105)
 *   y   = is_rms ? res * rsqrt(mean(res^2) + eps) : (res - mean) * rsqrt(var + eps)
 *   y   = y * weight (+ bias)                                   (weight/bias may be NULL)
 *   y_out[r] = y                                                (if y_out != NULL)
 *   y_mod[r] = y * (1 + scale[b]) + shift[b]                    (if y_mod != NULL)   modulate()
 * x, branch, x_out, y_out, y_mod share x_dtype; residual/residual_out have res_dtype; weight/bias
 * w_dtype; gate/shift/scale are (batch, cols) rows of mod_dtype with pitch mod_batch_stride.
 * ------------------------------------------------------------------------------------------ */
typedef struct zigma_norm_params {
    int32_t rows, cols, rows_per_batch;
    int32_t is_rms;
    int32_t x_dtype, res_dtype, w_dtype, mod_dtype;
    float eps;
    int32_t flags;  /* reserved, must be 0 */
    int64_t x_row_stride, branch_row_stride, x_out_row_stride;
    int64_t res_row_stride, res_out_row_stride;
    int64_t y_row_stride, y_mod_row_stride;
    int64_t mod_batch_stride;
    const void *x;
    const void *branch, *gate;  /* both or neither */
    void *x_out;
    const void *residual;       /* or NULL */
    void *residual_out;         /* or NULL */
    const void *weight, *bias;  /* or NULL */
    void *y_out;                /* or NULL */
    const void *shift, *scale;  /* both or neither */
    void *y_mod;                /* required iff shift != NULL */
} zigma_norm_params_t;

int zigma_add_norm_fwd(const zigma_norm_params_t *p, void *stream);

/* ------------------------------------------------------------------------------------------
 * dt_proj + bias + softplus on the matrix cores (bf16 in / fp32 accumulate / bf16 out).
 * Replaces the skinny GEMM  delta = delta_proj_weight @ x_dbl[:, :dt_rank].T  of the reference
 * (dis_mamba/mamba_ssm/ops/selective_scan_interface.py:323, K = dt_rank) together with the
 * softplus(delta + delta_bias) its scan kernel applies first (selective_scan_fwd_kernel.cuh:153-156):
 *
 *   out[m, d] = act( sum_{r<k} x[m, r] * w[d, r] + bias[d] ),   act = softplus with pass-through above 20
 *
 * x: (m, >=k) rows of pitch x_row_stride (the first k columns of x_dbl);  w: (n, k) = dt_proj.weight;
 * bias: float32 (n) or NULL;  out: (m, n).  The selective scan is then called with delta_softplus = 0 and
 * delta_bias = NULL.  Limits: dtype bf16, k <= 48 and a multiple of 8, n % 64 == 0, x/w rows 16-byte aligned.
 * ------------------------------------------------------------------------------------------ */
typedef struct zigma_dtproj_params {
    int64_t m;              /* tokens (batch * seqlen) */
    int32_t n, k;           /* d_inner, dt_rank */
    int32_t dtype;          /* zigma_dtype_t of x, w, out */
    int32_t softplus;       /* 0: plain affine */
    int32_t flags;          /* reserved, must be 0 */
    int32_t pad_;
    int64_t x_row_stride, w_row_stride, out_row_stride;
    const void *x, *w;
    const void *bias;
    void *out;
} zigma_dtproj_params_t;

int zigma_dt_proj_softplus_fwd(const zigma_dtproj_params_t *p, void *stream);

/* ------------------------------------------------------------------------------------------
 * Selective scan backward (token-major operands, real A, variable B/C, one group).
 * Replaces  selective_scan_cuda.bwd  (reference dis_mamba/csrc/selective_scan/selective_scan.cpp:338-492, kernel
 * selective_scan_bwd_kernel.cuh:59-329, reverse_scan.cuh) for the layout the fused block uses.  Given dout = dL/d(out_z)
 * (or dL/d(out) when z == NULL) it produces (selective_scan_bwd_kernel.cuh:161-329):
 *   g = dout * silu(z);  dz = dout * out * sigmoid(z) * (1 + z (1 - sigmoid(z)))          (out = the forward's UNGATED y)
 *   dh_l = g_l C_l + a_{l+1} dh_{l+1};   dC[b,n,l] = sum_d g_l h_l;   dB[b,n,l] = sum_d dh_l delta'_l u_l
 *   du = g D + delta' sum_n dh B;   ddelta = (sum_n dh (B u + A a h_{l-1})) * softplus'(delta + bias)
 *   dA[d,n] = sum_{b,l} dh delta' a h_{l-1};   dD[d] = sum g u;   ddelta_bias[d] = sum ddelta
 * u, delta, z, out, dout, du, ddelta, dz: (batch, seqlen, dim), channel stride 1, io_dtype; B, C: (batch, dstate, seqlen)
 * logical with arbitrary strides, io_dtype; dB, dC: float32, same logical shape, arbitrary strides (e.g. columns of a
 * float32 d(x_dbl) buffer); dA (dim, dstate), dD (dim), ddelta_bias (dim): float32 contiguous, WRITTEN (not accumulated).
 * The reverse sweep re-creates the states from checkpoints every 16 steps, which the kernel writes itself in a first
 * forward phase: the caller provides `workspace` of zigma_selective_scan_bwd_workspace_bytes() bytes (the library
 * never allocates).  All sums are formed in a fixed order: results are bit-reproducible run to run.
 * Limits: dim % 64 == 0, dstate 16 or 8.
 * ------------------------------------------------------------------------------------------ */
typedef struct zigma_scan_bwd_params {
    int32_t batch, dim, seqlen, dstate;
    int32_t delta_softplus;
    int32_t io_dtype;
    int32_t flags;           /* reserved, must be 0 */
    int32_t reset_period;    /* > 0 (multiple of 16): every batch row is a concatenation of independent sequences of that many steps,
                              * as in zigma_scan_params_t: the state restarts there and no gradient crosses the boundary */
    int64_t u_batch_stride, u_l_stride;
    int64_t delta_batch_stride, delta_l_stride;
    int64_t z_batch_stride, z_l_stride;
    int64_t out_batch_stride, out_l_stride;
    int64_t dout_batch_stride, dout_l_stride;
    int64_t du_batch_stride, du_l_stride;
    int64_t ddelta_batch_stride, ddelta_l_stride;
    int64_t dz_batch_stride, dz_l_stride;
    int64_t A_d_stride, A_dstate_stride;
    int64_t B_batch_stride, B_dstate_stride, B_l_stride;
    int64_t C_batch_stride, C_dstate_stride, C_l_stride;
    int64_t dB_batch_stride, dB_dstate_stride, dB_l_stride;
    int64_t dC_batch_stride, dC_dstate_stride, dC_l_stride;
    const void *u, *delta, *A, *B, *C;
    const void *D;           /* float32 (dim) or NULL */
    const void *delta_bias;  /* float32 (dim) or NULL */
    const void *z;           /* or NULL */
    const void *out;         /* ungated forward output; required iff z != NULL */
    const void *dout;
    void *du, *ddelta;
    void *dz;                /* required iff z != NULL */
    float *dA, *dB, *dC;
    float *dD;               /* required iff D != NULL */
    float *ddelta_bias;      /* required iff delta_bias != NULL */
    void *workspace;
    int64_t workspace_bytes;
    /* optional int32[seqlen] row tables, the forward's own (zigma_scan_params_t): scan position k reads z from / writes dz
     * to row z_row_index[k], and reads out / dout from row out_row_index[k].  NULL = row k. */
    const int32_t *z_row_index;
    const int32_t *out_row_index;
    /* optional: the checkpoints the forward wrote (zigma_scan_params_t.checkpoints); NULL = recompute them here */
    const float *checkpoints;
} zigma_scan_bwd_params_t;

int64_t zigma_selective_scan_bwd_workspace_bytes(const zigma_scan_bwd_params_t *p);
int zigma_selective_scan_bwd(const zigma_scan_bwd_params_t *p, void *stream);

/* ------------------------------------------------------------------------------------------
 * Depthwise causal conv1d (+ bias, + SiLU) backward, token-major operands.
 * Replaces  causal_conv1d_cuda.causal_conv1d_bwd  (reference dis_causal_conv1d/csrc/causal_conv1d.cpp:191-283,
 * kernels causal_conv1d_bwd.cu:46-240,301-470).  x, dout, dx: (batch, seqlen, dim), channel stride 1, io_dtype;
 * dout is in SCAN order (the order the forward wrote `out` in); x is read through x_row_index exactly as in the
 * forward and dx is scattered through the same table (dx[row[k]] = dx'[k]; the table must be a permutation).
 * dweight (dim, width), dbias (dim): float32 contiguous, WRITTEN.  Sums are formed in a fixed order.
 * workspace: zigma_causal_conv1d_bwd_workspace_bytes() bytes.
 * ------------------------------------------------------------------------------------------ */
typedef struct zigma_conv_bwd_params {
    int32_t batch, dim, seqlen, width;
    int32_t silu_activation;
    int32_t io_dtype;  /* x, dout, dx */
    int32_t w_dtype;   /* weight, bias */
    int32_t flags;     /* reserved, must be 0 */
    int64_t x_batch_stride, x_l_stride;
    int64_t dout_batch_stride, dout_l_stride;
    int64_t dx_batch_stride, dx_l_stride;
    int64_t weight_c_stride, weight_width_stride;
    const void *x, *weight;
    const void *bias;  /* or NULL */
    const void *dout;
    void *dx;
    float *dweight;
    float *dbias;      /* required iff bias != NULL */
    const int32_t *x_row_index;
    void *workspace;
    int64_t workspace_bytes;
    int32_t reset_period;   /* > 0 (multiple of 16): independent sequences of that many positions along seqlen (zigma_conv_params_t) */
    int32_t pad_;
} zigma_conv_bwd_params_t;

int64_t zigma_causal_conv1d_bwd_workspace_bytes(const zigma_conv_bwd_params_t *p);
int zigma_causal_conv1d_bwd(const zigma_conv_bwd_params_t *p, void *stream);

/* ------------------------------------------------------------------------------------------
 * RMSNorm / LayerNorm (+ residual add, prenorm) backward.
 * Replaces `_layer_norm_bwd` + `_layer_norm_bwd_kernel` (reference dis_mamba/mamba_ssm/ops/triton/layernorm.py:196-377).
 * xsum = the tensor that was normalised in the forward (x, or the forward's residual_out = x + residual), res_dtype;
 * mean / rstd are recomputed from it.  dy: gradient of the normalised output (x_dtype); dresidual_out (optional,
 * res_dtype): gradient arriving at the prenorm residual output.  Writes dx (x_dtype) and / or dresidual (res_dtype)
 * — both carry  (wdy - xhat*mean(xhat*wdy) - mean(wdy)) * rstd + dresidual_out  — and dweight / dbias (float32 (cols),
 * WRITTEN, fixed summation order).  workspace: zigma_add_norm_bwd_workspace_bytes() bytes.
 * ------------------------------------------------------------------------------------------ */
typedef struct zigma_norm_bwd_params {
    int32_t rows, cols;
    int32_t is_rms;
    int32_t x_dtype, res_dtype, w_dtype;
    float eps;
    int32_t flags;           /* reserved, must be 0 */
    int64_t xsum_row_stride, dy_row_stride, dres_out_row_stride, dx_row_stride, dres_row_stride;
    const void *xsum;
    const void *weight;      /* or NULL */
    const void *dy;
    const void *dresidual_out;  /* or NULL */
    void *dx;                /* or NULL */
    void *dresidual;         /* or NULL (at least one of dx / dresidual) */
    float *dweight;          /* or NULL */
    float *dbias;            /* or NULL */
    void *workspace;
    int64_t workspace_bytes;
} zigma_norm_bwd_params_t;

int64_t zigma_add_norm_bwd_workspace_bytes(const zigma_norm_bwd_params_t *p);
int zigma_add_norm_bwd(const zigma_norm_bwd_params_t *p, void *stream);

/* ------------------------------------------------------------------------------------------
 * Backward of the block's elementwise glue (modulate, gated branch add; reference model_zigma.py:53-54,441-458) in one pass:
 *   out[r, :]   = dy[r, :] * (s[r / rows_per_batch, :] + s_add)                       (out may be NULL)
 *   r1[b, c, :] = sum over the 64 rows of chunk c of sample b of dy[r, :] * a[r, :]
 *   r2[b, c, :] = the same sum of dy[r, :]                                              (r2 may be NULL)
 * modulate backward: a = x, s = scale, s_add = 1 (out = dx, sum_c r1 = dscale, sum_c r2 = dshift); gated add backward: a = branch,
 * s = gate, s_add = 0 (out = dbranch, sum_c r1 = dgate).  dy, a, out: (rows, cols) bf16 rows; s: (batch, cols) bf16 rows;
 * r1, r2: float32 [batch][rows_per_batch / 64][cols], summed over the chunks by the caller (fixed order, no float atomics).
 * Limits: bf16, cols % 128 == 0, rows_per_batch % 64 == 0, 16-byte aligned rows.
 * ------------------------------------------------------------------------------------------ */
typedef struct zigma_glue_bwd_params {
    int64_t rows;
    int32_t cols, rows_per_batch;
    int32_t dtype;           /* ZIGMA_BF16 */
    int32_t flags;           /* reserved, must be 0 */
    float s_add;
    int32_t pad_;
    int64_t dy_row_stride, a_row_stride, out_row_stride, s_batch_stride;
    const void *dy, *a, *s;
    void *out;
    void *r1, *r2;
} zigma_glue_bwd_params_t;

int zigma_scale_reduce_bwd(const zigma_glue_bwd_params_t *p, void *stream);

/* ------------------------------------------------------------------------------------------
 * Cross-attention core over a short context:  out = softmax(scale * Q K^T) V  per (sample, head), no mask.
 * Replaces the scaled_dot_product_attention / xformers call of CrossAttention.forward (reference model_zigma.py:113-127;
 * ZigMa: 8 heads x 64, 77 text tokens).  q, out: (batch, seqlen, heads*head_dim) rows; k, v: (batch, n_ctx, heads*head_dim)
 * rows (any row / batch pitch: e.g. slices of one batched K/V projection); head h = columns [h*head_dim, (h+1)*head_dim).
 * Limits: bf16, head_dim 64, n_ctx <= 128, rows 16-byte aligned.
 * ------------------------------------------------------------------------------------------ */
typedef struct zigma_xattn_params {
    int32_t batch, seqlen, n_ctx, heads, head_dim;
    int32_t dtype;
    int32_t flags;   /* reserved, must be 0 */
    float scale;     /* head_dim^-0.5 in the reference; must be finite and > 0 (else ZIGMA_ERR_UNSUPPORTED) */
    int64_t q_batch_stride, q_row_stride;
    int64_t k_batch_stride, k_row_stride;
    int64_t v_batch_stride, v_row_stride;
    int64_t o_batch_stride, o_row_stride;
    const void *q, *k, *v;
    void *out;
} zigma_xattn_params_t;

int zigma_cross_attn_fwd(const zigma_xattn_params_t *p, void *stream);

/* ------------------------------------------------------------------------------------------
 * Backward of zigma_cross_attn_fwd (what autograd derives for the scaled_dot_product_attention call of the reference,
 * model_zigma.py:113-127, when it trains; nothing of the forward is saved — the probabilities are recomputed):
 *   dq (batch, seqlen, heads*head_dim) bf16;  dk_part, dv_part: fp32 partial sums, contiguous
 *   (chunks, batch, n_ctx, heads*head_dim) with chunks = zigma_cross_attn_bwd_chunks(seqlen) — the caller adds the chunks
 *   (deterministic: no atomics).  Same operand layout and limits as the forward; dout has the layout of out.
 * ------------------------------------------------------------------------------------------ */
typedef struct zigma_xattn_bwd_params {
    int32_t batch, seqlen, n_ctx, heads, head_dim;
    int32_t dtype;
    int32_t flags;   /* reserved, must be 0 */
    float scale;
    int32_t chunks;  /* zigma_cross_attn_bwd_chunks(seqlen) */
    int32_t pad_;
    int64_t q_batch_stride, q_row_stride;
    int64_t k_batch_stride, k_row_stride;
    int64_t v_batch_stride, v_row_stride;
    int64_t do_batch_stride, do_row_stride;
    int64_t dq_batch_stride, dq_row_stride;
    const void *q, *k, *v, *dout;
    void *dq, *dk_part, *dv_part;
} zigma_xattn_bwd_params_t;

int zigma_cross_attn_bwd_chunks(int seqlen);
int zigma_cross_attn_bwd(const zigma_xattn_bwd_params_t *p, void *stream);

/* ------------------------------------------------------------------------------------------
 * x_proj: out[m, n] = sum_k x[m, k] * w[n, k]  for the skinny projection of the Mamba block (n = dt_rank + 2 d_state).
 * Replaces F.linear(conv1d_out, x_proj_weight) (reference dis_mamba/mamba_ssm/ops/selective_scan_interface.py:318-322).
 * x: (m, k) rows (the conv output u, token-major); w: (n, k) = x_proj.weight; out: (m, n).  bf16 in, fp32 accumulate,
 * bf16 out.  Limits: n <= 96, k % 256 == 0, x / w rows 16-byte aligned.  Below 16 384 rows (and k <= 1536) the library splits K over the
 * waves of 32-row workgroups and adds the partial tiles in a fixed order (round 5); from 16 384 rows on a workgroup streams 256 rows.
 * ------------------------------------------------------------------------------------------ */
typedef struct zigma_xproj_params {
    int64_t m;
    int32_t n, k;
    int32_t dtype;
    int32_t flags;           /* reserved, must be 0 */
    int64_t x_row_stride, w_row_stride, out_row_stride;
    const void *x, *w;
    void *out;
} zigma_xproj_params_t;

int zigma_x_proj_fwd(const zigma_xproj_params_t *p, void *stream);

/* ------------------------------------------------------------------------------------------
 * conv_x_proj: the depthwise causal conv1d (+ bias, SiLU) over the reordered sequence AND x_proj of its result in one pass:
 *   u[b, k, c]   = silu(conv_bias[c] + sum_{w<4} conv_weight[c, w] * x[b, x_row_index[k - 3 + w], c])     (x[<0] = 0)
 *   out[b*L + k, n] = sum_c u[b, k, c] * w[n, c]
 * Replaces causal_conv1d_fn(..., activation="silu") + F.linear(conv1d_out, x_proj_weight) of MambaInnerFn.forward
 * (reference selective_scan_interface.py:307-322) with the gather of mamba_simple.py:362-370 in the loads.  u (needed by
 * the scan) is written once and not read back.  bf16 throughout, fp32 accumulation; u is rounded to bf16 BEFORE the
 * projection, as in the reference.  x: (batch, seqlen, dim) channel-contiguous rows; conv_weight: (dim, 4) contiguous;
 * conv_bias: (dim); w: (n, dim) rows; u: (batch, seqlen, dim) in scan order; out: (batch * seqlen, n) rows.
 * Limits: width 4, bias required, seqlen % 32 == 0, batch * seqlen % 256 == 0, dim % 64 == 0, n <= 96 and n % 8 == 0, 16-byte
 * aligned rows (x, u, w, out).
 * flags: 0; probes: 1 = three LDS stages, 2 = eight-wave workgroups, 4 / 8 = phases skipped (results wrong).
 * ------------------------------------------------------------------------------------------ */
typedef struct zigma_conv_xproj_params {
    int32_t batch, seqlen, dim, n;
    int32_t dtype;           /* ZIGMA_BF16 */
    int32_t flags;
    int64_t x_batch_stride, x_l_stride;
    int64_t u_batch_stride, u_l_stride;
    int64_t w_row_stride, out_row_stride;
    const void *x, *conv_weight, *conv_bias, *w;
    void *u, *out;
    const int32_t *x_row_index;   /* or NULL */
} zigma_conv_xproj_params_t;

int zigma_conv_x_proj_fwd(const zigma_conv_xproj_params_t *p, void *stream);

/* ------------------------------------------------------------------------------------------
 * Dense projection on the matrix cores:  out = x @ w^T (+ bias) (+ SiLU on a column range), bf16 in / fp32 accumulate / bf16 out.
 * Replaces the cuBLAS GEMMs behind F.linear at Mamba.in_proj (reference mamba_simple.py:290-294), out_proj
 * (selective_scan_interface.py:365) and CrossAttention.to_q / to_out (model_zigma.py:104-135).
 * x: (m, k) rows; w: (n, k) rows (nn.Linear layout); out: (m, n) rows; bias: bf16 (n) or NULL.
 * silu_from_col: output columns >= this value leave as silu(value) (in_proj writes silu(z) for the gate half, consumed by
 * the scan under ZIGMA_SCAN_Z_PREACTIVATED); pass n for a plain projection.  Must be a multiple of 32.
 * Limits: bf16; k % 64 == 0; n % 128 == 0; x / w rows 16-byte aligned, out rows 8-byte aligned.
 * ZIGMA_LINEAR_WS (flags): the weight-stationary kernel (csrc/linear_ws.hip — a panel of w lives in the registers of a workgroup, only
 * the rows of x stream; the in_proj of the default path).  Same result bit for bit.  Limits, else ZIGMA_ERR_UNSUPPORTED: no bias /
 * residual; k = 512 or 640 with 256-feature panels (pw = 256), or — since round 5 — k = 1280 or 1536 with 128-feature panels (pw = 128:
 * out_proj below the tiled kernel's token floor); n % pw == 0 and n <= 8192, m % 512 == 0 with m / 512 >= 32 / (n / pw), x rows a
 * multiple of 128 elements apart, out rows 16-byte aligned.  silu_from_col < n (round 5, pw = 256 only): a multiple of 128 — whole
 * 128-column groups leave as silu(.) of the fp32 accumulator.
 * ZIGMA_LINEAR_SM (flags; round 5): the few-token tiled kernel (csrc/linear_sm.hip: tiles of 128 rows x n / 4 columns, one per workgroup —
 * 8192 rows x 640 columns are exactly 256 tiles).  Same result bit for bit.  Limits, else ZIGMA_ERR_UNSUPPORTED: no bias / activation /
 * residual, k % 64 == 0 and k >= 128, m % 128 == 0 (n % 128 == 0: tiles of 160, 192 or 128 columns), out rows 16-byte aligned.
 * ------------------------------------------------------------------------------------------ */
#define ZIGMA_LINEAR_WS 0x4000
#define ZIGMA_LINEAR_SM 0x8000
typedef struct zigma_linear_params {
    int64_t m;
    int32_t n, k;
    int32_t dtype;           /* ZIGMA_BF16 */
    int32_t flags;           /* 0, or ZIGMA_LINEAR_WS (other bits: probes of tools/, refused by the shipped library) */
    int32_t silu_from_col;
    int32_t pad_;
    int64_t x_row_stride, w_row_stride, out_row_stride;
    const void *x, *w;
    const void *bias;        /* or NULL */
    void *out;
    /* optional gated residual in the epilogue (ABI 4) — CrossAttention's `hidden + gate_msa * to_out(...)` (model_zigma.py:447-449):
     *   out[m, :] = residual[m, :] + gate[m / rows_per_batch, :] * bf16(x @ w^T + bias)[m, :]
     * residual: (m, n) bf16 rows of pitch res_row_stride; gate: (m / rows_per_batch, n) bf16 rows of pitch gate_batch_stride;
     * rows_per_batch % 256 == 0.  residual == NULL: plain projection. */
    const void *residual, *gate;
    int64_t res_row_stride, gate_batch_stride;
    int32_t rows_per_batch, pad2_;
} zigma_linear_params_t;

int zigma_linear_fwd(const zigma_linear_params_t *p, void *stream);

/* ------------------------------------------------------------------------------------------
 * The small per-forward operators around the blocks (bf16 models; each replaces a chain of library GEMM + ATen elementwise launches):
 *
 * zigma_patch_embed_fwd: PatchEmbed (a conv with kernel == stride; timm, call site model_zigma.py:608-614,924) + bias, and the
 *   position table of model_zigma.py:939-940:  out[b, l, :] = bf16(bf16(W x_patch(b, l) + bias) + pos[l, :]).
 *   x (batch, in_chans, height, width) with element strides (w stride 1); weight (embed_dim, in_chans * patch * patch) contiguous;
 *   bias (embed_dim) or NULL; pos (L, embed_dim) rows of pitch pos_row_stride or NULL; out (batch, L, embed_dim), L = (h / p)(w / p).
 *   Limits: embed_dim % 8 == 0, in_chans * patch^2 * embed_dim * 4 <= 64 KB, out / pos / bias 16-byte aligned.
 * zigma_timestep_embed_fwd: TimestepEmbedder.timestep_embedding (model_zigma.py:247-268): out[b] = [cos(t_b f), sin(t_b f)] with t and the
 *   dim / 2 frequencies f in the model dtype, product and functions in fp32.
 * zigma_final_layer_fwd: FinalLayer without conditioning (model_zigma.py:313-337): out = Linear(LayerNorm(x, no affine, eps)), n_out <= 16,
 *   cols % 8 == 0, cols <= 2048; x rows 16-byte aligned; out (rows, n_out) rows of pitch out_row_stride.
 * zigma_skinny_linear_fwd: out = act(x) W^T + bias for m <= 64 rows (the timestep MLP :232-275 and the adaLN modulation :441, :447 of all
 *   blocks in one call); flags bit 0: act = SiLU (rounded to bf16 like the reference's module), else identity.
 *   Limits: n % 16 == 0, k % 128 == 0, k <= 1024; x, w rows 16-byte aligned, out rows 8-byte aligned, bias 8-byte aligned.
 * ------------------------------------------------------------------------------------------ */
typedef struct zigma_patch_embed_params {
    int32_t batch, in_chans, height, width, patch, embed_dim;
    int32_t dtype;   /* ZIGMA_BF16 */
    int32_t flags;   /* reserved, must be 0 */
    int64_t x_batch_stride, x_chan_stride, x_row_stride;
    int64_t pos_row_stride, out_batch_stride, out_row_stride;
    const void *x, *weight, *bias, *pos;
    void *out;
} zigma_patch_embed_params_t;

typedef struct zigma_timestep_embed_params {
    int32_t batch, dim;
    int32_t dtype;   /* ZIGMA_BF16 */
    int32_t flags;   /* reserved, must be 0 */
    int64_t out_row_stride;
    const void *t, *freqs;
    void *out;
} zigma_timestep_embed_params_t;

typedef struct zigma_final_layer_params {
    int64_t rows;
    int32_t cols, n_out;
    int32_t dtype;   /* ZIGMA_BF16 */
    int32_t flags;   /* reserved, must be 0 */
    float eps;
    int32_t pad_;
    int64_t x_row_stride, out_row_stride;
    const void *x, *weight, *bias;
    void *out;
} zigma_final_layer_params_t;

typedef struct zigma_skinny_params {
    int32_t m, n, k;
    int32_t dtype;   /* ZIGMA_BF16 */
    int32_t flags;   /* bit 0: SiLU on x */
    int32_t pad_;
    int64_t x_row_stride, w_row_stride, out_row_stride;
    const void *x, *w, *bias;
    void *out;
} zigma_skinny_params_t;

int zigma_patch_embed_fwd(const zigma_patch_embed_params_t *p, void *stream);
int zigma_timestep_embed_fwd(const zigma_timestep_embed_params_t *p, void *stream);
int zigma_final_layer_fwd(const zigma_final_layer_params_t *p, void *stream);
int zigma_skinny_linear_fwd(const zigma_skinny_params_t *p, void *stream);

/* ------------------------------------------------------------------------------------------
 * Box calibration (bench.py `box_calib`; csrc/calib.hip).  NOT part of the drop-in surface: no reference interface maps to it.  Two fixed
 * kernels whose times tell a slower box from slower code between rounds: mode 0 copies `bytes` (a multiple of 4096, 16-byte aligned
 * pointers) from src to dst; mode 1 / 2 run `iters` x 16 v_fma_f32 / `iters` x 8 v_exp_f32 per wave on 1280 workgroups of 4 waves (5 waves
 * per SIMD on 256 CUs) and write one float per thread to dst (`bytes` >= 1280 * 256 * 4 is the size of that buffer).
 * ------------------------------------------------------------------------------------------ */
typedef struct zigma_calib_params {
    int32_t mode, iters;
    int64_t bytes;
    const void *src;
    void *dst;
} zigma_calib_params_t;
int zigma_calib_launch(const zigma_calib_params_t *p, void *stream);

/* ------------------------------------------------------------------------------------------ */
const char *zigma_strerror(int status);
int zigma_abi_version(void);
/* Name of the kernel variant the last dispatch on this thread selected (for tests / profiling). */
const char *zigma_last_kernel(void);

#ifdef __cplusplus
}
#endif
#endif /* ZIGMA_HIP_H_ */
