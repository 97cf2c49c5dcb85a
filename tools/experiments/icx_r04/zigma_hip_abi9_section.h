/* the section include/zigma_hip.h carried while zigma_in_conv_x_proj_fwd was part of the library (ABI 9, round 4) */
/* ------------------------------------------------------------------------------------------
 * in_conv_x_proj (ABI 9): the x-HALF of Mamba.in_proj, the gather, the conv, SiLU and x_proj in one kernel — the x half of `xz`
 * never reaches memory:
 *   x[b, t, c]      = bf16( sum_e h[b, t, e] * w_in[c, e] )                                   c < dim   (rows 0..dim-1 of in_proj.weight)
 *   u[b, k, c]      = silu(conv_bias[c] + sum_{w<4} conv_weight[c, w] * x[b, x_row_index[k - 3 + w], c])     (x[<0] = 0)
 *   out[b*L + k, n] = sum_c u[b, k, c] * w[n, c]
 * Replaces the first `dim` output columns of F.linear(hidden_states, in_proj.weight) (reference mamba_simple.py:290-294), the
 * gather xz[:, :, perm] (mamba_simple.py:362-370), causal_conv1d_fn(..., activation="silu") and F.linear(conv1d_out,
 * x_proj_weight) (selective_scan_interface.py:307-322).  The z half of in_proj stays a projection of its own (zigma_linear_fwd
 * on rows dim..2*dim-1 of the weight).  Against in_proj + zigma_conv_x_proj_fwd this saves the write and the re-read of x:
 * 2 * 2 * batch * seqlen * dim bytes.  x is rounded to bf16 before the conv (it is a bf16 tensor in the reference), u before x_proj.
 * h: (batch, seqlen, k) rows, token order; w_in: (dim, k) rows; conv_weight: (dim, 4) contiguous; conv_bias: (dim);
 * w: (n, dim) rows; u: (batch, seqlen, dim) in SCAN order; out: (batch * seqlen, n) rows.
 * A workgroup walks `tiles` consecutive tiles of 128 scan positions; the three x rows in front of its first tile come from a small
 * pre-pass (second kernel of the same call) through `workspace` (zigma_in_conv_x_proj_fwd_workspace_bytes()).
 * Limits: bf16; k % 128 == 0 and k <= 768; dim % 64 == 0 and dim <= 1536; seqlen % 128 == 0; n <= 80, n % 8 == 0; 16-byte aligned rows.
 * flags: 0; timing probes (results wrong): 2 = no x product, 4 = no u stores, 8 = no conv arithmetic, 16 = no W_in stream, 32 = no fragment reads.
 * ------------------------------------------------------------------------------------------ */
typedef struct zigma_in_conv_xproj_params {
    int32_t batch, seqlen, dim, n, k;
    int32_t dtype;           /* ZIGMA_BF16 */
    int32_t flags;
    int32_t pad_;
    int64_t h_batch_stride, h_l_stride;
    int64_t u_batch_stride, u_l_stride;
    int64_t win_row_stride, w_row_stride, out_row_stride;
    const void *h, *w_in, *conv_weight, *conv_bias, *w;
    void *u, *out;
    const int32_t *x_row_index;   /* or NULL */
    void *workspace;
    int64_t workspace_bytes;
} zigma_in_conv_xproj_params_t;

int zigma_in_conv_x_proj_fwd(const zigma_in_conv_xproj_params_t *p, void *stream);
int64_t zigma_in_conv_x_proj_fwd_workspace_bytes(const zigma_in_conv_xproj_params_t *p);
