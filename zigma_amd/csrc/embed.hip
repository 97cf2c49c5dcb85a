// The small per-forward operators around the blocks, gfx950: patch embedding (+ bias + position table), sinusoidal timestep
// features, final LayerNorm + output projection.  C ABI: zigma_patch_embed_fwd, zigma_timestep_embed_fwd, zigma_final_layer_fwd.
//
// Reference: PatchEmbed (timm; ZigMa call site model_zigma.py:608-614,924) followed by `+ pos_embed` (:939-940);
// TimestepEmbedder.timestep_embedding (:247-268); FinalLayer.forward without conditioning (:313-337, built with cond=False :817-821).
// None of them is GEMM-shaped (K = 3 input channels; 3 output channels): as library GEMMs + ATen elementwise kernels they cost
// 120 + 60 + 60 us of a 18.9 ms forward in ~20 launches of ~5 us minimum each; here each is ONE HBM-bound pass.
// Rounding points follow the reference's bf16 evaluation: conv + bias -> bf16, + pos -> bf16; LayerNorm -> bf16, linear + bias -> bf16.
#include "zigma_common.h"

namespace zigma {

// ---------------------------------------------------------------------------------------------------------------------------
// patch embedding: out[b, l, e] = bf16( bf16( sum_k w[e, k] x[b, patch l, k] + bias[e] ) + pos[l, e] ),  k = (c, dy, dx)
// block -> 16 tokens x 16 feature lanes, every thread walks its token's features in steps of 128 (the token's index arithmetic is paid once
// per thread, 256 contiguous bytes leave per token and step); the weight sits transposed in LDS as fp32 [k][e]: a thread's 8 features of one
// k are two 16-byte reads; a token's K inputs come from the cache.
// KT = in_chans * patch^2 as a compile-time constant (3, 4, 12, 16; 0 = any): the token's KT inputs, the bias and position pieces of a chunk are
// ALL requested before the first product — with run-time loops every input, bias and position piece was a load followed by its own wait
// (25 exposed latencies per thread: 45 us for the 84 MB of the headline shape)
template <int KT>
__global__ __launch_bounds__(256) void patch_embed_kernel(const zigma_patch_embed_params_t p) {
    extern __shared__ __attribute__((aligned(16))) float s_w[];          // [K][E]
    const int E = p.embed_dim, K = KT ? KT : p.in_chans * p.patch * p.patch, E8 = E / 8;
    for (int piece = threadIdx.x; piece < (E * K) / 8; piece += blockDim.x) {        // 16-byte pieces of the (E, K) weight; one division per piece
        const uint4 wv = reinterpret_cast<const uint4 *>(p.weight)[piece];
        const uint32_t ww[4] = {wv.x, wv.y, wv.z, wv.w};
        int e = (piece * 8) / K, k = (piece * 8) % K;
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            s_w[k * E + e] = to_float<BF16>(static_cast<uint16_t>(ww[i >> 1] >> ((i & 1) * 16)));
            if (++k == K) { k = 0; ++e; }
        }
    }
    __syncthreads();
    const int gw = p.width / p.patch, gh = p.height / p.patch, L = gw * gh, pp = p.patch * p.patch;
    const uint16_t *x = reinterpret_cast<const uint16_t *>(p.x);
    // thread -> (token of the block's 16, feature lane of 16): the token's index arithmetic once per thread, then E / 128 pieces of 8 features;
    // blockIdx.y = sample (no division by L)
    const int jl = threadIdx.x & 15, tl = threadIdx.x >> 4;
    const int b = blockIdx.y;
    for (int l = blockIdx.x * 16 + tl; l < L; l += gridDim.x * 16) {
        const int py = l / gw, px = l % gw;
        const uint16_t *xt = x + static_cast<int64_t>(b) * p.x_batch_stride + (py * p.patch) * p.x_row_stride + px * p.patch;
        uint16_t *orow = reinterpret_cast<uint16_t *>(p.out) + static_cast<int64_t>(b) * p.out_batch_stride + static_cast<int64_t>(l) * p.out_row_stride;
        const uint16_t *prow = p.pos ? reinterpret_cast<const uint16_t *>(p.pos) + l * p.pos_row_stride : nullptr;
        uint16_t xr[KT ? KT : 1];
        if constexpr (KT > 0) {
#pragma unroll
            for (int k = 0; k < KT; ++k) {
                const int c = k / pp, r = k % pp;
                xr[k] = xt[c * p.x_chan_stride + (r / p.patch) * p.x_row_stride + (r % p.patch)];
            }
        }
        // chunks of 8 pieces per thread (1024 features per token and chunk)
        for (int j0 = jl; j0 < E8; j0 += 16 * 8) {
            uint4 bv[8], pv[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int j = j0 + 16 * q;
                bv[q] = pv[q] = make_uint4(0, 0, 0, 0);
                if (j < E8) {
                    if (p.bias) bv[q] = *reinterpret_cast<const uint4 *>(reinterpret_cast<const uint16_t *>(p.bias) + j * 8);
                    if (prow) pv[q] = *reinterpret_cast<const uint4 *>(prow + j * 8);
                }
            }
            uint32_t ov[8][4];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int j = j0 + 16 * q;
                if (j >= E8) break;
                float acc[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                auto mac = [&](int k, float xv) {
                    const float4 w0 = *reinterpret_cast<const float4 *>(&s_w[k * E + j * 8]), w1 = *reinterpret_cast<const float4 *>(&s_w[k * E + j * 8 + 4]);
                    acc[0] = __builtin_fmaf(w0.x, xv, acc[0]); acc[1] = __builtin_fmaf(w0.y, xv, acc[1]);
                    acc[2] = __builtin_fmaf(w0.z, xv, acc[2]); acc[3] = __builtin_fmaf(w0.w, xv, acc[3]);
                    acc[4] = __builtin_fmaf(w1.x, xv, acc[4]); acc[5] = __builtin_fmaf(w1.y, xv, acc[5]);
                    acc[6] = __builtin_fmaf(w1.z, xv, acc[6]); acc[7] = __builtin_fmaf(w1.w, xv, acc[7]);
                };
                if constexpr (KT > 0) {
#pragma unroll
                    for (int k = 0; k < KT; ++k) mac(k, to_float<BF16>(xr[k]));
                } else {
                    int k = 0;
                    for (int c = 0; c < p.in_chans; ++c)
                        for (int dy = 0; dy < p.patch; ++dy)
                            for (int dx = 0; dx < p.patch; ++dx, ++k) mac(k, to_float<BF16>(xt[c * p.x_chan_stride + dy * p.x_row_stride + dx]));
                }
                uint16_t o[8];
                {
                    const uint32_t bw[4] = {bv[q].x, bv[q].y, bv[q].z, bv[q].w};         // (zeros without a bias: the sum is unchanged)
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] += to_float<BF16>(static_cast<uint16_t>(bw[i >> 1] >> ((i & 1) * 16)));
                }
#pragma unroll
                for (int i = 0; i < 8; ++i) o[i] = from_float<BF16>(acc[i]);
                if (prow) {
                    const uint32_t pw[4] = {pv[q].x, pv[q].y, pv[q].z, pv[q].w};
#pragma unroll
                    for (int i = 0; i < 8; ++i)
                        o[i] = from_float<BF16>(to_float<BF16>(o[i]) + to_float<BF16>(static_cast<uint16_t>(pw[i >> 1] >> ((i & 1) * 16))));
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) ov[q][i] = o[2 * i] | (static_cast<uint32_t>(o[2 * i + 1]) << 16);
            }
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                const int j = j0 + 16 * q;
                if (j >= E8) break;
                *reinterpret_cast<uint4 *>(orow + j * 8) = make_uint4(ov[q][0], ov[q][1], ov[q][2], ov[q][3]);
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------------------
// timestep features: out[b, j] = bf16(cos(t_b f_j)), out[b, half + j] = bf16(sin(t_b f_j)); t and the frequency table in the model dtype
// (the reference forms both in it, :259-262), product and functions in fp32
__global__ void timestep_embed_kernel(const zigma_timestep_embed_params_t p) {
    const int half = p.dim / 2;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= p.batch * half) return;
    const int b = i / half, j = i % half;
    const float arg = to_float<BF16>(reinterpret_cast<const uint16_t *>(p.t)[b]) * to_float<BF16>(reinterpret_cast<const uint16_t *>(p.freqs)[j]);
    uint16_t *o = reinterpret_cast<uint16_t *>(p.out) + static_cast<int64_t>(b) * p.out_row_stride;
    o[j] = from_float<BF16>(cosf(arg));
    o[half + j] = from_float<BF16>(sinf(arg));
    if ((p.dim & 1) && j == 0) o[p.dim - 1] = 0;
}

// ---------------------------------------------------------------------------------------------------------------------------
// final layer: y = bf16(LayerNorm(x, no affine)); out[r, o] = bf16(sum_e w[o, e] y[e] + bias[o]),  n_out <= 16
// 16 lanes per row (4 rows per wave, 16 per workgroup): the row (E <= 2048 features, 8 per lane and pass of 128) stays in registers for the
// two statistics passes and the n_out dot products; sums over the 16 lanes are 4 cross-lane steps; w sits in LDS as bf16 and the 4 rows of
// a wave read the same pieces (broadcast)
constexpr int kFlMaxPass = 16;      // E <= 16 lanes * 8 * 16
template <int NP>                   // passes: E <= 128 NP
__global__ __launch_bounds__(256) void final_layer_kernel(const zigma_final_layer_params_t p) {
    extern __shared__ __attribute__((aligned(16))) uint16_t s_fw[];     // [n_out][E]
    const int E = p.cols, NO = p.n_out;
    for (int i = threadIdx.x; i < NO * E / 8; i += blockDim.x)
        reinterpret_cast<uint4 *>(s_fw)[i] = reinterpret_cast<const uint4 *>(p.weight)[i];
    __syncthreads();
    const int l16 = threadIdx.x & 15, rl = threadIdx.x >> 4;
    const float inv_e = 1.f / static_cast<float>(E);
    auto sum16 = [](float v) {
#pragma unroll
        for (int m = 8; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
        return v;
    };
    for (int64_t r0 = static_cast<int64_t>(blockIdx.x) * 16; r0 < p.rows; r0 += static_cast<int64_t>(gridDim.x) * 16) {
        const int64_t r = r0 + rl;
        const bool live = r < p.rows;
        const uint16_t *xr = reinterpret_cast<const uint16_t *>(p.x) + (live ? r : 0) * p.x_row_stride;
        float v[NP][8];
        float s = 0.f;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int e0 = q * 128 + l16 * 8;
            uint4 w = make_uint4(0, 0, 0, 0);
            if (e0 < E) w = *reinterpret_cast<const uint4 *>(xr + e0);
            const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
            for (int i = 0; i < 8; ++i) { v[q][i] = to_float<BF16>(static_cast<uint16_t>(ww[i >> 1] >> ((i & 1) * 16))); s += v[q][i]; }
        }
        const float mean = sum16(s) * inv_e;
        float s2 = 0.f;
#pragma unroll
        for (int q = 0; q < NP; ++q) {
            const int e0 = q * 128 + l16 * 8;
#pragma unroll
            for (int i = 0; i < 8; ++i) { const float d = e0 < E ? v[q][i] - mean : 0.f; v[q][i] = d; s2 += d * d; }
        }
        const float rstd = rsqrtf(sum16(s2) * inv_e + p.eps);
#pragma unroll
        for (int q = 0; q < NP; ++q)
#pragma unroll
            for (int i = 0; i < 8; ++i) v[q][i] = to_float<BF16>(from_float<BF16>(v[q][i] * rstd));      // the LayerNorm output is a bf16 tensor
        float res = 0.f;      // lane o of the row's 16 keeps output o
        for (int o = 0; o < NO; ++o) {
            float acc = 0.f;
#pragma unroll
            for (int q = 0; q < NP; ++q) {
                const int e0 = q * 128 + l16 * 8;
                if (e0 < E) {
                    const uint4 w = *reinterpret_cast<const uint4 *>(s_fw + o * E + e0);
                    const uint32_t ww[4] = {w.x, w.y, w.z, w.w};
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc = __builtin_fmaf(to_float<BF16>(static_cast<uint16_t>(ww[i >> 1] >> ((i & 1) * 16))), v[q][i], acc);
                }
            }
            acc = sum16(acc);
            if (l16 == o) res = acc;
        }
        if (live && l16 < NO) {
            if (p.bias) res += to_float<BF16>(reinterpret_cast<const uint16_t *>(p.bias)[l16]);
            reinterpret_cast<uint16_t *>(p.out)[r * p.out_row_stride + l16] = from_float<BF16>(res);
        }
    }
}

}  // namespace zigma

using namespace zigma;

extern "C" int zigma_patch_embed_fwd(const zigma_patch_embed_params_t *pp, void *stream_) {
    if (!pp) return ZIGMA_ERR_NULL;
    (void)hipGetLastError();
    const zigma_patch_embed_params_t &p = *pp;
    if (p.batch < 0 || p.in_chans < 1 || p.patch < 1 || p.embed_dim < 8 || p.height < 1 || p.width < 1) return ZIGMA_ERR_SHAPE;
    if (p.flags != 0) return ZIGMA_ERR_UNSUPPORTED;
    if (p.dtype != ZIGMA_BF16) return ZIGMA_ERR_DTYPE;
    if (p.embed_dim % 8 != 0 || p.height % p.patch != 0 || p.width % p.patch != 0) return ZIGMA_ERR_SHAPE;
    const int64_t K = static_cast<int64_t>(p.in_chans) * p.patch * p.patch, lds = K * p.embed_dim * 4;
    if (lds > 65536) return ZIGMA_ERR_SHAPE;
    if (p.batch == 0) return ZIGMA_OK;
    if (!p.x || !p.weight || !p.out) return ZIGMA_ERR_NULL;
    if (reinterpret_cast<uintptr_t>(p.out) % 16 != 0 || p.out_row_stride % 8 != 0 || p.out_batch_stride % 8 != 0) return ZIGMA_ERR_STRIDE;
    if (reinterpret_cast<uintptr_t>(p.weight) % 16 != 0) return ZIGMA_ERR_STRIDE;
    if (p.bias && reinterpret_cast<uintptr_t>(p.bias) % 16 != 0) return ZIGMA_ERR_STRIDE;
    if (p.pos && (reinterpret_cast<uintptr_t>(p.pos) % 16 != 0 || p.pos_row_stride % 8 != 0)) return ZIGMA_ERR_STRIDE;
    const int64_t L = static_cast<int64_t>(p.height / p.patch) * (p.width / p.patch);
    if (L > 0x7fffffff || p.batch > 65535) return ZIGMA_ERR_SHAPE;
    const int64_t want = (L + 15) / 16;
    const int64_t per = p.batch >= 64 ? 16 : (p.batch >= 8 ? 64 : 1024);      // ~1024+ workgroups, each staging the weight once for several tokens
    const dim3 grid(static_cast<unsigned>(want < per ? want : per), static_cast<unsigned>(p.batch));
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    switch (K) {
        case 3: hipLaunchKernelGGL(patch_embed_kernel<3>, grid, dim3(256), static_cast<size_t>(lds), stream, p); break;
        case 4: hipLaunchKernelGGL(patch_embed_kernel<4>, grid, dim3(256), static_cast<size_t>(lds), stream, p); break;
        case 12: hipLaunchKernelGGL(patch_embed_kernel<12>, grid, dim3(256), static_cast<size_t>(lds), stream, p); break;
        case 16: hipLaunchKernelGGL(patch_embed_kernel<16>, grid, dim3(256), static_cast<size_t>(lds), stream, p); break;
        default: hipLaunchKernelGGL(patch_embed_kernel<0>, grid, dim3(256), static_cast<size_t>(lds), stream, p); break;
    }
    set_last_kernel("patch_embed");
    return check_launch();
}

extern "C" int zigma_timestep_embed_fwd(const zigma_timestep_embed_params_t *pp, void *stream_) {
    if (!pp) return ZIGMA_ERR_NULL;
    (void)hipGetLastError();
    const zigma_timestep_embed_params_t &p = *pp;
    if (p.batch < 0 || p.dim < 2) return ZIGMA_ERR_SHAPE;
    if (p.flags != 0) return ZIGMA_ERR_UNSUPPORTED;
    if (p.dtype != ZIGMA_BF16) return ZIGMA_ERR_DTYPE;
    if (p.batch == 0) return ZIGMA_OK;
    if (!p.t || !p.freqs || !p.out) return ZIGMA_ERR_NULL;
    const int n = p.batch * (p.dim / 2);
    hipLaunchKernelGGL(timestep_embed_kernel, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream_), p);
    set_last_kernel("timestep_embed");
    return check_launch();
}

extern "C" int zigma_final_layer_fwd(const zigma_final_layer_params_t *pp, void *stream_) {
    if (!pp) return ZIGMA_ERR_NULL;
    (void)hipGetLastError();
    const zigma_final_layer_params_t &p = *pp;
    if (p.rows < 0 || p.cols < 8 || p.n_out < 1) return ZIGMA_ERR_SHAPE;
    if (p.flags != 0) return ZIGMA_ERR_UNSUPPORTED;
    if (p.dtype != ZIGMA_BF16) return ZIGMA_ERR_DTYPE;
    if (p.cols % 8 != 0 || p.cols > 128 * kFlMaxPass || p.n_out > 16) return ZIGMA_ERR_SHAPE;
    if (p.rows == 0) return ZIGMA_OK;
    if (!p.x || !p.weight || !p.out) return ZIGMA_ERR_NULL;
    if (reinterpret_cast<uintptr_t>(p.x) % 16 != 0 || p.x_row_stride % 8 != 0 || reinterpret_cast<uintptr_t>(p.weight) % 16 != 0) return ZIGMA_ERR_STRIDE;
    const int64_t want = (p.rows + 15) / 16;
    const dim3 grid(static_cast<unsigned>(want < 4096 ? want : 4096)), block(256);
    const size_t lds = static_cast<size_t>(p.n_out) * p.cols * 2;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    const int np = (p.cols + 127) / 128;
    if (np <= 5) hipLaunchKernelGGL(final_layer_kernel<5>, grid, block, lds, stream, p);
    else if (np <= 8) hipLaunchKernelGGL(final_layer_kernel<8>, grid, block, lds, stream, p);
    else hipLaunchKernelGGL(final_layer_kernel<16>, grid, block, lds, stream, p);
    set_last_kernel("final_layer");
    return check_launch();
}
