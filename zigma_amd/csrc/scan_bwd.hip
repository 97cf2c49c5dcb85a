// Selective scan BACKWARD, token-major operands, gfx950.  C ABI: zigma_selective_scan_bwd (include/zigma_hip.h).
//
// Replaces selective_scan_bwd_kernel (reference dis_mamba/csrc/selective_scan/selective_scan_bwd_kernel.cuh:59-329
// + reverse_scan.cuh).  The reference lays the sequence across a thread block, re-runs a block-wide forward scan per
// 2048-step chunk and then a block-wide REVERSE scan of (a, dh) pairs.  Here — same decomposition as the forward
// kernel (scan_tok.inc) — a lane is a channel and time runs sequentially inside it, so the reverse recurrence
//     dh_l = g_l C_l + a_{l+1} dh_{l+1}
// needs no scan at all; what it needs is h_l in reverse order.  288 GB of HBM make the cheap answer affordable:
//   phase 1  forward recurrence over the whole sequence, h written to a checkpoint buffer every 16 steps
//            (batch * dim * dstate * L/16 floats: 335 MB at B=64, L=1024, Di=1280 — the caller's workspace);
//   phase 2  tiles of 16 steps in REVERSE order: reload the checkpoint, recompute the 16 states into registers
//            (64 VGPRs), run the reverse recurrence and every gradient of the tile.
// workgroup = one sample x one 64-channel slab, NW waves x 4 states (as forward).  Per tile:
//   prologue (each wave, its own rows): softplus, sigmoid', g = dout * silu(z), dz -> HBM;  dt, u, g -> LDS
//   core (all waves, all 16 steps, own 4 states): dh, dA; per-step partial sums over the wave's states of
//        dh*B and A*dh*a*h_{l-1} -> LDS;  the cross-CHANNEL sums dB_l = sum_d dh dt u, dC_l = sum_d g h_l are
//        transposed through LDS per 4-step group (conflict-free pitch) and summed by 64 lanes in parallel,
//        the per-slab partials go to the workspace and are summed over the slabs by a finishing kernel
//        (fixed order: bit-reproducible, unlike the reference's float atomics)
//   epilogue (own rows): du = dt * sum_n dh B + g D,  ddelta = (u * sum_n dh B + sum_n A dh a h) * sigmoid(dt_raw)
// B_l / C_l: the workgroup widens the tile's 16 x (N + N) values to fp32 once into LDS (one element of each per thread, fetched a tile
// ahead); every wave reads its 4 states of a step with one wave-uniform ds_read_b128 per operand (a broadcast: no VALU slot, plain
// operands) and the recurrences run two states per instruction (v_pk_mul_f32 / v_pk_fma_f32), as in scan_tok2_kernel.
#include "scan_helpers.h"

namespace zigma {

constexpr int kBT = 16;                       // steps per tile
constexpr int kRedPitch = 260;                // floats per step of the transposition buffer (64 lanes x 4 states + 4)

struct BwdWs {
    float *ck;   // [batch][slabs][n_tiles][dstate][64]      h at the START of every tile
    float *bc;   // [batch][slabs][n_tiles*16][2][dstate]    per-slab partial dB, dC
    float *pa;   // [batch][dim][dstate + 2]                 per-sample dA, dD, ddelta_bias
};

template <typename IO, int NW>
__global__ __launch_bounds__(64 * NW, 2) void scan_bwd_kernel(const zigma_scan_bwd_params_t p, const BwdWs ws) {
    constexpr int LT = kBT, RPT = LT / NW, NG = LT / 4, NST = 4 * NW;
    using io_t = typename IO::raw;
    constexpr int ES = static_cast<int>(sizeof(io_t));
    __shared__ float s_dv[LT][64], s_u[LT][64], s_g[LT][64];
    __shared__ __attribute__((aligned(8))) float s_part[NW][LT][64][2];              // (sum_n dh B, sum_n A dh a h) per wave
    __shared__ __attribute__((aligned(16))) float s_red[NW][2][4 * kRedPitch];        // [wave][dB | dC][step in group][lane][state]
    __shared__ __attribute__((aligned(16))) float s_bc[LT][2][16];                    // [step][B | C][state] of the tile, fp32

    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.y, slab = blockIdx.x, n_slabs = gridDim.x;
    const int c = slab * 64 + lane;
    const int L = p.seqlen, N = p.dstate;
    const int n0 = wave * 4;
    const int n_tiles = (L + LT - 1) / LT;
    const bool has_z = p.z != nullptr, sp_on = p.delta_softplus != 0;
    // timing probes (results wrong; probe builds of tools/bwd_probe.py only, compile-time so that the shipped loops stay branch-free):
    // -DZIGMA_SCANBWD_PROBE=mask: 1 no cross-channel reduction, 2 no barriers, 4 no du / ddelta / dz stores, 8 no forward recompute
#ifdef ZIGMA_SCANBWD_PROBE
    constexpr int probe = ZIGMA_SCANBWD_PROBE;
#else
    constexpr int probe = 0;
#endif

    float a2[4], An[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        An[j] = reinterpret_cast<const float *>(p.A)[c * p.A_d_stride + (n0 + j) * p.A_dstate_stride];
        a2[j] = An[j] * kLog2e;
    }
    const float Dv = p.D ? reinterpret_cast<const float *>(p.D)[c] : 0.f;
    const float bias = p.delta_bias ? reinterpret_cast<const float *>(p.delta_bias)[c] : 0.f;

    const unsigned lane_off = static_cast<unsigned>(c) * ES;
    const int64_t Lm1 = L - 1, dimb = static_cast<int64_t>(p.dim) * ES;
    auto row_rsrc = [&](const void *base, int64_t bstride, int64_t lstride) {
        return make_rsrc(reinterpret_cast<const io_t *>(base) + b * bstride, base ? Lm1 * lstride * ES + dimb : 0);
    };
    const int u_ls = static_cast<int>(p.u_l_stride) * ES, d_ls = static_cast<int>(p.delta_l_stride) * ES;
    const int z_ls = static_cast<int>(p.z_l_stride) * ES, o_ls = static_cast<int>(p.out_l_stride) * ES;
    const int do_ls = static_cast<int>(p.dout_l_stride) * ES, du_ls = static_cast<int>(p.du_l_stride) * ES;
    const int dd_ls = static_cast<int>(p.ddelta_l_stride) * ES, dz_ls = static_cast<int>(p.dz_l_stride) * ES;
    const rsrc_t u_rs = row_rsrc(p.u, p.u_batch_stride, p.u_l_stride), d_rs = row_rsrc(p.delta, p.delta_batch_stride, p.delta_l_stride);
    const rsrc_t z_rs = row_rsrc(p.z, p.z_batch_stride, p.z_l_stride), o_rs = row_rsrc(p.out, p.out_batch_stride, p.out_l_stride);
    const rsrc_t do_rs = row_rsrc(p.dout, p.dout_batch_stride, p.dout_l_stride);
    const rsrc_t du_rs = row_rsrc(p.du, p.du_batch_stride, p.du_l_stride), dd_rs = row_rsrc(p.ddelta, p.ddelta_batch_stride, p.ddelta_l_stride);
    const rsrc_t dz_rs = row_rsrc(p.dz, p.dz_batch_stride, p.dz_l_stride);
    const int B_ls = static_cast<int>(p.B_l_stride) * ES, C_ls = static_cast<int>(p.C_l_stride) * ES;
    const rsrc_t B_rs = make_rsrc(reinterpret_cast<const io_t *>(p.B) + b * p.B_batch_stride,
                                  Lm1 * B_ls + ((p.dstate - 1) * p.B_dstate_stride + 1) * ES);
    const rsrc_t C_rs = make_rsrc(reinterpret_cast<const io_t *>(p.C) + b * p.C_batch_stride,
                                  Lm1 * C_ls + ((p.dstate - 1) * p.C_dstate_stride + 1) * ES);
    // B / C staging: thread -> (step, state) of the tile, one element of each (generic strides)
    const int bc_step = static_cast<int>(threadIdx.x) / N, bc_n = static_cast<int>(threadIdx.x) % N;     // 64 NW threads = LT N values
    const unsigned B_lane = static_cast<unsigned>(bc_n * static_cast<int>(p.B_dstate_stride) * ES);
    const unsigned C_lane = static_cast<unsigned>(bc_n * static_cast<int>(p.C_dstate_stride) * ES);
    auto clampk = [&](int k) { return k < L ? k : L - 1; };
    auto bc_raw = [&](rsrc_t rs, unsigned lane_base, int ls, int t) {      // steps beyond L: clamped (they carry dt = g = 0)
        return buf_ld<IO>(rs, lane_base + static_cast<unsigned>(clampk(t * LT + bc_step) * ls), 0);
    };
    float *ck = (p.checkpoints ? const_cast<float *>(p.checkpoints) : ws.ck) + (static_cast<int64_t>(b) * n_slabs + slab) * n_tiles * N * 64;

    // ================================ phase 1: forward, checkpoints ================================================
    v2f a2A = {a2[0], a2[1]}, a2B = {a2[2], a2[3]};
    v2f hA = {0.f, 0.f}, hB = {0.f, 0.f};
    // operands are fetched ONE TILE AHEAD and kept raw (widened where consumed): a workgroup never waits for a load it
    // issued in the same tile (2 waves / SIMD cannot hide HBM latency by occupancy)
    io_t pu[RPT], pd[RPT], pb, pc;
    auto fetch_fwd = [&](int t) {
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int kk = clampk(t * LT + wave * RPT + i);
            pu[i] = buf_ld<IO>(u_rs, lane_off, kk * u_ls);
            pd[i] = buf_ld<IO>(d_rs, lane_off, kk * d_ls);
        }
        pb = bc_raw(B_rs, B_lane, B_ls, t);
    };
    const bool own_ck = p.checkpoints == nullptr;      // else: the forward kernel already wrote them
    if (own_ck) fetch_fwd(0);
#pragma unroll 1
    for (int t = 0; t < (own_ck ? n_tiles : 0); ++t) {
        if (p.reset_period > 0 && (t * LT) % p.reset_period == 0) hA = hB = v2f{0.f, 0.f};      // start of an independent sequence
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int row = wave * RPT + i, k = t * LT + row;
            const float uf = to_float<IO>(pu[i]);
            float dv = to_float<IO>(pd[i]) + bias;
            if (sp_on) dv = softplus20(dv);
            if (k >= L) dv = 0.f;
            s_dv[row][lane] = dv;
            s_u[row][lane] = k < L ? uf : 0.f;
        }
        s_bc[bc_step][0][bc_n] = to_float<IO>(pb);
        if (t + 1 < n_tiles) fetch_fwd(t + 1);
        ck[(static_cast<int64_t>(t) * N + n0 + 0) * 64 + lane] = hA.x;
        ck[(static_cast<int64_t>(t) * N + n0 + 1) * 64 + lane] = hA.y;
        ck[(static_cast<int64_t>(t) * N + n0 + 2) * 64 + lane] = hB.x;
        ck[(static_cast<int64_t>(t) * N + n0 + 3) * 64 + lane] = hB.y;
        __syncthreads();
#pragma unroll
        for (int s = 0; s < LT; ++s) {
            const float dv = s_dv[s][lane], du = dv * s_u[s][lane];
            const v4f Bv = *reinterpret_cast<const v4f *>(&s_bc[s][0][n0]);
            const v2f dtv = {dv, dv}, duv = {du, du};
            const v2f xA = a2A * dtv, xB = a2B * dtv;
            const v2f eA = {fast_exp2(xA.x), fast_exp2(xA.y)}, eB = {fast_exp2(xB.x), fast_exp2(xB.y)};
            hA = __builtin_elementwise_fma(eA, hA, v2f{Bv.x, Bv.y} * duv);
            hB = __builtin_elementwise_fma(eB, hB, v2f{Bv.z, Bv.w} * duv);
        }
        __syncthreads();
    }

    // ================================ phase 2: reverse sweep =========================================================
    v2f adhA = {0.f, 0.f}, adhB = {0.f, 0.f}, dAA = {0.f, 0.f}, dAB = {0.f, 0.f};
    const v2f AnA = {An[0], An[1]}, AnB = {An[2], An[3]};
    float dD_acc = 0.f, db_acc = 0.f;
    float *bc_out = ws.bc + (static_cast<int64_t>(b) * n_slabs + slab) * n_tiles * LT * 2 * N;
    float *red_b = &s_red[wave][0][0], *red_c = &s_red[wave][1][0];
    // reduction role of this lane inside a 4-step group: (which, step, state) = 32 sums, each split over two half waves
    const int r_half = lane >> 5, r_which = (lane >> 4) & 1, r_si = (lane >> 2) & 3, r_j = lane & 3;
    const float *red_src = &s_red[wave][r_which][r_si * kRedPitch + r_half * 32 * 4 + r_j];

    io_t pdo[RPT], pz[RPT], po[RPT];
    float h0n[4];
    int zi_c = 0, oi_c = 0, zi_n = 0, oi_n = 0;   // row tables (lane i mod RPT <- row i of this wave) of the tile in the
                                                  // prefetch registers / of the tile after it
    auto load_tabs = [&](int t) {
        const int kt = clampk(t * LT + wave * RPT + (lane & (RPT - 1)));
        if (p.z_row_index) zi_n = p.z_row_index[kt];
        if (p.out_row_index) oi_n = p.out_row_index[kt];
    };
    auto fetch_bwd = [&](int t) {                 // consumes (zi_n, oi_n) as the tables of tile t
        zi_c = zi_n;
        oi_c = oi_n;
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int kk = clampk(t * LT + wave * RPT + i);
            const int zrow = p.z_row_index ? __builtin_amdgcn_readlane(zi_c, i) : kk;
            const int orow = p.out_row_index ? __builtin_amdgcn_readlane(oi_c, i) : kk;
            pu[i] = buf_ld<IO>(u_rs, lane_off, kk * u_ls);
            pd[i] = buf_ld<IO>(d_rs, lane_off, kk * d_ls);
            pdo[i] = buf_ld<IO>(do_rs, lane_off, orow * do_ls);
            if (has_z) {
                pz[i] = buf_ld<IO>(z_rs, lane_off, zrow * z_ls);
                po[i] = buf_ld<IO>(o_rs, lane_off, orow * o_ls);
            }
        }
        pb = bc_raw(B_rs, B_lane, B_ls, t);
        pc = bc_raw(C_rs, C_lane, C_ls, t);
#pragma unroll
        for (int j = 0; j < 4; ++j) h0n[j] = ck[(static_cast<int64_t>(t) * N + n0 + j) * 64 + lane];
    };
    load_tabs(n_tiles - 1);
    fetch_bwd(n_tiles - 1);
    if (n_tiles > 1) load_tabs(n_tiles - 2);

#pragma unroll 1
    for (int t = n_tiles - 1; t >= 0; --t) {
        // tile t + 1 started an independent sequence: nothing flows back from it
        if (p.reset_period > 0 && ((t + 1) * LT) % p.reset_period == 0) adhA = adhB = v2f{0.f, 0.f};
        // ---- prologue: own rows (operands of this tile are in the prefetch registers) ----------------------------
        float dvr[RPT], ur[RPT], gr[RPT], sgr[RPT];
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int row = wave * RPT + i, k = t * LT + row, kk = clampk(k);
            const int zrow = p.z_row_index ? __builtin_amdgcn_readlane(zi_c, i) : kk;
            const float uf = to_float<IO>(pu[i]);
            const float draw = to_float<IO>(pd[i]) + bias;
            const float dof = to_float<IO>(pdo[i]);
            float dv = draw, sg = 1.f;
            if (sp_on) {
                dv = softplus20(draw);
                sg = draw <= 20.f ? fast_rcp(1.f + fast_exp2(-draw * kLog2e)) : 1.f;
            }
            float g = dof;
            if (has_z) {
                const float zf = to_float<IO>(pz[i]);
                const float yf = to_float<IO>(po[i]);
                const float sz = fast_rcp(1.f + fast_exp2(-zf * kLog2e));
                g = dof * zf * sz;
                const float dz = dof * yf * sz * (1.f + zf * (1.f - sz));
                if (k < L) buf_st<IO>(from_float<IO>(dz), dz_rs, lane_off, zrow * dz_ls);
            }
            const bool live = k < L;
            dvr[i] = live ? dv : 0.f; ur[i] = live ? uf : 0.f; gr[i] = live ? g : 0.f; sgr[i] = sg;
            s_dv[row][lane] = dvr[i];
            s_u[row][lane] = ur[i];
            s_g[row][lane] = gr[i];
        }
        s_bc[bc_step][0][bc_n] = to_float<IO>(pb);
        s_bc[bc_step][1][bc_n] = to_float<IO>(pc);
        const v2f h0A = {h0n[0], h0n[1]}, h0B = {h0n[2], h0n[3]};
        if (t > 0) {
            fetch_bwd(t - 1);
            if (t > 1) load_tabs(t - 2);
        }
        if (!(probe & 2)) __syncthreads();

        // ---- forward recompute of the 16 states -----------------------------------------------------------------
        v2f hsA[LT], hsB[LT];
        {
            // operands one step ahead in registers (2 waves per SIMD do not hide an LDS round trip per step)
            v2f hhA = h0A, hhB = h0B;
            float dvn = s_dv[0][lane], un = s_u[0][lane];
            v4f Bn = *reinterpret_cast<const v4f *>(&s_bc[0][0][n0]);
#pragma unroll
            for (int s = 0; s < ((probe & 8) ? 1 : LT); ++s) {
                const float dv = dvn, du = dv * un;
                const v4f Bv = Bn;
                if (s + 1 < LT) {
                    dvn = s_dv[s + 1][lane];
                    un = s_u[s + 1][lane];
                    Bn = *reinterpret_cast<const v4f *>(&s_bc[s + 1][0][n0]);
                }
                const v2f dtv = {dv, dv}, duv = {du, du};
                const v2f xA = a2A * dtv, xB = a2B * dtv;
                const v2f eA = {fast_exp2(xA.x), fast_exp2(xA.y)}, eB = {fast_exp2(xB.x), fast_exp2(xB.y)};
                hhA = __builtin_elementwise_fma(eA, hhA, v2f{Bv.x, Bv.y} * duv);
                hhB = __builtin_elementwise_fma(eB, hhB, v2f{Bv.z, Bv.w} * duv);
                hsA[s] = hhA;
                hsB[s] = hhB;
            }
        }
        // ---- reverse recurrence + gradients (two states per instruction) ----------------------------------------------
        float rdv = s_dv[LT - 1][lane], ru = s_u[LT - 1][lane], rg = s_g[LT - 1][lane];
        v4f rB = *reinterpret_cast<const v4f *>(&s_bc[LT - 1][0][n0]), rC = *reinterpret_cast<const v4f *>(&s_bc[LT - 1][1][n0]);
#pragma unroll
        for (int g = NG - 1; g >= 0; --g) {
#pragma unroll
            for (int si = 3; si >= 0; --si) {
                const int s = g * 4 + si;
                const float dv = rdv, uu = ru, gg = rg;
                const float du = dv * uu;
                const v4f Bv = rB, Cv = rC;
                if (s > 0) {        // the previous step's operands (the sweep runs backwards) are requested before this step's arithmetic
                    rdv = s_dv[s - 1][lane]; ru = s_u[s - 1][lane]; rg = s_g[s - 1][lane];
                    rB = *reinterpret_cast<const v4f *>(&s_bc[s - 1][0][n0]);
                    rC = *reinterpret_cast<const v4f *>(&s_bc[s - 1][1][n0]);
                }
                const v2f dtv = {dv, dv}, duv = {du, du}, ggv = {gg, gg};
                const v2f xA = a2A * dtv, xB = a2B * dtv;
                const v2f eA = {fast_exp2(xA.x), fast_exp2(xA.y)}, eB = {fast_exp2(xB.x), fast_exp2(xB.y)};
                const v2f dhA = __builtin_elementwise_fma(v2f{Cv.x, Cv.y}, ggv, adhA), dhB = __builtin_elementwise_fma(v2f{Cv.z, Cv.w}, ggv, adhB);
                const v2f hmA = s > 0 ? hsA[s > 0 ? s - 1 : 0] : h0A, hmB = s > 0 ? hsB[s > 0 ? s - 1 : 0] : h0B;
                const v2f t2A = dhA * (eA * hmA), t2B = dhB * (eB * hmB);         // dh * a_l * h_{l-1}
                dAA = __builtin_elementwise_fma(dtv, t2A, dAA);
                dAB = __builtin_elementwise_fma(dtv, t2B, dAB);
                const v2f sA2 = __builtin_elementwise_fma(AnB, t2B, AnA * t2A);
                const v2f sp2 = __builtin_elementwise_fma(dhB, v2f{Bv.z, Bv.w}, dhA * v2f{Bv.x, Bv.y});
                adhA = eA * dhA;
                adhB = eB * dhB;
                const v2f pBA = dhA * duv, pBB = dhB * duv, pCA = ggv * hsA[s], pCB = ggv * hsB[s];
                *reinterpret_cast<v2f *>(&s_part[wave][s][lane][0]) = v2f{sp2.x + sp2.y, sA2.x + sA2.y};
                *reinterpret_cast<v4f *>(red_b + si * kRedPitch + lane * 4) = v4f{pBA.x, pBA.y, pBB.x, pBB.y};
                *reinterpret_cast<v4f *>(red_c + si * kRedPitch + lane * 4) = v4f{pCA.x, pCA.y, pCB.x, pCB.y};
            }
            if constexpr ((probe & 1) != 0) continue;
            // cross-channel sums of the group: lane -> (half wave, dB | dC, step, state); 32 values each, then fold the halves
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            float rv[32];                                   // all reads in flight before the first add
#pragma unroll
            for (int q = 0; q < 32; ++q) rv[q] = red_src[q * 4];
#pragma unroll
            for (int w = 16; w > 0; w >>= 1) {
#pragma unroll
                for (int q = 0; q < w; ++q) rv[q] += rv[q + w];
            }
            float acc = rv[0];
            acc += __shfl_xor(acc, 32, 64);
            if (lane < 32) bc_out[(static_cast<int64_t>(t * LT + g * 4 + r_si) * 2 + r_which) * N + n0 + r_j] = acc;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();                       // next group overwrites the buffer
        }
        if (!(probe & 2)) __syncthreads();
        // ---- epilogue: own rows -----------------------------------------------------------------------------------
#pragma unroll
        for (int i = 0; i < RPT; ++i) {
            const int row = wave * RPT + i, k = t * LT + row;
            float SP = 0.f, SA = 0.f;
#pragma unroll
            for (int w = 0; w < NW; ++w) {
                const v2f v = *reinterpret_cast<const v2f *>(&s_part[w][row][lane][0]);
                SP += v.x;
                SA += v.y;
            }
            const float duo = __builtin_fmaf(dvr[i], SP, gr[i] * Dv);
            const float dd = __builtin_fmaf(ur[i], SP, SA) * sgr[i];
            if (k < L && !(probe & 4)) {
                buf_st<IO>(from_float<IO>(duo), du_rs, lane_off, k * du_ls);
                buf_st<IO>(from_float<IO>(dd), dd_rs, lane_off, k * dd_ls);
                db_acc += dd;
                dD_acc = __builtin_fmaf(gr[i], ur[i], dD_acc);
            }
        }
        // s_dv/s_u/s_g/s_part are rewritten only after the next tile's first barrier -> no third barrier needed:
        // the prologue of tile t-1 writes s_dv..s_g, which the core of tile t no longer reads (barrier above).
    }
    // ---- per-sample parameter gradients -> workspace (summed over the batch by the finishing kernel) -------------------
    float *pa = ws.pa + (static_cast<int64_t>(b) * p.dim + c) * (N + 2);
    pa[n0 + 0] = dAA.x;
    pa[n0 + 1] = dAA.y;
    pa[n0 + 2] = dAB.x;
    pa[n0 + 3] = dAB.y;
    __syncthreads();
    s_part[wave][0][lane][0] = dD_acc;
    s_part[wave][0][lane][1] = db_acc;
    __syncthreads();
    if (wave == 0) {
        float sD = 0.f, sb = 0.f;
#pragma unroll
        for (int w = 0; w < NW; ++w) { sD += s_part[w][0][lane][0]; sb += s_part[w][0][lane][1]; }
        pa[N] = sD;
        pa[N + 1] = sb;
    }
}

// dB[b, n, l] = sum over slabs of the per-slab partials (same for dC); dA, dD, ddelta_bias = sums over the batch.
__global__ void scan_bwd_finish_bc(const zigma_scan_bwd_params_t p, const BwdWs ws, int n_slabs, int n_tiles) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;   // (b, l, which, n)
    const int N = p.dstate;
    const int64_t total = static_cast<int64_t>(p.batch) * p.seqlen * 2 * N;
    if (i >= total) return;
    const int n = static_cast<int>(i % N), which = static_cast<int>((i / N) % 2);
    const int l = static_cast<int>((i / (2 * N)) % p.seqlen), b = static_cast<int>(i / (static_cast<int64_t>(2) * N * p.seqlen));
    float acc = 0.f;
    for (int s = 0; s < n_slabs; ++s)
        acc += ws.bc[(((static_cast<int64_t>(b) * n_slabs + s) * n_tiles * kBT + l) * 2 + which) * N + n];
    if (which == 0) p.dB[b * p.dB_batch_stride + n * p.dB_dstate_stride + l * p.dB_l_stride] = acc;
    else p.dC[b * p.dC_batch_stride + n * p.dC_dstate_stride + l * p.dC_l_stride] = acc;
}
__global__ void scan_bwd_finish_params(const zigma_scan_bwd_params_t p, const BwdWs ws) {
    const int64_t i = static_cast<int64_t>(blockIdx.x) * blockDim.x + threadIdx.x;   // (d, n')  n' in [0, N+2)
    const int N = p.dstate, W = N + 2;
    if (i >= static_cast<int64_t>(p.dim) * W) return;
    const int d = static_cast<int>(i / W), n = static_cast<int>(i % W);
    float acc = 0.f;
    for (int b = 0; b < p.batch; ++b) acc += ws.pa[(static_cast<int64_t>(b) * p.dim + d) * W + n];
    if (n < N) p.dA[d * N + n] = acc;
    else if (n == N) { if (p.dD) p.dD[d] = acc; }
    else if (p.ddelta_bias) p.ddelta_bias[d] = acc;
}

static void bwd_ws_layout(const zigma_scan_bwd_params_t &p, int64_t &ck, int64_t &bc, int64_t &pa) {
    const int64_t n_tiles = (p.seqlen + kBT - 1) / kBT, slabs = p.dim / 64;
    ck = p.checkpoints ? 0 : static_cast<int64_t>(p.batch) * slabs * n_tiles * p.dstate * 64;
    bc = static_cast<int64_t>(p.batch) * slabs * n_tiles * kBT * 2 * p.dstate;
    pa = static_cast<int64_t>(p.batch) * p.dim * (p.dstate + 2);
}

template <typename IO>
static void launch_bwd(const zigma_scan_bwd_params_t &p, const BwdWs &ws, hipStream_t stream) {
    dim3 grid(p.dim / 64, p.batch);
    if (p.dstate == 16) hipLaunchKernelGGL((scan_bwd_kernel<IO, 4>), grid, dim3(256), 0, stream, p, ws);
    else hipLaunchKernelGGL((scan_bwd_kernel<IO, 2>), grid, dim3(128), 0, stream, p, ws);
}

}  // namespace zigma

using namespace zigma;

extern "C" int64_t zigma_selective_scan_bwd_workspace_bytes(const zigma_scan_bwd_params_t *p) {
    if (!p || p->batch <= 0 || p->dim <= 0 || p->seqlen <= 0 || p->dstate <= 0) return 0;
    int64_t ck, bc, pa;
    bwd_ws_layout(*p, ck, bc, pa);
    return (ck + bc + pa) * static_cast<int64_t>(sizeof(float));
}

extern "C" int zigma_selective_scan_bwd(const zigma_scan_bwd_params_t *pp, void *stream_) {
    if (!pp) return ZIGMA_ERR_NULL;
    (void)hipGetLastError();
    const zigma_scan_bwd_params_t &p = *pp;
    if (p.batch < 0 || p.dim < 0 || p.seqlen < 0) return ZIGMA_ERR_SHAPE;
    if (p.flags != 0) return ZIGMA_ERR_UNSUPPORTED;
    if (p.dim % 64 != 0 || (p.dstate != 16 && p.dstate != 8) || p.batch > 65535) return ZIGMA_ERR_SHAPE;
    if (p.reset_period < 0 || p.reset_period % kBT != 0) return ZIGMA_ERR_SHAPE;
    if (p.batch == 0 || p.dim == 0 || p.seqlen == 0) return ZIGMA_OK;   // nothing to write (parameter gradients: caller zero-fills)
    if (!p.u || !p.delta || !p.A || !p.B || !p.C || !p.dout || !p.du || !p.ddelta || !p.dA || !p.dB || !p.dC) return ZIGMA_ERR_NULL;
    if (p.z && (!p.out || !p.dz)) return ZIGMA_ERR_NULL;
    if ((p.D && !p.dD) || (p.delta_bias && !p.ddelta_bias)) return ZIGMA_ERR_NULL;
    if (!p.workspace || p.workspace_bytes < zigma_selective_scan_bwd_workspace_bytes(pp)) return ZIGMA_ERR_NULL;
    if (reinterpret_cast<uintptr_t>(p.workspace) % 16 != 0) return ZIGMA_ERR_STRIDE;
    const int es = p.io_dtype == ZIGMA_F32 ? 4 : 2;
    const int64_t lim = (int64_t(1) << 31) - 1;
    const int64_t strides[] = {p.u_l_stride, p.delta_l_stride, p.z_l_stride, p.out_l_stride, p.dout_l_stride, p.du_l_stride,
                               p.ddelta_l_stride, p.dz_l_stride, p.B_l_stride, p.C_l_stride};
    for (int64_t s : strides)
        if (s < 0 || (p.seqlen - 1) * s * es + static_cast<int64_t>(p.dim) * es > lim) return ZIGMA_ERR_STRIDE;
    int64_t ck, bc, pa;
    bwd_ws_layout(p, ck, bc, pa);
    BwdWs ws;
    ws.ck = reinterpret_cast<float *>(p.workspace);
    ws.bc = ws.ck + ck;
    ws.pa = ws.bc + bc;
    hipStream_t stream = static_cast<hipStream_t>(stream_);
    ZIGMA_DISPATCH_DTYPE(p.io_dtype, IO, { launch_bwd<IO>(p, ws, stream); })
    const int n_tiles = (p.seqlen + kBT - 1) / kBT, n_slabs = p.dim / 64;
    const int64_t nbc = static_cast<int64_t>(p.batch) * p.seqlen * 2 * p.dstate;
    hipLaunchKernelGGL(scan_bwd_finish_bc, dim3(static_cast<unsigned>((nbc + 255) / 256)), dim3(256), 0, stream, p, ws, n_slabs, n_tiles);
    const int64_t npa = static_cast<int64_t>(p.dim) * (p.dstate + 2);
    hipLaunchKernelGGL(scan_bwd_finish_params, dim3(static_cast<unsigned>((npa + 255) / 256)), dim3(256), 0, stream, p, ws);
    set_last_kernel("scan_bwd_tok");
    return check_launch();
}
