"""Generator + runner of the second VALU micro-benchmark (settle the scan kernel's VALU floor).

Every mode is a loop whose body is a fixed inline-asm instruction sequence (hipcc cannot reorder or fold it); time is
taken INSIDE the kernel with s_memtime (shader cycles, MI355X_MICROARCH.md) next to s_memrealtime (100 MHz), so the
figures are cycles per wave-instruction per SIMD independent of DVFS, and the effective shader clock of each run is
reported beside them (the round-1 tool divided wall time by an assumed 2.4 GHz).

    python tools/ubench2/gen.py            # writes tools/ubench2/ubench2.hip, builds libubench2.so, runs if a GPU is there
"""
import ctypes
import json
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
DPP = "row_newbcast:{k} row_mask:0xf bank_mask:0xf bound_ctrl:1"

MODES = []      # (name, n_instr_per_iter, body_c_code)


def asm(lines, outs, ins, volatile=True):
    """lines use {o0}.. for outputs, {i0}.. for inputs"""
    nm = {}
    for j in range(len(outs)):
        nm[f"o{j}"] = f"%{j}"
    for j in range(len(ins)):
        nm[f"i{j}"] = f"%{len(outs) + j}"
    text = "\\n\\t".join(l.format(**nm) for l in lines)
    o = ", ".join(f'"{c}"({v})' for c, v in outs)
    i = ", ".join(f'"{c}"({v})' for c, v in ins)
    return f'asm volatile("{text}" : {o} : {i});'


def simple16(name, fmt, ins=(("v", "c0"), ("v", "c1")), n=16):
    outs = [("+v", f"d[{j}]") for j in range(n)]
    lines = [fmt.replace("{d}", "{o%d}" % j).replace("{k}", str(j % 16)) for j in range(n)]
    MODES.append((name, n, asm(lines, outs, list(ins))))


simple16("v_fma_f32 (VOP3, d = d*c0 + c1)", "v_fma_f32 {d}, {d}, {i0}, {i1}")
simple16("v_fmac_f32 (d += c0*c1)", "v_fmac_f32_e32 {d}, {i0}, {i1}")
simple16("v_mul_f32 (d = c0*d)", "v_mul_f32_e32 {d}, {i0}, {d}")
simple16("v_exp_f32 (d = exp2(c0))", "v_exp_f32_e32 {d}, {i0}")
simple16("v_mul_f32_dpp row_newbcast (d = bcast(c0)*c1)", "v_mul_f32_dpp {d}, {i0}, {i1} " + DPP)
simple16("v_fmac_f32_dpp row_newbcast (d += bcast(c0)*c1)", "v_fmac_f32_dpp {d}, {i0}, {i1} " + DPP)
simple16("v_mul_f32 SGPR src (d = s*c1)", "v_mul_f32_e32 {d}, {i0}, {i1}", ins=(("s", "s0"), ("v", "c1")))
simple16("v_fmac_f32 SGPR src (d += s*c1)", "v_fmac_f32_e32 {d}, {i0}, {i1}", ins=(("s", "s0"), ("v", "c1")))

# packed
outs = [("+v", f"p[{j}]") for j in range(8)]
MODES.append(("v_pk_mul_f32 (8 per iter)", 8, asm([f"v_pk_mul_f32 {{o{j}}}, {{o{j}}}, {{i0}}" for j in range(8)], outs, [("v", "pc")])))
MODES.append(("v_pk_fma_f32 (8 per iter)", 8, asm([f"v_pk_fma_f32 {{o{j}}}, {{o{j}}}, {{i0}}, {{i1}}" for j in range(8)], outs, [("v", "pc"), ("v", "pc2")])))

# 4 exp + 16 fma interleaved (does the transcendental overlap the plain pipe inside one wave?)
lines = []
for j in range(16):
    if j % 4 == 0:
        lines.append(f"v_exp_f32_e32 {{o{16 + j // 4}}}, {{i0}}")
    lines.append(f"v_fma_f32 {{o{j}}}, {{o{j}}}, {{i0}}, {{i1}}")
MODES.append(("mix: 4 v_exp + 16 v_fma interleaved 1:4", 20,
              asm(lines, [("+v", f"d[{j}]") for j in range(16)] + [("+v", f"e[{j}]") for j in range(4)], [("v", "c0"), ("v", "c1")])))
# 8 exp then 16 fma grouped
lines = [f"v_exp_f32_e32 {{o{16 + j}}}, {{i0}}" for j in range(8)] + [f"v_fma_f32 {{o{j}}}, {{o{j}}}, {{i0}}, {{i1}}" for j in range(16)]
MODES.append(("mix: 8 v_exp then 16 v_fma (grouped)", 24,
              asm(lines, [("+v", f"d[{j}]") for j in range(16)] + [("+v", f"e[{j}]") for j in range(8)], [("v", "c0"), ("v", "c1")])))

# dependent mul -> exp pairs (the compiler's order in the scan core)
lines = []
for j in range(8):
    lines += [f"v_mul_f32_e32 {{o{j}}}, {{i0}}, {{i1}}", f"v_exp_f32_e32 {{o{j}}}, {{o{j}}}"]
MODES.append(("8 x (v_mul -> dependent v_exp)", 16, asm(lines, [("=&v", f"d[{j}]") for j in range(8)], [("v", "c0"), ("v", "c1")])))


def core_step(S, form):
    """one recurrence step of 4 states (20 instructions).  outs: h0..h3, y, t0..t3, q0..q3; ins: dv, du, a0..a3, B, C."""
    # operand numbering: o0-3 h, o4 y, o5-8 t (exp), o9-12 q (B*du); i0 dv, i1 du, i2-5 a2, i6 B, i7 C  (or sgpr: i6-9 B0-3, i10-13 C0-3)
    L = []
    if form in ("dpp", "dpp_grouped"):
        if form == "dpp":      # compiler-like: mul/exp pairs, then B*du / fma pairs, then the y chain
            for j in range(4):
                L += [f"v_mul_f32_e32 {{o{5 + j}}}, {{i0}}, {{i{2 + j}}}", f"v_exp_f32_e32 {{o{5 + j}}}, {{o{5 + j}}}"]
            for j in range(4):
                L += [f"v_mul_f32_dpp {{o{9 + j}}}, {{i6}}, {{i1}} " + DPP.format(k=S * 4 + j), f"v_fmac_f32_e32 {{o{9 + j}}}, {{o{5 + j}}}, {{o{j}}}",
                      f"v_mov_b32_e32 {{o{j}}}, {{o{9 + j}}}"]
            L = [l for l in L if not l.startswith("v_mov")]        # (h lives in q after the fmac; swap roles instead of moving)
        else:                  # grouped inside the step: 4 mul, 4 exp, 4 mul_dpp, 4 fma
            L += [f"v_mul_f32_e32 {{o{5 + j}}}, {{i0}}, {{i{2 + j}}}" for j in range(4)]
            L += [f"v_exp_f32_e32 {{o{5 + j}}}, {{o{5 + j}}}" for j in range(4)]
            L += [f"v_mul_f32_dpp {{o{9 + j}}}, {{i6}}, {{i1}} " + DPP.format(k=S * 4 + j) for j in range(4)]
            L += [f"v_fmac_f32_e32 {{o{9 + j}}}, {{o{5 + j}}}, {{o{j}}}" for j in range(4)]
        L += [f"v_mul_f32_dpp {{o4}}, {{i7}}, {{o9}} " + DPP.format(k=S * 4)]
        L += [f"v_fmac_f32_dpp {{o4}}, {{i7}}, {{o{9 + j}}} " + DPP.format(k=S * 4 + j) for j in range(1, 4)]
        L += [f"v_mov_b32_e32 {{o{j}}}, {{o{9 + j}}}" for j in range(0)]   # no moves: next step reads h from q (see body)
    return L


def core_body(form, steps=4):
    """`steps` recurrence steps; h ping-pongs between two register sets so no moves are needed"""
    code = []
    for S in range(steps):
        hin, hout = ("ha", "hb") if S % 2 == 0 else ("hb", "ha")
        if form in ("dpp", "dpp_grouped"):
            outs = [("+v", f"{hin}[{j}]") for j in range(4)] + [("=&v", f"y[{S}]")] + [("=&v", f"t[{j}]") for j in range(4)] + \
                   [("=&v", f"{hout}[{j}]") for j in range(4)]
            ins = [("v", f"dv[{S}]"), ("v", f"du[{S}]")] + [("v", f"a2[{j}]") for j in range(4)] + [("v", "Bf"), ("v", "Cf")]
            code.append(asm(core_step(S, form), outs, ins))
        elif form in ("sgpr", "sgpr_grouped"):
            outs = [("+v", f"{hin}[{j}]") for j in range(4)] + [("=&v", f"y[{S}]")] + [("=&v", f"t[{j}]") for j in range(4)] + \
                   [("=&v", f"{hout}[{j}]") for j in range(4)]
            ins = [("v", f"dv[{S}]"), ("v", f"du[{S}]")] + [("v", f"a2[{j}]") for j in range(4)] + \
                  [("s", f"sB[{S * 4 + j}]") for j in range(4)] + [("s", f"sC[{S * 4 + j}]") for j in range(4)]
            L = []
            if form == "sgpr":
                for j in range(4):
                    L += [f"v_mul_f32_e32 {{o{5 + j}}}, {{i0}}, {{i{2 + j}}}", f"v_exp_f32_e32 {{o{5 + j}}}, {{o{5 + j}}}"]
                for j in range(4):
                    L += [f"v_mul_f32_e32 {{o{9 + j}}}, {{i{6 + j}}}, {{i1}}", f"v_fmac_f32_e32 {{o{9 + j}}}, {{o{5 + j}}}, {{o{j}}}"]
            else:
                L += [f"v_mul_f32_e32 {{o{5 + j}}}, {{i0}}, {{i{2 + j}}}" for j in range(4)]
                L += [f"v_exp_f32_e32 {{o{5 + j}}}, {{o{5 + j}}}" for j in range(4)]
                L += [f"v_mul_f32_e32 {{o{9 + j}}}, {{i{6 + j}}}, {{i1}}" for j in range(4)]
                L += [f"v_fmac_f32_e32 {{o{9 + j}}}, {{o{5 + j}}}, {{o{j}}}" for j in range(4)]
            L += [f"v_mul_f32_e32 {{o4}}, {{i10}}, {{o9}}"]
            L += [f"v_fmac_f32_e32 {{o4}}, {{i{10 + j}}}, {{o{9 + j}}}" for j in range(1, 4)]
            code.append(asm(L, outs, ins))
    return "\n        ".join(code)


MODES.append(("scan core, DPP operands, compiler order (4 steps x 4 states = 80 instr)", 80, core_body("dpp")))
MODES.append(("scan core, DPP operands, grouped per step (80 instr)", 80, core_body("dpp_grouped")))
MODES.append(("scan core, SGPR operands, compiler order (80 instr)", 80, core_body("sgpr")))
MODES.append(("scan core, SGPR operands, grouped per step (80 instr)", 80, core_body("sgpr_grouped")))


def core_wide(form):
    """4 steps: phase-grouped ACROSS the steps: 16 mul, 16 exp, 16 B*du, then the 4 h chains (16 fma), then 16 C ops."""
    code = []
    bsrc = (lambda S, j: ("v", "Bf")) if form == "dpp" else (lambda S, j: ("s", f"sB[{S * 4 + j}]"))
    # 1. 16 mul  e[S*4+j] = dv[S] * a2[j]
    outs = [("=&v", f"e16[{k}]") for k in range(16)]
    ins = [("v", f"dv[{S}]") for S in range(4)] + [("v", f"a2[{j}]") for j in range(4)]
    code.append(asm([f"v_mul_f32_e32 {{o{S * 4 + j}}}, {{i{S}}}, {{i{4 + j}}}" for S in range(4) for j in range(4)], outs, ins))
    # 2. 16 exp
    code.append(asm([f"v_exp_f32_e32 {{o{k}}}, {{o{k}}}" for k in range(16)], [("+v", f"e16[{k}]") for k in range(16)], []))
    # 3. 16 B*du
    outs = [("=&v", f"q16[{k}]") for k in range(16)]
    if form == "dpp":
        ins = [("v", f"du[{S}]") for S in range(4)] + [("v", "Bf")]
        code.append(asm([f"v_mul_f32_dpp {{o{S * 4 + j}}}, {{i4}}, {{i{S}}} " + DPP.format(k=S * 4 + j) for S in range(4) for j in range(4)], outs, ins))
    else:
        ins = [("v", f"du[{S}]") for S in range(4)] + [("s", f"sB[{k}]") for k in range(16)]
        code.append(asm([f"v_mul_f32_e32 {{o{S * 4 + j}}}, {{i{4 + S * 4 + j}}}, {{i{S}}}" for S in range(4) for j in range(4)], outs, ins))
    # 4. chains: q16[S*4+j] = e16[S*4+j] * h_prev + q16[S*4+j]   (h_prev = ha[j] for S = 0, else q16[(S-1)*4+j])
    outs = [("+v", f"q16[{k}]") for k in range(16)]
    ins = [("v", f"e16[{k}]") for k in range(16)] if False else []
    L = []
    # (operand budget: 16 q + 4 ha + up to 10 e at a time -> two statements)
    for half in range(2):
        ks = range(half * 8, half * 8 + 8)
        o = [("+v", f"q16[{k}]") for k in ks] + ([] if half == 0 else [])
        i = [("v", f"e16[{k}]") for k in ks] + ([("v", f"ha[{j}]") for j in range(4)] if half == 0 else [("v", f"q16[{4 + j}]") for j in range(4)])
        L = []
        for n_, k in enumerate(ks):
            S, j = divmod(k, 4)
            if S % 2 == 0:      # first step of this half: previous h is an input operand
                L.append(f"v_fmac_f32_e32 {{o{n_}}}, {{i{n_}}}, {{i{8 + j}}}")
            else:
                L.append(f"v_fmac_f32_e32 {{o{n_}}}, {{i{n_}}}, {{o{n_ - 4}}}")
        code.append(asm(L, o, i))
    # 5. y[S] = sum_j C[S,j] * h[S,j]
    outs = [("=&v", f"y[{S}]") for S in range(4)]
    if form == "dpp":
        ins = [("v", f"q16[{k}]") for k in range(16)] + [("v", "Cf")]
        L = []
        for S in range(4):
            L.append(f"v_mul_f32_dpp {{o{S}}}, {{i16}}, {{i{S * 4}}} " + DPP.format(k=S * 4))
            L += [f"v_fmac_f32_dpp {{o{S}}}, {{i16}}, {{i{S * 4 + j}}} " + DPP.format(k=S * 4 + j) for j in range(1, 4)]
        code.append(asm(L, outs, ins))
    else:
        for half in range(2):
            o = [("=&v", f"y[{S}]") for S in range(half * 2, half * 2 + 2)]
            i = [("v", f"q16[{k}]") for k in range(half * 8, half * 8 + 8)] + [("s", f"sC[{k}]") for k in range(half * 8, half * 8 + 8)]
            L = []
            for s_ in range(2):
                L.append(f"v_mul_f32_e32 {{o{s_}}}, {{i{8 + s_ * 4}}}, {{i{s_ * 4}}}")
                L += [f"v_fmac_f32_e32 {{o{s_}}}, {{i{8 + s_ * 4 + j}}}, {{i{s_ * 4 + j}}}" for j in range(1, 4)]
            code.append(asm(L, o, i))
    code.append("ha[0] = q16[12]; ha[1] = q16[13]; ha[2] = q16[14]; ha[3] = q16[15];")
    return "\n        ".join(code)


MODES.append(("scan core, DPP operands, phase-grouped across 4 steps (80 instr)", 80, core_wide("dpp")))
MODES.append(("scan core, SGPR operands, phase-grouped across 4 steps (80 instr)", 80, core_wide("sgpr")))

SRC_HEAD = r'''// GENERATED by tools/ubench2/gen.py — do not edit.  VALU issue-rate micro-benchmarks for gfx950 (not product code).
#include <hip/hip_runtime.h>
#include <stdint.h>
typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, int iters, float seed) {
    float d[16], e[8], ha[4], hb[4], y[4], t[4], dv[4], du[4], a2[4], e16[16], q16[16];
    v2f p[8];
    const float lanef = 1e-3f * threadIdx.x;
#pragma unroll
    for (int i = 0; i < 16; ++i) { d[i] = seed + 0.01f * i + lanef; e16[i] = 0.f; q16[i] = 0.f; }
#pragma unroll
    for (int i = 0; i < 8; ++i) { e[i] = 0.f; p[i] = v2f{d[i], d[i] + 0.5f}; }
#pragma unroll
    for (int i = 0; i < 4; ++i) { ha[i] = d[i]; hb[i] = 0.f; y[i] = 0.f; t[i] = 0.f; dv[i] = 0.01f + lanef; du[i] = 0.3f + lanef; a2[i] = -1.f - i; }
    float c0 = 0.999f + 1e-6f * threadIdx.x, c1 = 1e-3f, Bf = 0.5f + lanef, Cf = 0.25f + lanef;
    const v2f pc = {0.999f, 0.998f}, pc2 = {0.001f, 0.002f};
    float s0 = seed * 0.5f;
    float sB[16], sC[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { sB[i] = __builtin_amdgcn_readfirstlane(seed + 0.1f * i); sC[i] = __builtin_amdgcn_readfirstlane(seed - 0.05f * i); }
    s0 = __builtin_amdgcn_readfirstlane(s0);
    asm volatile("" : "+v"(c0), "+v"(c1), "+v"(Bf), "+v"(Cf));
    __syncthreads();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
'''
SRC_TAIL = r'''
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    const unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += d[i] + e16[i] + q16[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += e[i] + p[i].x + p[i].y;
#pragma unroll
    for (int i = 0; i < 4; ++i) s += ha[i] + hb[i] + y[i] + t[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * 4 + (threadIdx.x >> 6);
        cyc[4 * w] = t0;
        cyc[4 * w + 1] = t1;
        cyc[4 * w + 2] = r0;
        cyc[4 * w + 3] = r1;
    }
}
'''


def write_source(path):
    with open(path, "w") as fh:
        fh.write(SRC_HEAD)
        for i, (name, n, body) in enumerate(MODES):
            fh.write(f"        if constexpr (MODE == {i}) {{   // {name}\n        {body}\n        }}\n")
        fh.write(SRC_TAIL)
        fh.write('extern "C" int ubench2_launch(int mode, int blocks, int iters, float *out, unsigned long long *cyc, void *stream) {\n'
                 "    hipStream_t st = static_cast<hipStream_t>(stream);\n    dim3 g(blocks), b(256);\n    switch (mode) {\n")
        for i in range(len(MODES)):
            fh.write(f"        case {i}: hipLaunchKernelGGL(k<{i}>, g, b, 0, st, out, cyc, iters, 0.5f); break;\n")
        fh.write("        default: return -1;\n    }\n    return hipGetLastError() == hipSuccess ? 0 : -5;\n}\n")


def main():
    src, lib = os.path.join(HERE, "ubench2.hip"), os.path.join(HERE, "libubench2.so")
    write_source(src)
    if "--no-build" not in sys.argv:
        subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", src, "-o", lib], check=True)
    import torch
    if not torch.cuda.is_available():
        print(f"built {lib}: {len(MODES)} modes; no GPU here")
        return
    L = ctypes.CDLL(lib)
    L.ubench2_launch.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]
    out = torch.empty(256 * 8 * 256, device="cuda")
    cyc = torch.zeros(256 * 8 * 4 * 4, device="cuda", dtype=torch.int64)
    st = torch.cuda.current_stream().cuda_stream
    iters = 2000
    res = []
    for wps in (1, 2, 4, 5, 6, 8):
        blocks = 256 * wps
        for mode, (name, n, _) in enumerate(MODES):
            for rep in range(2):
                cyc.zero_()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                assert L.ubench2_launch(mode, blocks, iters, out.data_ptr(), cyc.data_ptr(), st) == 0
                e1.record()
                torch.cuda.synchronize()
            c = cyc[: blocks * 4 * 4].view(-1, 4).double()
            shader = (c[:, 1] - c[:, 0]).mean().item()              # s_memtime ticks per wave
            real = (c[:, 3] - c[:, 2]).mean().item()                # s_memrealtime ticks per wave
            span = (c[:, 3].max() - c[:, 2].min()).item()           # first wave start -> last wave end, realtime ticks
            wall_us = e0.elapsed_time(e1) * 1e3                      # HIP events around the launch
            conc = (c[:, 3] - c[:, 2]).sum().item() / span / 1024    # average resident waves per SIMD while the kernel ran
            rt_mhz = span / wall_us                                   # realtime counter rate implied by the event time
            per_instr = shader / (iters * n * conc)
            mhz = shader / real * 100.0
            wall_per_instr_ns = wall_us * 1e3 / (iters * n * wps)    # wall ns per wave-instruction per SIMD (all waves resident)
            res.append(dict(waves_per_simd=wps, mode=mode, name=name, instr_per_iter=n, memtime_ticks_per_iter_per_wave=shader / iters,
                            ticks_per_instr_per_simd=per_instr, memtime_mhz_vs_realtime=mhz, resident_waves_per_simd=conc,
                            event_wall_us=wall_us, realtime_span_ticks=span, realtime_mhz_vs_events=rt_mhz,
                            wall_ns_per_instr_per_simd=wall_per_instr_ns))
            print(f"w/SIMD={wps} {name:78s} {shader / iters:8.1f} tick/iter/wave  {per_instr:6.2f} tick/instr/SIMD  memtime {mhz:5.0f} MHz  "
                  f"resident {conc:4.2f}  wall {wall_us:7.1f} us = {wall_per_instr_ns:5.2f} ns/instr/SIMD  (realtime ctr {rt_mhz:5.1f} MHz)", flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(res, open("gpurun_out/ubench2.json", "w"), indent=1)


if __name__ == "__main__":
    main()
