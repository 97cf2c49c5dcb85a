#!/usr/bin/env python
"""Headline benchmark: denoiser-forward throughput of ZigMa on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

metric  (BASELINE.json): denoiser-forward latents/sec = B*L tokens per second of ZigMa.forward,
        ZigMa E=640, depth=18, L=32x32, zigzagN8 — workload = BASELINE configs[1]: the README model
        (in_ch=3, img 32, has_text with 77x768 context) in bf16 at B=64 per GPU, synthetic x / t / y.
step    = one ZigMa.forward over the per-rank batch (inputs resident in HBM before the timed region).
N > 1   = batch-sharded replicas (weak scaling: B=64 per rank); the path has no collective inside the
          forward — ranks only meet at the timing barriers and at the final all_gather of the velocities
          (the analogue of accelerator.gather in sample_acc.py:435), which is inside the timed region.
roofline  : the fused zigzag selective-scan kernel, timed with HIP events around every launch of it
            inside the timed steps; achieved = algorithmic bytes / mean launch time.
            `frac` is the HBM-roofline fraction the metric asks for; the kernel itself is VALU-issue limited
            (DESIGN.md §3.1), so the line also carries `valu_frac` = the recurrence's VALU floor at the guide's issue
            rates (4 plain ops x 2 cycles + 1 v_exp_f32 x 8 cycles per (element, state) wave-instruction group)
            divided by the measured launch time.  `traffic`: HBM bytes per launch from two rocprofv3 PMC passes (FETCH_SIZE doubled,
            WRITE_SIZE — MI355X_MICROARCH.md) of the same kernel launch as the model makes it, run as child processes AFTER the
            timed region when `rocprofv3` is on the box (`traffic_source: "measured ..."`); otherwise the byte count of the
            committed passes, labelled with the file it came from.
cpu_baseline : on the host cores, bounded sample (forwards of the same model at B=2), rank 0, N=1 only.  Where
            /root/reference is mounted (the build container) the reference's OWN CPU path — its unmodified `ZigMa.forward`
            over `selective_scan_ref` / `causal_conv1d_ref` through oracle/ref_shim.py, `kind: "reference"`; on the GPU box
            (a Python reference cannot travel) the torch-CPU restatement of the SAME ATen formulation (oracle/torch_port.py:
            materialised (B, D, L, N) einsum tensors, per-step loop, torch's CPU thread pool; pinned to the reference's golden
            outputs), `kind: "port-torch"`, `threads` = torch threads used.  (The numpy oracle, `kind: "port"`, is the last resort.)
check       : one forward OUTSIDE the timed region compared with the same model run through the unfused composition
            (library projections, no epilogue fusion, the two-kernel conv / x_proj): finite + norm-wise distance.
N > 1 without a launcher: `python bench.py --gpus N` re-executes itself under torch.distributed.run with N ranks.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12        # B/s, MI355X HBM3E spec (MI355X_MICROARCH.md)
SCLK = 2.4e9             # Hz, max shader clock (MI355X_MICROARCH.md)

WORKLOADS = {
    # BASELINE configs[1]: README model, bf16, B=64 on one MI355X
    "readme_text_b64": dict(model=dict(in_channels=3, img_dim=32, embed_dim=640, depth=18, patch_size=1, has_text=True,
                                       d_context=768, n_context_token=77, scan_type="zigzagN8", use_pe=2),
                            batch=64, x=(3, 32, 32), y=("text", 77, 768)),
}


class ScanTimer:
    """HIP-event pairs around scan launches (events are recorded on the stream the kernel goes to).  `every` = bracket every n-th launch:
    an event pair costs the stream ~6 us of idle time on either side of the kernel it brackets (kernel trace of round 5,
    profiles/r05_e_bench_kernel_trace_gaps.txt: 5.7 + 6.0 us around every scan against 0 between all other kernels), i.e. 216 us = 1.2 % of
    the forward with all 18 launches bracketed.  The default 7 is coprime with the 18 launches of a forward: over the timed steps every layer
    is sampled (51 pairs in 20 steps), the step carries 2-3 pairs instead of 18."""

    def __init__(self, every=1):
        self.pairs = []
        self.enabled = False
        self.every = max(1, int(every))
        self.count = 0

    def install(self):
        from zigma_amd import selective_scan_interface as ssi
        raw = ssi.scan_raw
        timer = self

        def timed(*a, **k):
            if not timer.enabled:
                return raw(*a, **k)
            timer.count += 1
            if timer.count % timer.every:
                return raw(*a, **k)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = raw(*a, **k)
            e1.record()
            timer.pairs.append((e0, e1))
            return r
        ssi.scan_raw = timed

    def mean_ms(self):
        return sum(a.elapsed_time(b) for a, b in self.pairs) / max(len(self.pairs), 1)


def measured_valu_rates():
    """(plain, packed, exp) ns per wave-instruction per SIMD from profiles/valu_rates_gfx950.json — the ONE place the measured issue rates live (the file
    cites the micro-benchmark they come from); the literal fall-back is that file's content, for a checkout without profiles/"""
    try:
        r = json.load(open(os.path.join(ROOT, "profiles", "valu_rates_gfx950.json")))["ns_per_wave_instruction_per_simd"]
        return float(r["plain_vop2"]), float(r["packed_f32_two_results"]), float(r["v_exp_f32"])
    except (OSError, KeyError, ValueError):
        return 1.35, 2.28, 3.43


def pmc_traffic_live(limit_s=150):
    """HBM bytes per scan launch measured NOW: two rocprofv3 --pmc passes (FETCH_SIZE; WRITE_SIZE — they do not fit one pass) over
    tools/scan_one.py (the scan exactly as the model launches it at the headline shape), FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950.  Child processes with a hard time limit; None when rocprofv3 is missing or fails."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3")
    if exe is None or os.environ.get("ZIGMA_BENCH_PMC", "1") == "0":
        return None, None
    vals = {}
    t_end = time.time() + limit_s
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="zigma_pmc_")
        try:
            subprocess.run([exe, "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "pmc", "--", sys.executable,
                            os.path.join(ROOT, "tools", "scan_one.py")], capture_output=True, text=True,
                           timeout=max(10, t_end - time.time()), cwd=d, env=dict(os.environ, TMPDIR=d, N="8"))
            got = []
            for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(f)):
                    if "scan_tok" in r["Kernel_Name"] and r["Counter_Name"] == counter:
                        got.append(float(r["Counter_Value"]))
            if not got:
                return None, None
            vals[counter] = sum(got) / len(got)
        except Exception:
            return None, None
        finally:
            shutil.rmtree(d, ignore_errors=True)
    return int(2 * vals["FETCH_SIZE"] * 1024 + vals["WRITE_SIZE"] * 1024), "measured: rocprofv3 --pmc FETCH_SIZE (x2) / WRITE_SIZE over tools/scan_one.py"


def pmc_traffic():
    """HBM bytes per scan launch from the committed rocprofv3 PMC passes (profiles/*pmc_scan*.json written by
    tools/pmc_scan.sh; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950), or None."""
    import glob
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "*pmc_scan*.json")) if "bwd" not in f)
    if not files:
        return None, None
    try:
        d = json.load(open(files[-1]))
        return d.get("hbm_bytes_per_launch"), "profiles/" + os.path.basename(files[-1])
    except Exception:
        return None, None


def build_model(cfg, device, dtype, seed=0):
    from zigma_amd.model_zigma import ZigMa
    torch.manual_seed(seed)
    m = ZigMa(device=device, dtype=dtype, **cfg).eval()
    g = torch.Generator(device="cpu").manual_seed(seed + 1)
    with torch.no_grad():       # default init zeroes every adaLN gate; use O(1) gates so the mixers matter
        for blk in m.blocks:
            w, b = blk.adaLN_modulation[-1].weight, blk.adaLN_modulation[-1].bias
            w.copy_((torch.randn(w.shape, generator=g) * 0.02).to(w))
            b.copy_((torch.randn(b.shape, generator=g) * 0.5).to(b))
        if hasattr(m, "pos_embed"):
            m.pos_embed.copy_((torch.randn(m.pos_embed.shape, generator=g) * 0.02).to(m.pos_embed))
    return m


def make_inputs(wl, batch, device, seed):
    g = torch.Generator(device="cpu").manual_seed(seed)
    x = torch.randn((batch,) + wl["x"], generator=g).to(device)
    t = torch.rand(batch, generator=g).to(device)
    y = torch.rand((batch,) + wl["y"][1:], generator=g).to(device=device, dtype=torch.bfloat16)
    return x, t, y


def check_against_unfused(model, x, t, y, out):
    """One forward of the SAME model through the unfused composition — library projections (no own MFMA kernel, no gated-add
    epilogue), the two-kernel conv / x_proj, the torch compositions of the embedding / conditioning / final-layer operators — outside the timed region: the timed path's output must be finite and agree
    norm-wise (bf16 model: the two compositions round at different points, ~5e-3)."""
    import zigma_amd.routing as zr
    import zigma_amd.model_zigma as mz
    import zigma_amd.selective_scan_interface as ssi
    import zigma_amd.embed as ze
    saved = (zr.POLICY, mz.FUSE_OUT_PROJ_ADD, mz.FUSE_OUT_PROJ_ADD_NO_TEXT, ssi.USE_CONV_X_PROJ, ze.USE_EMBED_KERNELS)
    zr.POLICY, mz.FUSE_OUT_PROJ_ADD, mz.FUSE_OUT_PROJ_ADD_NO_TEXT, ssi.USE_CONV_X_PROJ, ze.USE_EMBED_KERNELS = "off", False, False, False, False
    try:
        with torch.no_grad():
            ref = model(x, t, y)
    finally:
        zr.POLICY, mz.FUSE_OUT_PROJ_ADD, mz.FUSE_OUT_PROJ_ADD_NO_TEXT, ssi.USE_CONV_X_PROJ, ze.USE_EMBED_KERNELS = saved
    finite = bool(torch.isfinite(out).all())
    err = float((out.double() - ref.double()).norm() / ref.double().norm())
    if not finite or not err < 3e-2:
        raise SystemExit(f"bench.py: the timed path's output fails its check (finite={finite}, rel err vs the unfused composition {err:.3e})")
    return dict(finite=finite, rel_err_vs_unfused=err, bound=3e-2, checksum=float(out.double().sum()))


def check_against_reference(model, wl, x, t, y):
    """The driver line's REFERENCE distance.  AFTER the timed region the same model object gets the weights of the committed
    reference run (tests/golden/r2_readme_b2.npz: the UNMODIFIED reference's ZigMa on CPU, fp32 and bf16, oracle/make_golden_r2.py; weights = the
    deterministic fill both sides regenerate from the fixture's seed), the reference run's two samples replace batch positions 0 and B-1 of the TIMED
    inputs, and one forward of the timed path — same batch size, same kernels, same routing — is compared with the reference's own outputs for those two
    rows.  Bars as in tests/test_gpu_baseline_shapes.py: no further from the reference's fp32 result than 1.1 x the reference's own bf16 run is, and
    within 1e-2 of the reference's bf16 result.  (The fill function is part of the fixture — tests/golden/param_fill.py, the weights in generator form — and
    runs here outside the timed region; nothing under oracle/ is imported by this check.)"""
    import ast
    import numpy as np
    import importlib.util
    spec = importlib.util.spec_from_file_location("zigma_golden_param_fill", os.path.join(ROOT, "tests", "golden", "param_fill.py"))
    pf = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(pf)
    fill_state = pf.fill_state
    path = os.path.join(ROOT, "tests", "golden", "r2_readme_b2.npz")
    if not os.path.exists(path) or x.shape[0] < 2:
        return None
    g = np.load(path)
    if ast.literal_eval(str(g["cfg"])) != wl["model"]:
        return None
    fill_state(model, int(g["seed"]))
    x2, t2, y2 = x.clone(), t.clone(), y.clone()
    for pos, src in ((0, 0), (x.shape[0] - 1, 1)):
        x2[pos] = torch.from_numpy(g["x"][src]).to(x2)
        t2[pos] = float(g["t"][src])
        y2[pos] = torch.from_numpy(g["y"][src]).to(y2)
    with torch.no_grad():
        out = model(x2.bfloat16(), t2.bfloat16(), y2)[[0, x.shape[0] - 1]].float().cpu().numpy()
    rel = lambda a, b: float(np.linalg.norm(a.astype(np.float64) - b) / np.linalg.norm(b))
    noise = float(g["ref_bf16_vs_fp32"])
    e32, e16 = rel(out, g["out"]), rel(out, g["out_bf16"])
    ok = bool(np.isfinite(out).all() and e32 < 1.1 * noise and e16 < 1e-2)
    if not ok:
        raise SystemExit(f"bench.py: the timed path fails its reference check (vs reference fp32 {e32:.3e}, bar {1.1 * noise:.3e}; vs reference bf16 {e16:.3e}, bar 1e-2)")
    return dict(rel_err_vs_reference_fp32=e32, bar_fp32=1.1 * noise, reference_own_bf16_vs_fp32=noise, rel_err_vs_reference_bf16=e16, bar_bf16=1e-2,
                fixture="tests/golden/r2_readme_b2.npz (the unmodified reference on CPU; samples at batch positions 0 and B-1 of the timed inputs)", ok=ok)


def box_calib(device):
    """Two FIXED kernels (csrc/calib.hip) timed in this process after the timed region: a 1 GiB copy (GB/s, read + write) and the issue time of
    v_fma_f32 / v_exp_f32 at the scan kernel's occupancy (ns per wave-instruction per SIMD).  Boxes of the pool differ by 3-4 %: with these
    two numbers in the line a change of the headline between rounds can be told from a change of silicon."""
    from zigma_amd import _lib
    n = 1 << 30
    src = torch.empty(n, device=device, dtype=torch.uint8).fill_(1)
    dst = torch.empty(n, device=device, dtype=torch.uint8)
    sink = torch.empty(1280 * 256, device=device, dtype=torch.float32)

    def timed(mode, iters, reps):
        P = _lib.CalibParams()
        P.mode, P.iters = mode, iters
        P.bytes, P.src, P.dst = (n, src.data_ptr(), dst.data_ptr()) if mode == 0 else (sink.numel() * 4, None, sink.data_ptr())
        _lib.call("zigma_calib_launch", P, device)          # warm-up
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps):
            _lib.call("zigma_calib_launch", P, device)
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) * 1e-3 / reps
    t_copy = timed(0, 0, 10)
    iters = 20000
    t_fma, t_exp = timed(1, iters, 5), timed(2, iters, 5)
    return dict(copy_1GiB_GBps=2 * n / t_copy / 1e9, v_fma_ns=t_fma / (iters * 16 * 5) * 1e9, v_exp_ns=t_exp / (iters * 8 * 5) * 1e9,
                note="csrc/calib.hip: 1 GiB copy (read + write bytes / time); ns per wave-instruction per SIMD at 5 waves per SIMD, 1024 SIMDs busy")


def _host_threads():
    try:
        return len(os.sched_getaffinity(0))
    except Exception:
        return os.cpu_count() or 1


def _cpu_model():
    try:
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name"):
                return ln.split(":", 1)[1].strip()
    except Exception:
        pass
    return "unknown"


def cpu_baseline_reference(wl, seed=0, budget_s=20.0):
    """The reference's own pure-CPU path on the host cores: the unmodified ZigMa.forward through
    oracle/ref_shim.py (selective_scan_ref, selective_scan_interface.py:86-152, is ~90 % of it), B=2, fp32, as many
    forwards as fit in `budget_s` (at least one); plus selective_scan_ref alone at the layer's shape."""
    import contextlib
    import io
    from oracle import ref_shim
    threads = min(_host_threads(), 32)            # torch's CPU kernels stop scaling (and start spinning) long before 128
    torch.set_num_threads(threads)
    with contextlib.redirect_stdout(io.StringIO()):           # the reference prints its tables while it builds
        mz, ssi, _, _ = ref_shim.reference_modules()
        torch.manual_seed(seed)
        m = mz.ZigMa(device="cpu", **wl["model"]).eval()
    g = torch.Generator().manual_seed(seed)
    with torch.no_grad():
        for blk in m.blocks:                                  # non-zero gates, as on the GPU side
            b = blk.adaLN_modulation[-1].bias
            b.copy_(torch.randn(b.shape, generator=g) * 0.5)
    B = 2
    L = (wl["model"]["img_dim"] // wl["model"]["patch_size"]) ** 2
    x = torch.randn((B,) + wl["x"], generator=g)
    t = torch.rand(B, generator=g)
    y = torch.rand((B,) + wl["y"][1:], generator=g)
    n, t0 = 0, time.perf_counter()
    with torch.no_grad():
        while n == 0 or (time.perf_counter() - t0) * (n + 1) / n < budget_s:
            m(x, t, y)
            n += 1
    dt = (time.perf_counter() - t0) / n
    # selective_scan_ref alone, the shape one layer sees at B=2 (u, delta, z: (B, Di, L); B, C: (B, N, L))
    Di, N = 2 * wl["model"]["embed_dim"], 16
    u, dl, z = (torch.randn(B, Di, L, generator=g) for _ in range(3))
    A = -torch.rand(Di, N, generator=g)
    Bm, Cm = torch.randn(B, N, L, generator=g), torch.randn(B, N, L, generator=g)
    D, db = torch.randn(Di, generator=g), torch.rand(Di, generator=g)
    with torch.no_grad():
        ssi.selective_scan_ref(u, dl, A, Bm, Cm, D, z, db, True)
        s0 = time.perf_counter()
        ssi.selective_scan_ref(u, dl, A, Bm, Cm, D, z, db, True)
        sdt = time.perf_counter() - s0
    scan_bytes = B * L * (4 * 4 * Di + 2 * 4 * N) + 4 * Di * (N + 2)          # fp32 I/O
    src = "/root/reference"
    return dict(value=B * L / dt, unit="tokens/s", cores=threads, kind="reference", cpu=_cpu_model(),
                sample=f"{n} forward(s) of the reference's own ZigMa (E=640, depth=18, has_text; selective_scan_ref + "
                       f"causal_conv1d_ref via oracle/ref_shim.py, from {src}) at B={B}, fp32, torch "
                       f"{torch.__version__.split('+')[0]} CPU, {threads} threads, {dt:.1f} s per forward",
                selective_scan_ref=dict(tokens_per_s=B * L / sdt, GBps=scan_bytes / sdt / 1e9, seconds=sdt,
                                        shape=f"B={B}, Di={Di}, L={L}, N={N}, fp32"))


def cpu_baseline_port_torch(wl, seed=0, budget_s=20.0):
    """The reference's CPU arithmetic where the reference itself is not mounted: oracle/torch_port.py (the same ATen ops and the
    same torch thread pool as selective_scan_ref / mamba_inner_ref / the model's forward), README model, B=2, fp32."""
    import numpy as np
    from oracle import torch_port as tp
    threads = min(_host_threads(), 32)
    torch.set_num_threads(threads)
    from zigma_amd.model_zigma import ZigMa
    torch.manual_seed(seed)
    m = ZigMa(device="cpu", dtype=torch.float32, **wl["model"])
    state = {k: v.detach().numpy() for k, v in m.state_dict().items()}
    rng = np.random.default_rng(seed)
    for k in state:                     # non-zero gates, as on the GPU side
        if "adaLN_modulation.1.bias" in k:
            state[k] = (rng.standard_normal(state[k].shape) * 0.5).astype(np.float32)
    pm = tp.ZigMaTorchPort(state, wl["model"])
    B = 2
    L = (wl["model"]["img_dim"] // wl["model"]["patch_size"]) ** 2
    x = rng.standard_normal((B,) + wl["x"]).astype(np.float32)
    t = rng.random(B).astype(np.float32)
    y = rng.random((B,) + wl["y"][1:]).astype(np.float32)
    n, t0 = 0, time.perf_counter()
    with torch.no_grad():
        while n == 0 or (time.perf_counter() - t0) * (n + 1) / n < budget_s:
            pm.forward(x, t, y)
            n += 1
    dt = (time.perf_counter() - t0) / n
    g = torch.Generator().manual_seed(seed)
    Di, N = 2 * wl["model"]["embed_dim"], 16
    u, dl, z = (torch.randn(B, Di, L, generator=g) for _ in range(3))
    A = -torch.rand(Di, N, generator=g)
    Bm, Cm = torch.randn(B, N, L, generator=g), torch.randn(B, N, L, generator=g)
    D, db = torch.randn(Di, generator=g), torch.rand(Di, generator=g)
    with torch.no_grad():
        tp.selective_scan_ref(u, dl, A, Bm, Cm, D, z, db, True)
        s0 = time.perf_counter()
        tp.selective_scan_ref(u, dl, A, Bm, Cm, D, z, db, True)
        sdt = time.perf_counter() - s0
    scan_bytes = B * L * (4 * 4 * Di + 2 * 4 * N) + 4 * Di * (N + 2)          # fp32 I/O
    return dict(value=B * L / dt, unit="tokens/s", cores=threads, threads=threads, kind="port-torch", cpu=_cpu_model(),
                sample=f"{n} forward(s) of the same model (E=640, depth=18, has_text) at B={B}, fp32, oracle/torch_port.py = the "
                       f"reference's ATen formulation (selective_scan_ref's einsum tensors + per-step loop), torch "
                       f"{torch.__version__.split('+')[0]} CPU, {threads} threads, {dt:.1f} s per forward",
                selective_scan_ref=dict(tokens_per_s=B * L / sdt, GBps=scan_bytes / sdt / 1e9, seconds=sdt,
                                        shape=f"B={B}, Di={Di}, L={L}, N={N}, fp32"))


def cpu_baseline_port(wl, seed=0):
    """Fallback: the numpy oracle's forward of the same architecture at B=2 (only when the reference is not importable)."""
    import numpy as np
    from oracle import zigma_oracle as zo
    threads = min(_host_threads(), 32)                   # BLAS threads of the numpy GEMMs; the recurrence itself is one thread
    try:
        from threadpoolctl import threadpool_limits
        threadpool_limits(threads)
    except Exception:
        pass
    from zigma_amd.model_zigma import ZigMa
    torch.manual_seed(seed)
    m = ZigMa(device="cpu", dtype=torch.float32, **wl["model"])
    state = {k: v.detach().numpy() for k, v in m.state_dict().items()}
    rng = np.random.default_rng(seed)
    for k in state:                     # non-zero gates, as on the GPU side
        if "adaLN_modulation.1.bias" in k:
            state[k] = (rng.standard_normal(state[k].shape) * 0.5).astype(np.float32)
    om = zo.ZigMaOracle(state, wl["model"])
    B = 2
    x = rng.standard_normal((B,) + wl["x"]).astype(np.float32)
    t = rng.random(B).astype(np.float32)
    y = rng.random((B,) + wl["y"][1:]).astype(np.float32)
    t0 = time.perf_counter()
    om.forward(x, t, y)
    dt = time.perf_counter() - t0
    L = (wl["model"]["img_dim"] // wl["model"]["patch_size"]) ** 2
    return dict(value=B * L / dt, unit="tokens/s", cores=1, threads_recurrence=1, threads_blas=threads, kind="port", cpu=_cpu_model(),
                sample=f"1 forward of the same model (E=640, depth=18, has_text) at B={B}, fp32 numpy oracle (recurrence on ONE "
                       f"thread, the GEMMs on {threads} BLAS threads), {dt:.1f} s")


def cpu_baseline(wl, name, limit_s=240):
    """Runs in a CHILD process with a hard time limit (a host with many cores and a small CPU quota can make the torch CPU
    path crawl; the GPU numbers of the line must not depend on it)."""
    import subprocess
    from oracle import ref_shim
    for kind in (["reference"] if ref_shim.available() else []) + ["port-torch", "port"]:
        try:
            r = subprocess.run([sys.executable, os.path.abspath(__file__), "--cpu-baseline-only", kind, "--workload", name],
                               capture_output=True, text=True, timeout=limit_s,
                               env=dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES=""))
            lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if r.returncode == 0 and lines:
                return json.loads(lines[-1])
            print(f"bench.py: {kind} CPU baseline failed (rc {r.returncode}): {r.stderr[-300:]}", file=sys.stderr)
        except subprocess.TimeoutExpired:
            print(f"bench.py: {kind} CPU baseline exceeded {limit_s} s on this host; trying the next kind", file=sys.stderr)
    return None


def respawn_under_launcher(n):
    """`python bench.py --gpus N` with no launcher around it: become `python -m torch.distributed.run ... bench.py ...`
    (one rank per GPU, rendezvous on 127.0.0.1) — the analogue of `accelerate launch` in the reference (README.md:165)."""
    import socket
    have = torch.cuda.device_count()
    if have < n:
        raise SystemExit(f"bench.py: --gpus {n} but only {have} GPU(s) are visible; refusing to report a smaller job")
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    sys.stdout.flush()
    os.execv(sys.executable, cmd)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--workload", default="readme_text_b64")
    ap.add_argument("--batch", type=int, default=None, help="per-GPU batch (default: the workload's)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-scan-events", action="store_true")
    ap.add_argument("--scan-events-every", type=int, default=7, help="bracket every n-th scan launch of the timed region with HIP events (1: all)")
    ap.add_argument("--no-check", action="store_true",
                    help="skip the untimed self-check forward (kernel traces of the timed path: its unfused composition runs library GEMMs)")
    ap.add_argument("--cpu-baseline-only", default=None, help=argparse.SUPPRESS)      # child-process leg of cpu_baseline()
    args = ap.parse_args()
    if args.cpu_baseline_only:
        fn = {"reference": cpu_baseline_reference, "port-torch": cpu_baseline_port_torch}.get(args.cpu_baseline_only, cpu_baseline_port)
        print(json.dumps(fn(WORKLOADS[args.workload])), flush=True)
        return

    from zigma_amd import sharded_sampling as ss
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the HIP path has no CPU fallback)")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        respawn_under_launcher(args.gpus)                 # does not return
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    rank, world, _ = ss.init_from_env(backend="nccl", device=device)
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but the job has WORLD_SIZE={world}")
    dist = None
    if world > 1:
        import torch.distributed as dist

    wl = WORKLOADS[args.workload]
    batch = args.batch or wl["batch"]
    model = build_model(wl["model"], device, torch.bfloat16)
    x, t, y = make_inputs(wl, batch, device, seed=1234 + rank)      # per-rank seed, sample_acc.py:58
    L = (wl["model"]["img_dim"] // wl["model"]["patch_size"]) ** 2
    gathered = torch.empty((world * batch,) + wl["x"], device=device) if world > 1 else None

    timer = ScanTimer(every=args.scan_events_every)
    timer.install()

    def step():
        with torch.no_grad():
            v = model(x, t, y)
        if world > 1:
            ss.gather_samples(v, world, out=gathered)
        return v

    for _ in range(args.warmup):          # scan-event timing only inside the timed region
        step()
    v_check = step()                      # (every rank: the step holds the gather)
    check = check_against_unfused(model, x, t, y, v_check) if (rank == 0 and not args.no_check) else None       # outside the timed region
    timer.enabled = not args.no_scan_events
    elapsed = ss.timed_steps(step, args.steps, 0, device, world)
    timer.enabled = False

    if rank == 0:
        Di, N = 2 * wl["model"]["embed_dim"], 16
        algo_bytes = batch * L * (4 * 2 * Di + 2 * 2 * N) + 4 * Di * (N + 2)       # BASELINE.md §2 / SURVEY §8d, bf16 I/O: u, delta, z, out_z + B, C
        from zigma_amd import selective_scan_interface as _ssi
        R = -(-wl["model"]["embed_dim"] // 16)
        dt_in = _ssi.DT_PROJ_IN_SCAN and not _ssi.split_chunk_len(batch, Di, L)
        # with dt_proj + softplus inside the kernel (round 4) the delta stream does not exist: what the launch really moves is
        # u, z, out_z + the x_dbl row (dt columns, B, C) + W_dt
        moved_bytes = batch * L * (3 * 2 * Di + 2 * (R + 2 * N)) + 4 * Di * (N + 2) + 2 * Di * R if dt_in else algo_bytes
        roof = None
        if timer.pairs:
            ms = timer.mean_ms()
            ach = algo_bytes / (ms * 1e-3)
            traffic, traffic_src = (None, None) if (args.no_cpu_baseline or world > 1) else pmc_traffic_live()      # (both are the slow, untimed legs; N = 1 only)
            if traffic is None:
                traffic, traffic_src = pmc_traffic()
            # VALU floor of the recurrence at the guide's issue rates (MI355X_MICROARCH.md: v_fma_f32 2 cycles per wave64
            # instruction per SIMD, transcendental quarter rate = 8): 4 plain + 1 exp per (element, state), 1024 SIMDs
            groups = batch * L * Di * N / 64
            valu_floor_us = groups * (4 * 2 + 8) / (1024 * SCLK) * 1e6
            # the same floor at the rates this chip sustains (tools/ubench2, profiles/r02_ubench2_valu_rates.txt: wall ns per
            # wave instruction per SIMD — v_exp_f32 3.43, v_pk_mul/fma_f32 2.28 for two results, plain VOP2 1.35): per 4 states
            # 4 exp + 6 packed + 4 plain = 32.8 ns
            rp, rk, re_ = measured_valu_rates()
            valu_floor_measured_us = groups / 4 * (4 * re_ + 6 * rk + 4 * rp) * 1e-9 / 1024 * 1e6
            roof = dict(bound="hbm", kernel="scan_tok2 (fused zigzag selective scan" + (", dt_proj + softplus inside" if dt_in else "") + ")", achieved=ach / 1e9,
                        peak=HBM_PEAK / 1e9, unit="GB/s", frac=ach / HBM_PEAK, traffic=traffic,
                        traffic_source=traffic_src, launch_us=ms * 1e3, launches=len(timer.pairs), launches_bracketed="every %d-th of %d" % (timer.every, timer.count),
                        algorithmic_bytes=algo_bytes, algorithmic_bytes_note="SURVEY 8(d) formula of the selective scan (u, delta, z, out_z, B, C)",
                        bytes_moved_by_design=moved_bytes, frac_of_bytes_moved=moved_bytes / (ms * 1e-3) / HBM_PEAK,
                        dt_proj_inside=bool(dt_in), limiter="valu", valu_floor_us=valu_floor_us,
                        valu_frac=valu_floor_us / (ms * 1e3), valu_floor_measured_rates_us=valu_floor_measured_us,
                        valu_frac_measured_rates=valu_floor_measured_us / (ms * 1e3),
                        valu_rates_source="profiles/valu_rates_gfx950.json <- profiles/r02_ubench2_valu_rates.txt (tools/ubench2: ns per wave-instruction per SIMD)",
                        idle_accounting="profiles/r06_c_scan_idle_probe_b64.json (tools/scan_idle_probe.py, same instruction stream with its waits removed one by one): "
                                        "VALU stream alone 224-229 us; + row loads 237-241; + LDS operand reads / y hand-over / barriers = the kernel, 255-264 us stand-alone")
        line = dict(metric="denoiser-forward latents/sec (BxL tokens/s), ZigMa d=640 L=32^2",
                    value=world * batch * L * args.steps / elapsed, unit="tokens/s", n_gpus=world, steps=args.steps,
                    warmup=args.warmup, ms_per_step=elapsed / args.steps * 1e3, higher_is_better=True, scaling="weak",
                    vs_baseline=None, dtype="bf16", data="synthetic",
                    config=dict(workload=f"{args.workload}: ZigMa(in_ch=3,img=32,E=640,depth=18,zigzagN8,has_text 77x768), "
                                         f"B={batch}/GPU, bf16, one forward per step",
                                global_batch=world * batch, seq_len=L, parallelism=f"batch-sharded x{world}"),
                    roofline=roof, check=check)
        if not args.no_check:
            line["box_calib"] = box_calib(device)
            ref = check_against_reference(model, wl, x, t, y)        # (overwrites the model's weights: last use of the model)
            if ref is not None:
                line["check"] = dict(check or {}, **ref)
        if world == 1 and not args.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(wl, args.workload)
        print(json.dumps(line), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
