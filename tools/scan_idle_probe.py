"""Round 6: WHERE the scan kernel's idle VALU time goes.

The shipped instruction mix costs 213 us at the measured issue rates; the kernel takes 265-290.  This tool times the SAME instruction stream with one
cause of waiting removed at a time — compile-time probe builds of csrc/scan_tok2.inc (results wrong, VALU instructions unchanged):

  NOBC     B_l / C_l operands without their two ds_read_b128 broadcast reads per step (an opaque register value instead)
  NODTDU   (dt, dt u) without its ds_read_b64 per step
  NOY      no partial-y hand-over (16 ds_write_b32 per tile-wave, 16 reads in the epilogue)
  NOBAR    no workgroup barriers
  NOLOAD   no 16-bit row loads (u, u', z) — the round-5 probe
and combinations up to "nothing but the VALU stream".  Every variant is its own library (only scan_tok_bf16.o differs; the other objects come from the
main build), all are loaded into ONE process and timed in interleaved rounds with HIP events.

    python tools/scan_idle_probe.py --build      # in the container (hipcc): tools/probe_libs/libzigma_scanprobe_<name>.so
    python tools/scan_idle_probe.py              # on the GPU box: one JSON line, also written to gpurun_out/scan_idle_probe.json
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
LIBDIR = os.path.join(ROOT, "tools", "probe_libs")
VARIANTS = {
    "base": [],
    "no_bc_reads": ["NOBC"],
    "no_dtdu_reads": ["NODTDU"],
    "no_y_handover": ["NOY"],
    "no_barriers": ["NOBAR"],
    "no_row_loads": ["NOLOAD"],
    "no_operand_reads": ["NOBC", "NODTDU"],
    "no_lds_in_core": ["NOBC", "NODTDU", "NOY"],
    "no_lds_no_barriers": ["NOBC", "NODTDU", "NOY", "NOBAR"],
    "valu_stream_only": ["NOBC", "NODTDU", "NOY", "NOBAR", "NOLOAD"],
}


def lib_path(name):
    return os.path.join(LIBDIR, f"libzigma_scanprobe_{name}.so")


def build_all():
    from zigma_amd import build as zb
    zb.build(verbose=False)
    os.makedirs(LIBDIR, exist_ok=True)
    objs = [os.path.join(zb.OBJ, s.replace(".hip", ".o")) for s in zb.SOURCES]
    for name, defs in VARIANTS.items():
        o = os.path.join(LIBDIR, f"scan_tok_bf16_{name}.o")
        cmd = [zb.HIPCC, *zb.FLAGS, *[f"-DZIGMA_SCAN_PROBE_{d}" for d in defs], "-c", os.path.join(zb.CSRC, "scan_tok_bf16.hip"), "-o", o]
        subprocess.run(cmd, check=True, capture_output=True)
        link = [x if not x.endswith("scan_tok_bf16.o") else o for x in objs]
        subprocess.run([zb.HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", *link, "-o", lib_path(name)], check=True)
        os.remove(o)
        print("built", lib_path(name), flush=True)


def main():
    import torch
    from zigma_amd import _lib
    from zigma_amd.selective_scan_interface import scan_raw
    handles = {}

    def use(path):
        if path not in handles:
            _lib._lib, _lib.LIB_PATH = None, path
            handles[path] = _lib.lib()
        _lib._lib, _lib.LIB_PATH = handles[path], path

    dev, dt = "cuda", torch.bfloat16
    B, L, Di, N, R = int(os.environ.get("B", 64)), 1024, int(os.environ.get("DI", 1280)), 16, 40
    torch.manual_seed(0)
    xz = torch.randn(B, L, 2 * Di, device=dev, dtype=dt)
    u = torch.randn(B, L, Di, device=dev, dtype=dt)
    xdbl = torch.randn(B, L, R + 2 * N, device=dev, dtype=dt)
    w = (R ** -0.5 * torch.randn(Di, R, device=dev)).to(dt)
    db = torch.randn(Di, device=dev) - 3
    A = -torch.exp(torch.log(torch.arange(1, N + 1, device=dev).float()) + 0.1 * torch.randn(Di, N, device=dev)).contiguous()
    D = torch.randn(Di, device=dev)
    perm = torch.randperm(L, device=dev).to(torch.int32)
    Bv, Cv = xdbl[:, :, R:R + N].transpose(1, 2).unsqueeze(1), xdbl[:, :, R + N:].transpose(1, 2).unsqueeze(1)
    z = xz[:, :, Di:].transpose(1, 2)
    y = torch.empty(B, L, Di, device=dev, dtype=dt)

    def run(name):
        use(lib_path(name))
        scan_raw(u.transpose(1, 2), None, A, Bv, Cv, D, z, db, True, out_z=y.transpose(1, 2), z_row_index=perm, out_row_index=perm, want_out=False,
                 dt_x=xdbl, dt_w=w)
        return _lib.last_kernel()

    names = [n for n in VARIANTS if os.path.exists(lib_path(n))]
    kern = {n: run(n) for n in names}
    torch.cuda.synchronize()
    times = {n: [] for n in names}
    for _ in range(6):
        for n in names:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                run(n)
            e1.record()
            torch.cuda.synchronize()
            times[n].append(e0.elapsed_time(e1) / 10 * 1e3)
    med = {n: sorted(v)[len(v) // 2] for n, v in times.items()}
    res = dict(shape=f"B={B} L={L} Di={Di} N={N} R={R} bf16, dt_proj inside", kernels=kern, us_median=med, us_min={n: min(v) for n, v in times.items()},
               saved_vs_base_us={n: med["base"] - v for n, v in med.items()}, defines={n: VARIANTS[n] for n in names})
    print(json.dumps(res))
    os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
    json.dump(res, open(os.path.join(ROOT, "gpurun_out", f"scan_idle_probe_b{B}.json"), "w"), indent=1)


if __name__ == "__main__":
    build_all() if "--build" in sys.argv else main()
