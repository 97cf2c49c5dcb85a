"""Round-5 A/B in ONE process (interleaved rounds, HIP events): the round-4 library (tools/libzigma_base_r04.so, built from the
round-4 sources) against the current one.
  scan      headline shape (B=64, L=1024, Di=1280, N=16, R=40, bf16, zigzag tables), dt_proj + softplus inside the kernel:
            r4 kernel | r5 kernel (step size in log2 units, D u as an fma) | r5 kernel with the pre-activated gate
  in_proj   weight-stationary kernel on the in_proj shape: r4 | r5 plain | r5 with silu on the gate half
  config 4  sequence-split scan (B=4, L=16384): r4 | r5 (priority rotation in both passes) | r5 with the rotation probed off
Prints one JSON line and writes gpurun_out/r05_scan_ab.json."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zigma_amd import _lib
from zigma_amd.linear import linear
from zigma_amd.selective_scan_interface import scan_raw

BASE = os.path.join(ROOT, "tools", "libzigma_base_r04.so")
NEW = _lib.LIB_PATH
PROBE_NL = os.path.join(ROOT, "tools", "libzigma_probe_noload.so")      # optional: zigma_amd.build.build(lib=..., extra_flags=("-DZIGMA_SCAN_PROBE_NOLOAD",)) — the scan without its 16-bit row loads (timing probe, results wrong)
_handles = {}


def use(path):
    if path not in _handles:
        _lib._lib, _lib.LIB_PATH = None, path
        _handles[path] = _lib.lib()
    _lib._lib, _lib.LIB_PATH = _handles[path], path


dev, dt = "cuda", torch.bfloat16
N, R, Di = 16, 40, 1280
torch.manual_seed(0)


def mk(B, L):
    d = {}
    d["xz"] = torch.randn(B, L, 2 * Di, device=dev, dtype=dt)
    d["u"] = torch.randn(B, L, Di, device=dev, dtype=dt)
    d["xdbl"] = torch.randn(B, L, R + 2 * N, device=dev, dtype=dt)
    d["w"] = (R ** -0.5 * torch.randn(Di, R, device=dev)).to(dt)
    d["db"] = torch.randn(Di, device=dev) - 3
    d["A"] = -torch.exp(torch.log(torch.arange(1, N + 1, device=dev).float()) + 0.1 * torch.randn(Di, N, device=dev)).contiguous()
    d["D"] = torch.randn(Di, device=dev)
    d["perm"] = torch.randperm(L, device=dev).to(torch.int32)
    d["Bv"] = d["xdbl"][:, :, R:R + N].transpose(1, 2).unsqueeze(1)
    d["Cv"] = d["xdbl"][:, :, R + N:].transpose(1, 2).unsqueeze(1)
    d["z"] = d["xz"][:, :, Di:].transpose(1, 2)
    d["zs"] = torch.nn.functional.silu(d["xz"][:, :, Di:].float()).to(dt).transpose(1, 2)
    d["delta"] = (0.5 * torch.rand(B, L, Di, device=dev)).to(dt)
    return d


h = mk(64, 1024)
outs = {}


def scan_dtp(name, lib, zact=False):
    def f():
        use(lib)
        y = outs.setdefault(name, torch.empty(64, 1024, Di, device=dev, dtype=dt))
        scan_raw(h["u"].transpose(1, 2), None, h["A"], h["Bv"], h["Cv"], h["D"], h["zs"] if zact else h["z"], h["db"], True,
                 out_z=y.transpose(1, 2), z_row_index=h["perm"], out_row_index=h["perm"], want_out=False, dt_x=h["xdbl"], dt_w=h["w"],
                 z_preactivated=zact)
    return f


# ---- config 4: sequence split
c4 = mk(4, 16384)
CH = 1024
xc = torch.empty(4, Di, 16384 // CH, 2 * N, device=dev, dtype=torch.float32)


def scan_c4(name, lib, flags=0):
    def f():
        use(lib)
        y = outs.setdefault(name, torch.empty(4, 16384, Di, device=dev, dtype=dt))
        scan_raw(c4["u"].transpose(1, 2), c4["delta"].transpose(1, 2), c4["A"], c4["Bv"], c4["Cv"], c4["D"], c4["z"], None, False,
                 out_z=y.transpose(1, 2), z_row_index=c4["perm"], out_row_index=c4["perm"], want_out=False, x=xc, chunk_len=CH, _probe_flags=flags)
    return f


# ---- in_proj
xin = torch.randn(65536, 640, device=dev, dtype=dt)
win = (640 ** -0.5 * torch.randn(2560, 640, device=dev)).to(dt)
o_in = torch.empty(65536, 2560, device=dev, dtype=dt)


def inproj(lib, silu=False):
    def f():
        use(lib)
        linear(xin, win, weight_stationary=True, out=o_in, silu_from_col=1280 if silu else None)
    return f


h16 = mk(16, 1024)


def scan_b16(name, lib):
    def f():
        use(lib)
        y = outs.setdefault(name, torch.empty(16, 1024, Di, device=dev, dtype=dt))
        scan_raw(h16["u"].transpose(1, 2), None, h16["A"], h16["Bv"], h16["Cv"], h16["D"], h16["z"], h16["db"], True,
                 out_z=y.transpose(1, 2), z_row_index=h16["perm"], out_row_index=h16["perm"], want_out=False, dt_x=h16["xdbl"], dt_w=h16["w"])
    return f


# ---- B = 8: sequence split as the model sizes it (160 (sample, slab) pairs -> chunks for ~768 workgroups)
from zigma_amd.selective_scan_interface import split_chunk_len
h8 = mk(8, 1024)
CH8 = split_chunk_len(8, Di, 1024)
xc8 = torch.empty(8, Di, -(-1024 // CH8), 2 * N, device=dev, dtype=torch.float32)


def scan_b8(name, lib, flags=0):
    def f():
        use(lib)
        y = outs.setdefault(name, torch.empty(8, 1024, Di, device=dev, dtype=dt))
        scan_raw(h8["u"].transpose(1, 2), h8["delta"].transpose(1, 2), h8["A"], h8["Bv"], h8["Cv"], h8["D"], h8["z"], None, False,
                 out_z=y.transpose(1, 2), z_row_index=h8["perm"], out_row_index=h8["perm"], want_out=False, x=xc8, chunk_len=CH8, _probe_flags=flags)
    return f


# ---- B = 16 (320 workgroups = 1.25 waves per SIMD in one pass): would the sequence split pay here too?  (the split cannot carry dt_proj:
# its cost is the dt_proj kernel + the two passes)
from zigma_amd.selective_scan_interface import dt_proj_softplus
xs16 = {c: torch.empty(16, Di, 1024 // c, 2 * N, device=dev, dtype=torch.float32) for c in (512, 256)}


def scan_b16_split(name, ch):
    def f():
        use(NEW)
        y = outs.setdefault(name, torch.empty(16, 1024, Di, device=dev, dtype=dt))
        dl = dt_proj_softplus(h16["xdbl"], R, h16["w"], h16["db"], True)
        scan_raw(h16["u"].transpose(1, 2), dl.transpose(1, 2), h16["A"], h16["Bv"], h16["Cv"], h16["D"], h16["z"], None, False,
                 out_z=y.transpose(1, 2), z_row_index=h16["perm"], out_row_index=h16["perm"], want_out=False, x=xs16[ch], chunk_len=ch)
    return f


# ---- the E = 768 models (Di = 1536, R = 48) at B = 64: 1536 workgroups = one round of six per CU (R6 form) vs a round of five + a tail
Di7, R7 = 1536, 48
torch.manual_seed(1)
g7 = dict(xz=torch.randn(64, 1024, 2 * Di7, device=dev, dtype=dt), u=torch.randn(64, 1024, Di7, device=dev, dtype=dt),
          xdbl=torch.randn(64, 1024, R7 + 2 * N, device=dev, dtype=dt), w=(R7 ** -0.5 * torch.randn(Di7, R7, device=dev)).to(dt), db=torch.randn(Di7, device=dev) - 3,
          A=-torch.exp(torch.log(torch.arange(1, N + 1, device=dev).float()) + 0.1 * torch.randn(Di7, N, device=dev)).contiguous(), D=torch.randn(Di7, device=dev),
          perm=torch.randperm(1024, device=dev).to(torch.int32))


def scan_e768(name, lib, flags=0):
    def f():
        use(lib)
        y = outs.setdefault(name, torch.empty(64, 1024, Di7, device=dev, dtype=dt))
        scan_raw(g7["u"].transpose(1, 2), None, g7["A"], g7["xdbl"][:, :, R7:R7 + N].transpose(1, 2).unsqueeze(1), g7["xdbl"][:, :, R7 + N:].transpose(1, 2).unsqueeze(1),
                 g7["D"], g7["xz"][:, :, Di7:].transpose(1, 2), g7["db"], True, out_z=y.transpose(1, 2), z_row_index=g7["perm"], out_row_index=g7["perm"],
                 want_out=False, dt_x=g7["xdbl"], dt_w=g7["w"], _probe_flags=flags)
    return f


PR = 1 << _lib.SCAN_PROBE_PRIO_SHIFT
groups = {
    "scan": {"r4": scan_dtp("s_r4", BASE), "r5": scan_dtp("s_r5", NEW), "r5_zact": scan_dtp("s_r5z", NEW, True),
             **({"r5_probe_no_row_loads": scan_dtp("s_nl", PROBE_NL)} if os.path.exists(PROBE_NL) else {})},
    "in_proj": {"r4": inproj(BASE), "r5": inproj(NEW), "r5_silu": inproj(NEW, True)},
    "scan_b16": {"r4": scan_b16("b_r4", BASE), "r5": scan_b16("b_r5", NEW), **({"r5_probe_no_row_loads": scan_b16("b_nl", PROBE_NL)} if os.path.exists(PROBE_NL) else {}), "r5_dtproj_plus_split_512": scan_b16_split("b_s512", 512),
                 "r5_dtproj_plus_split_256": scan_b16_split("b_s256", 256)},
    "scan_e768_b64": {"r4": scan_e768("g_r4", BASE), "r5_six_resident": scan_e768("g_r5", NEW), "r5_five_resident": scan_e768("g_r55", NEW, 1 << 10)},
    "scan_b8_split": {"r4": scan_b8("e_r4", BASE), "r5_rot": scan_b8("e_r5", NEW), "r5_norot": scan_b8("e_r5n", NEW, PR)},
    "config4_split": {"r4": scan_c4("c_r4", BASE), "r5_rot": scan_c4("c_r5", NEW), "r5_norot": scan_c4("c_r5n", NEW, PR)},
}
res = {}
for gname, fs in groups.items():
    for f in fs.values():
        f()
    torch.cuda.synchronize()
    times = {k: [] for k in fs}
    for rnd in range(7):
        for k, f in fs.items():
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10):
                f()
            e1.record(); torch.cuda.synchronize()
            times[k].append(e0.elapsed_time(e1) / 10 * 1e3)
    res[gname] = {"us_median": {k: round(sorted(v)[len(v) // 2], 2) for k, v in times.items()}, "us_min": {k: round(min(v), 2) for k, v in times.items()}}
rel = lambda a, b: float((a.float() - b.float()).norm() / b.float().norm())
res["scan"]["rel_diff_r5_vs_r4"] = rel(outs["s_r5"], outs["s_r4"])
res["scan"]["rel_diff_zact_vs_r4"] = rel(outs["s_r5z"], outs["s_r4"])
res["config4_split"]["rel_diff_r5_vs_r4"] = rel(outs["c_r5"], outs["c_r4"])
res["scan_b8_split"]["chunk_len"] = CH8
res["scan_e768_b64"]["six_vs_five_identical"] = bool(torch.equal(outs["g_r5"], outs["g_r55"]))
res["config4_split"]["rot_identical"] = bool(torch.equal(outs["c_r5"], outs["c_r5n"]))
algo = 64 * 1024 * (4 * 2 * Di + 2 * 2 * N) + 4 * Di * (N + 2)
res["scan"]["hbm_frac_formula"] = {k: round(algo / (v * 1e-6) / 8e12, 4) for k, v in res["scan"]["us_median"].items()}
print(json.dumps(res))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(res, open(os.path.join(ROOT, "gpurun_out", "r05_scan_ab.json"), "w"), indent=1)
