"""Round 6: scan_quad_kernel (barrier-free quad layout, csrc/scan_quad.inc) against scan_tok2_kernel's in-kernel-dt_proj form (pinned by probe bit 11) in ONE
process, interleaved rounds: the headline shape (B=64, Di=1280), B=16, the E=768 shape (Di=1536, R=48: six-resident forms), with / without row tables.
Prints one JSON line; also gpurun_out/scan_quad_ab.json."""
import json, os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from zigma_amd import _lib
from zigma_amd.selective_scan_interface import scan_raw
dev, dt = "cuda", torch.bfloat16
N, L = 16, 1024
torch.manual_seed(0)


def mk(B, Di, R):
    d = dict(xz=torch.randn(B, L, 2 * Di, device=dev, dtype=dt), u=torch.randn(B, L, Di, device=dev, dtype=dt), xdbl=torch.randn(B, L, R + 2 * N, device=dev, dtype=dt),
             w=(R ** -0.5 * torch.randn(Di, R, device=dev)).to(dt), db=torch.randn(Di, device=dev) - 3, D=torch.randn(Di, device=dev),
             A=-torch.exp(torch.log(torch.arange(1, N + 1, device=dev).float()) + 0.1 * torch.randn(Di, N, device=dev)).contiguous(),
             perm=torch.randperm(L, device=dev).to(torch.int32), y=torch.empty(B, L, Di, device=dev, dtype=dt), R=R, Di=Di)
    return d


def run(d, flags, tab=True):
    R, Di = d["R"], d["Di"]
    scan_raw(d["u"].transpose(1, 2), None, d["A"], d["xdbl"][:, :, R:R + N].transpose(1, 2).unsqueeze(1), d["xdbl"][:, :, R + N:].transpose(1, 2).unsqueeze(1), d["D"],
             d["xz"][:, :, Di:].transpose(1, 2), d["db"], True, out_z=d["y"].transpose(1, 2), z_row_index=d["perm"] if tab else None, out_row_index=d["perm"] if tab else None,
             want_out=False, dt_x=d["xdbl"], dt_w=d["w"], _probe_flags=flags)
    return _lib.last_kernel()


PIN = 1 << _lib.SCAN_PROBE_TOK2_SHIFT
NOROT = 1 << _lib.SCAN_PROBE_PRIO_SHIFT
cases = {"b64": mk(64, 1280, 40), "b16": mk(16, 1280, 40), "b64_e768": mk(64, 1536, 48), "b32": mk(32, 1280, 40)}
variants = [(c, v, t) for c in cases for v in ("quad", "tok2", "quad_norot") for t in (True, False) if not (t is False and c != "b64")]
res, names = {}, {}
for c, v, t in variants:
    names[f"{c}/{v}/{'tab' if t else 'notab'}"] = run(cases[c], {"quad": 0, "tok2": PIN, "quad_norot": NOROT}[v], t)
torch.cuda.synchronize()
for rnd in range(6):
    for c, v, t in variants:
        f = {"quad": 0, "tok2": PIN, "quad_norot": NOROT}[v]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            run(cases[c], f, t)
        e1.record(); torch.cuda.synchronize()
        res.setdefault(f"{c}/{v}/{'tab' if t else 'notab'}", []).append(e0.elapsed_time(e1) / 10 * 1e3)
# agreement of the two kernels on the headline shape
run(cases["b64"], 0); yq = cases["b64"]["y"].clone(); run(cases["b64"], PIN); yt = cases["b64"]["y"].clone()
rel = float((yq.float() - yt.float()).norm() / yt.float().norm())
out = dict(us_median={k: sorted(v)[len(v) // 2] for k, v in res.items()}, us_min={k: min(v) for k, v in res.items()}, kernels=names, rel_diff_quad_vs_tok2_b64=rel,
           finite=bool(torch.isfinite(yq).all()))
print(json.dumps(out))
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "gpurun_out", "scan_quad_ab.json"), "w"), indent=1)
