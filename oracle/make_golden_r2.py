"""Round-2 golden fixtures at the BASELINE shapes, produced by the UNMODIFIED reference on CPU (oracle/ref_shim.py):

    python -m oracle.make_golden_r2            # build container only (needs /root/reference); ~5 min

  r2_readme_b2     BASELINE configs[0]/[1] model: README ZigMa (E=640, depth=18, has_text 77x768, zigzagN8), B=2
  r2_video_t16     BASELINE configs[4] layer shapes: zzvideo_sst, E=768, 16 frames x 256 tokens, depth 3 (s, s, t), B=1
  r2_l16384        BASELINE configs[3] shape: 128x128 latents, patch 1 -> L=16384, E=640, zigzagN8 (N=128 tables), depth 2, B=1
  r2_small_video16 a tiny 16-frame model (T % 16 == 0 -> the no-copy temporal path) for quick runs
  r6_zigzag8_e768  the SHIPPED image yaml's layer shapes (config/model/zigzag8_b1_pe2.yaml:4-10: 32x32x4 latents, patch 1, E=768, zigzagN8,
                   use_pe 2), depth 3, B=2 — the E=768 routes (in_proj 768 -> 3072, x_proj K=1536, out_proj 1536 -> 768, the six-resident scan)
  r6_sweep2_e768   the same with scan_type="v2" (config/model/sweep2_b1_pe2.yaml:4-10): forward + flipped scan per layer, two parameter sets
                   (mamba_simple.py:304-339, selective_scan_interface.py:155-224)

Each fixture holds the inputs and the reference's output in fp32 (`out`) AND the output of the reference's own bf16 run
(`ZigMa(dtype=torch.bfloat16)`, bf16 inputs, `out_bf16`) — model_zigma.py:575,812-813.  Weights are not stored: both
sides regenerate them with oracle/param_fill.py from `seed`."""
import contextlib
import io
import os
import sys
import time

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
sys.path.insert(0, os.path.dirname(HERE))
from oracle import ref_shim  # noqa: E402
from oracle.param_fill import fill_state  # noqa: E402

CASES = {
    "r2_readme_b2": dict(cfg=dict(in_channels=3, img_dim=32, embed_dim=640, depth=18, patch_size=1, has_text=True,
                                  d_context=768, n_context_token=77, scan_type="zigzagN8", use_pe=2),
                         x=(2, 3, 32, 32), y=("text", 77, 768), seed=21),
    "r2_video_t16": dict(cfg=dict(in_channels=4, img_dim=32, embed_dim=768, depth=3, patch_size=2, num_classes=101,
                                  video_frames=16, scan_type="zzvideo_sst", use_pe=2, tpe=True),
                         x=(1, 16, 4, 32, 32), y=("class", 101), seed=22),
    "r2_l16384": dict(cfg=dict(in_channels=4, img_dim=128, embed_dim=640, depth=2, patch_size=1, scan_type="zigzagN8",
                               use_pe=2),
                      x=(1, 4, 128, 128), y=None, seed=23),
    "r2_small_video16": dict(cfg=dict(in_channels=4, img_dim=8, embed_dim=64, depth=6, patch_size=2, num_classes=7,
                                      video_frames=16, scan_type="zzvideo_sst", use_pe=2, tpe=True),
                             x=(2, 16, 4, 8, 8), y=("class", 7), seed=24),
    "r6_zigzag8_e768": dict(cfg=dict(in_channels=4, img_dim=32, embed_dim=768, depth=3, patch_size=1, scan_type="zigzagN8", use_pe=2),
                            x=(2, 4, 32, 32), y=None, seed=61),
    "r6_sweep2_e768": dict(cfg=dict(in_channels=4, img_dim=32, embed_dim=768, depth=3, patch_size=1, scan_type="v2", use_pe=2),
                           x=(2, 4, 32, 32), y=None, seed=62),
}


def make_inputs(case):
    g = torch.Generator().manual_seed(case["seed"] + 1000)
    x = torch.randn(case["x"], generator=g)
    t = torch.rand(case["x"][0], generator=g)
    y = None
    if case["y"] is not None and case["y"][0] == "text":
        y = torch.rand((case["x"][0],) + tuple(case["y"][1:]), generator=g)
    elif case["y"] is not None:
        y = torch.randint(0, case["y"][1], (case["x"][0],), generator=g)
    return x, t, y


def main(only=None):
    with contextlib.redirect_stdout(io.StringIO()):
        mz, _, _, _ = ref_shim.reference_modules()
    for name, case in CASES.items():
        if only and name not in only:
            continue
        x, t, y = make_inputs(case)
        outs = {}
        for tag, dtype in (("out", torch.float32), ("out_bf16", torch.bfloat16)):
            with contextlib.redirect_stdout(io.StringIO()):
                m = mz.ZigMa(device="cpu", dtype=dtype, **case["cfg"]).eval()
            fill_state(m, case["seed"])
            yy = y if (y is None or y.dtype == torch.int64) else y.to(dtype)
            t0 = time.time()
            with torch.no_grad():
                o = m(x.to(dtype), t.to(dtype), yy)
            outs[tag] = o.float().numpy()
            print(f"{name} {tag}: {time.time() - t0:.1f} s, absmean {np.abs(outs[tag]).mean():.4f}", flush=True)
            del m
        rel = np.linalg.norm(outs["out_bf16"] - outs["out"]) / np.linalg.norm(outs["out"])
        print(f"{name}: the reference's own bf16 run vs its fp32 run: rel err {rel:.3e}")
        arrs = dict(x=x.numpy(), t=t.numpy(), cfg=np.array(repr(case["cfg"])), seed=np.array(case["seed"]),
                    ref_bf16_vs_fp32=np.array(rel), **outs)
        if y is not None:
            arrs["y"] = y.numpy()
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **arrs)
        print(f"wrote {path} ({os.path.getsize(path) / 1024:.1f} KiB)")


if __name__ == "__main__":
    main(sys.argv[1:])
