"""Generate the golden fixtures under tests/golden/ by running the UNMODIFIED reference
(/root/reference) on CPU through oracle/ref_shim.py.

TEST INFRASTRUCTURE.  Run in the build container only (the GPU box has no /root/reference):

    python -m oracle.make_golden        (or: python oracle/make_golden.py — all four make_golden* scripts take both forms)

Everything stored is float32 / int64 numpy, produced by the reference's own functions:
  selective_scan_ref   dis_mamba/mamba_ssm/ops/selective_scan_interface.py:86-152
  mamba_inner_ref      ...:636-670
  causal_conv1d_ref    dis_causal_conv1d/causal_conv1d/causal_conv1d_interface.py:49-65
  rms_norm_ref / layer_norm_ref   dis_mamba/mamba_ssm/ops/triton/layernorm.py:19-48 (upcast=True)
  zigzag_path / hilbert_path / reverse_permut_np   utils/utils_zigzag.py
  ZigMa.forward        model_zigma.py:911-990
"""
import contextlib
import hashlib
import importlib.util
import io
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(os.path.dirname(HERE), "tests", "golden")
sys.path.insert(0, os.path.dirname(HERE))           # the repo root: `python -m oracle.make_golden` and `python oracle/make_golden.py` both work
from oracle import ref_shim  # noqa: E402


def npy(t):
    return t.detach().float().cpu().numpy() if torch.is_tensor(t) else t


def save(name, **arrs):
    path = os.path.join(OUT, name)
    np.savez_compressed(path, **{k: npy(v) for k, v in arrs.items() if v is not None})
    print(f"wrote {path}  ({os.path.getsize(path) / 1024:.1f} KiB)")


def scan_fixture(ssi, seqlen, groups=1, var_b=True, var_c=True, has_z=True, has_d=True, has_bias=True,
                 softplus=True, dim=4, dstate=8, batch=2):
    """The reference's own test recipe (dis_mamba/tests/ops/test_selective_scan.py:53-88), device cpu."""
    torch.random.manual_seed(0)
    A = -0.5 * torch.rand(dim, dstate)
    bshape = (dim, dstate) if not var_b else ((batch, dstate, seqlen) if groups == 1 else (batch, groups, dstate, seqlen))
    B = torch.randn(*bshape)
    cshape = (dim, dstate) if not var_c else ((batch, dstate, seqlen) if groups == 1 else (batch, groups, dstate, seqlen))
    C = torch.randn(*cshape)
    D = torch.randn(dim) if has_d else None
    z = torch.randn(batch, dim, seqlen) if has_z else None
    delta_bias = 0.5 * torch.rand(dim) if has_bias else None
    u = torch.randn(batch, dim, seqlen)
    delta = 0.5 * torch.rand(batch, dim, seqlen)
    out, last = ssi.selective_scan_ref(u, delta, A, B, C, D, z=z, delta_bias=delta_bias,
                                       delta_softplus=softplus, return_last_state=True)
    return dict(u=u, delta=delta, A=A, B=B, C=C, D=D, z=z, delta_bias=delta_bias, out=out, last_state=last,
                softplus=np.array(int(softplus)))


def main():
    os.makedirs(OUT, exist_ok=True)
    with contextlib.redirect_stdout(io.StringIO()):
        mz, ssi, cci, uz = ref_shim.reference_modules()

    # ---- scan-order tables -------------------------------------------------------------------
    tabs = {}
    with contextlib.redirect_stdout(io.StringIO()):
        for n in (4, 8, 16, 32):
            for i, p in enumerate(uz.zigzag_path(n)):
                tabs[f"zigzag_{n}_{i}"] = np.asarray(p, dtype=np.int64)
                tabs[f"zigzag_rev_{n}_{i}"] = np.asarray(uz.reverse_permut_np(p), dtype=np.int64)
            for i, p in enumerate(uz.hilbert_path(n)):
                tabs[f"hilbert_{n}_{i}"] = np.asarray(p, dtype=np.int64)
        digests = {}
        for n in (32, 128):
            digests[f"zigzag_{n}"] = [hashlib.sha256(np.asarray(p, dtype=np.int64).tobytes()).hexdigest()[:12]
                                      for p in uz.zigzag_path(n)]
            digests[f"hilbert_{n}"] = [hashlib.sha256(np.asarray(p, dtype=np.int64).tobytes()).hexdigest()[:12]
                                       for p in uz.hilbert_path(n)]
    for k, v in digests.items():
        tabs["digest_" + k] = np.array(v)
    save("paths.npz", **tabs)
    print("digests", digests)

    # ---- selective scan: reference test recipe, seed 0 --------------------------------------
    for name, kw in {
        "scan_L128": dict(seqlen=128), "scan_L1024": dict(seqlen=1024),
        "scan_L4096_sum": dict(seqlen=4096),
        "scan_g2_L256": dict(seqlen=256, groups=2),
        "scan_constBC_L128": dict(seqlen=128, var_b=False, var_c=False),
        "scan_constB_L128": dict(seqlen=128, var_b=False),
        "scan_plain_L100": dict(seqlen=100, has_z=False, has_d=False, has_bias=False, softplus=False),
        "scan_n16_L333": dict(seqlen=333, dim=6, dstate=16, batch=3),
    }.items():
        fx = scan_fixture(ssi, **kw)
        print(name, "out.sum", float(fx["out"].double().sum()), "absmean", float(fx["out"].abs().mean()),
              "state.sum", float(fx["last_state"].double().sum()))
        if name.endswith("_sum"):       # only the known-answer scalars (inputs regenerate from seed 0)
            fx = dict(out_sum=np.array(float(fx["out"].double().sum())),
                      out_absmean=np.array(float(fx["out"].abs().mean())),
                      out_head=fx["out"][0, 0, :3], state_sum=np.array(float(fx["last_state"].double().sum())))
        save(name + ".npz", **fx)

    # ---- causal conv1d ---------------------------------------------------------------------------
    torch.manual_seed(1)
    conv = {}
    for i, (b, d, l, w, bias, act) in enumerate([(2, 8, 151, 4, True, "silu"), (1, 5, 7, 3, False, None),
                                                 (3, 4, 2, 4, True, "silu"), (2, 6, 64, 2, True, None),
                                                 (2, 16, 1024, 4, True, "silu")]):
        x = torch.randn(b, d, l)
        wt = torch.randn(d, w)
        bs = torch.randn(d) if bias else None
        conv[f"x{i}"], conv[f"w{i}"], conv[f"b{i}"] = x, wt, bs
        conv[f"act{i}"] = np.array(0 if act is None else 1)
        conv[f"out{i}"] = cci.causal_conv1d_ref(x, wt, bs, act)
    conv["n"] = np.array(5)
    save("conv.npz", **conv)

    # ---- fused add + norm (reference torch refs, upcast=True == the Triton kernel's fp32 math) --
    spec = importlib.util.spec_from_file_location(
        "_ref_layernorm_real", ref_shim.REF + "/dis_mamba/mamba_ssm/ops/triton/layernorm.py")
    lnm = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(lnm)
    torch.manual_seed(2)
    x, r, w, b = torch.randn(3, 17, 40), torch.randn(3, 17, 40), torch.randn(40), torch.randn(40)
    y_rms, res_rms = lnm.rms_norm_ref(x, w, None, residual=r, eps=1e-5, prenorm=True, upcast=True)
    y_ln, res_ln = lnm.layer_norm_ref(x, w, b, residual=r, eps=1e-6, prenorm=True, upcast=True)
    y_rms0 = lnm.rms_norm_ref(x, w, None, residual=None, eps=1e-5, prenorm=False, upcast=True)
    save("norm.npz", x=x, r=r, w=w, b=b, y_rms=y_rms, res_rms=res_rms, y_ln=y_ln, res_ln=res_ln, y_rms0=y_rms0)

    # ---- mamba_inner_ref ---------------------------------------------------------------------------
    torch.manual_seed(3)
    Bsz, Di, L, N, R, W, E = 2, 48, 96, 16, 5, 4, 24
    xz = torch.randn(Bsz, 2 * Di, L)
    cw, cb = torch.randn(Di, 1, W) * 0.5, torch.randn(Di) * 0.1
    xw, dw = torch.randn(R + 2 * N, Di) * 0.2, torch.randn(Di, R) * 0.3
    ow, ob = torch.randn(E, Di) * 0.2, torch.randn(E) * 0.1
    A = -torch.exp(torch.randn(Di, N) * 0.5)
    D, dbias = torch.randn(Di), torch.rand(Di) * 0.5
    out = ssi.mamba_inner_ref(xz, cw, cb, xw, dw, ow, ob, A, None, None, D, delta_bias=dbias, delta_softplus=True)
    save("mamba_inner.npz", xz=xz, conv_w=cw, conv_b=cb, x_proj_w=xw, dt_proj_w=dw, out_proj_w=ow, out_proj_b=ob,
         A=A, D=D, delta_bias=dbias, out=out)

    # ---- whole model, small configs, fp32 --------------------------------------------------------
    def model_fixture(name, cfg, xshape, yfn, seed):
        torch.manual_seed(seed)
        with contextlib.redirect_stdout(io.StringIO()):
            m = mz.ZigMa(device="cpu", **cfg).eval()
        g = torch.Generator().manual_seed(seed + 100)
        with torch.no_grad():
            for blk in m.blocks:      # default init zeroes the gates -> mixers invisible
                blk.adaLN_modulation[-1].weight.normal_(std=0.5, generator=g)
                blk.adaLN_modulation[-1].bias.normal_(std=0.5, generator=g)
            if hasattr(m, "pos_embed") and cfg.get("use_pe", 0) == 2:
                m.pos_embed.normal_(std=0.02, generator=g)
            m.norm_f.weight.normal_(mean=1.0, std=0.1, generator=g)
            for blk in m.blocks:
                blk.norm.weight.normal_(mean=1.0, std=0.1, generator=g)
                blk.mixer.A_log.add_(torch.randn(blk.mixer.A_log.shape, generator=g) * 0.2)
                blk.mixer.D.normal_(mean=1.0, std=0.2, generator=g)
        x = torch.rand(*xshape, generator=g)
        t = torch.rand(xshape[0], generator=g)
        y = yfn(g)
        with torch.no_grad():
            out = m(x, t, y)
        arrs = {"sd." + k: v for k, v in m.state_dict().items()}
        arrs.update(x=x, t=t, y=y, out=out)
        arrs["cfg"] = np.array(repr(cfg))
        print(name, tuple(out.shape), "absmean", float(out.abs().mean()))
        save(name + ".npz", **arrs)

    model_fixture("zigma_text_zigzag2", dict(in_channels=3, embed_dim=32, depth=2, img_dim=8, patch_size=1,
                                            has_text=True, d_context=24, n_context_token=5, scan_type="zigzagN2",
                                            use_pe=2), (2, 3, 8, 8), lambda g: torch.rand(2, 5, 24, generator=g), 10)
    model_fixture("zigma_uncond_zigzag8", dict(in_channels=4, embed_dim=32, depth=9, img_dim=8, patch_size=1,
                                              scan_type="zigzagN8", use_pe=2), (2, 4, 8, 8), lambda g: None, 11)
    model_fixture("zigma_class_v2", dict(in_channels=4, embed_dim=32, depth=2, img_dim=8, patch_size=2,
                                        num_classes=10, scan_type="v2", use_pe=0), (3, 4, 8, 8),
                  lambda g: torch.randint(0, 10, (3,), generator=g), 12)
    model_fixture("zigma_hilbert2", dict(in_channels=4, embed_dim=32, depth=3, img_dim=8, patch_size=1,
                                        scan_type="hilbertN2", use_pe=1), (2, 4, 8, 8), lambda g: None, 13)
    model_fixture("zigma_video_sst", dict(in_channels=4, embed_dim=32, depth=6, img_dim=8, patch_size=2,
                                         num_classes=7, video_frames=3, scan_type="zzvideo_sst", use_pe=2, tpe=True),
                  (2, 3, 4, 8, 8), lambda g: torch.randint(0, 7, (2,), generator=g), 14)


if __name__ == "__main__":
    main()
