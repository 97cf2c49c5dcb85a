"""GPU tests of the small per-forward operators (csrc/embed.hip, csrc/skinny_linear.hip) against the torch compositions of the
reference formulas they replace (timm PatchEmbed + pos_embed, model_zigma.py:608-614,939-940; TimestepEmbedder :232-275; adaLN SiLU + Linear
:441; FinalLayer :313-337), evaluated in float64 on the same bf16 operands, and against the bf16 torch composition itself."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import rel_err

pytestmark = pytest.mark.gpu
DEV = "cuda"
BF = torch.bfloat16


def N(t):
    return t.detach().double().cpu().numpy()


@pytest.mark.parametrize("Bsz,C,H,p,E,with_pos,with_bias", [(64, 3, 32, 1, 640, True, True), (3, 4, 16, 2, 768, True, True),
                                                            (2, 4, 8, 1, 128, False, True), (5, 3, 12, 4, 64, True, False)])
def test_patch_embed_vs_unfold_gemm(Bsz, C, H, p, E, with_pos, with_bias):
    from zigma_amd import _lib
    from zigma_amd.embed import patch_embed
    g = torch.Generator(device="cpu").manual_seed(H * E)
    x = torch.randn(Bsz, C, H, H, generator=g).to(DEV, BF)
    w = (torch.randn(E, C, p, p, generator=g) * (C * p * p) ** -0.5).to(DEV, BF)
    b = torch.randn(E, generator=g).to(DEV, BF) if with_bias else None
    L = (H // p) ** 2
    pos = torch.randn(1, L, E, generator=g).to(DEV, BF) if with_pos else None
    got = patch_embed(x, w, b, pos)
    assert _lib.last_kernel() == "patch_embed" and got.shape == (Bsz, L, E)
    cols = x.reshape(Bsz, C, H // p, p, H // p, p).permute(0, 2, 4, 1, 3, 5).reshape(Bsz, L, C * p * p)
    ref = (cols.double() @ w.reshape(E, -1).double().T + (0 if b is None else b.double())).to(BF)       # the conv output is a bf16 tensor
    if pos is not None:
        ref = (ref.double() + pos.double()).to(BF)
    assert rel_err(N(got), N(ref)) < 2e-3
    assert float((got != ref).float().mean()) < 0.02          # same rounding points: the few differences are fp32-order ties
    # strided input view (a channel slice of a larger tensor)
    xb = torch.randn(Bsz, C + 2, H, H + 8, generator=g).to(DEV, BF)
    xv = xb[:, 1:C + 1, :, 4:4 + H]
    assert torch.equal(patch_embed(xv, w, b, pos), patch_embed(xv.contiguous(), w, b, pos))


@pytest.mark.parametrize("Bsz,dim", [(64, 256), (3, 64)])
def test_timestep_embed_is_the_torch_composition(Bsz, dim):
    from zigma_amd.embed import timestep_embed
    from zigma_amd.model_zigma import TimestepEmbedder
    t = (torch.rand(Bsz, device=DEV) * 1000.0).to(BF)
    freqs = TimestepEmbedder.frequencies(dim, BF).to(DEV)
    got = timestep_embed(t, freqs, dim)
    ref = TimestepEmbedder.timestep_embedding(t, dim, dtype=BF, freqs=freqs).to(BF)
    assert got.shape == ref.shape
    args = t.double()[:, None] * freqs.double()[None]
    ref64 = torch.cat([torch.cos(args), torch.sin(args)], -1)
    assert float((got.double() - ref64).abs().max()) < 6e-3          # bf16 rounding of values in [-1, 1] + fp32 argument rounding at t f ~ 1000
    assert float((got.double() - ref.double()).abs().max()) < 8e-3   # (both round the same fp32 functions; the device libm may differ in the last ulp)


@pytest.mark.parametrize("m,k,n,silu,bias", [(64, 640, 18 * 6 * 640, True, True), (64, 256, 640, False, True), (2, 640, 640, True, True),
                                             (17, 1024, 48, False, False), (64, 128, 16, True, False)])
def test_skinny_linear_vs_float64(m, k, n, silu, bias):
    from zigma_amd import _lib
    from zigma_amd.embed import skinny_linear
    g = torch.Generator(device="cpu").manual_seed(m + k + n)
    x = torch.randn(m, k, generator=g).to(DEV, BF)
    w = (torch.randn(n, k, generator=g) * k ** -0.5).to(DEV, BF)
    b = torch.randn(n, generator=g).to(DEV, BF) if bias else None
    got = skinny_linear(x, w, b, silu=silu)
    assert _lib.last_kernel() == "skinny_linear_mfma"
    xa = F.silu(x.float()).to(BF) if silu else x                       # the reference's SiLU module returns a bf16 tensor
    ref = xa.double() @ w.double().T + (0 if b is None else b.double())
    assert rel_err(N(got), N(ref)) < 3e-3                               # bf16 output
    lib = F.linear(F.silu(x) if silu else x, w, b)
    assert rel_err(N(got), N(lib)) < 4e-3
    # strided activations (a column window of a wider tensor), deterministic
    xw = torch.randn(m, k + 64, generator=g).to(DEV, BF)
    assert torch.equal(skinny_linear(xw[:, 32:32 + k], w, b, silu=silu), skinny_linear(xw[:, 32:32 + k].contiguous(), w, b, silu=silu))


@pytest.mark.parametrize("rows,E,n_out", [(65536, 640, 3), (1000, 768, 16), (7, 2048, 1), (130, 128, 4)])
def test_final_layer_vs_float64(rows, E, n_out):
    from zigma_amd import _lib
    from zigma_amd.embed import final_layer
    g = torch.Generator(device="cpu").manual_seed(rows + E)
    x = (torch.randn(rows, E, generator=g) * 1.5 + 0.3).to(DEV, BF)
    w = (torch.randn(n_out, E, generator=g) * E ** -0.5).to(DEV, BF)
    b = torch.randn(n_out, generator=g).to(DEV, BF)
    got = final_layer(x.view(1, rows, E), w, b, 1e-6)
    assert _lib.last_kernel() == "final_layer" and got.shape == (1, rows, n_out)
    y = F.layer_norm(x.double(), (E,), eps=1e-6).to(BF)                 # the LayerNorm output is a bf16 tensor
    ref = y.double() @ w.double().T + b.double()
    assert rel_err(N(got[0]), N(ref)) < 3e-3
    lib = F.linear(F.layer_norm(x, (E,), eps=1e-6), w, b)
    assert rel_err(N(got[0]), N(lib)) < 5e-3


def test_model_with_and_without_the_embed_kernels(monkeypatch):
    """README text model, bf16: the forward through the four kernels equals the torch composition within bf16 noise, and the kernels are
    the ones that ran (patch_embed, timestep_embed, skinny_linear x 3, final_layer)."""
    from zigma_amd import _lib, embed
    from zigma_amd.model_zigma import ZigMa
    torch.manual_seed(0)
    m = ZigMa(in_channels=3, embed_dim=256, depth=4, img_dim=32, patch_size=1, has_text=True, d_context=96, n_context_token=11,
              scan_type="zigzagN8", use_pe=2, device=DEV, dtype=BF).eval()
    with torch.no_grad():
        for blk in m.blocks:
            blk.adaLN_modulation[-1].weight.normal_(std=0.05)
            blk.adaLN_modulation[-1].bias.normal_(std=0.3)
        m.final_layer.linear.weight.normal_(std=0.05)
        x, t, y = torch.randn(16, 3, 32, 32, device=DEV, dtype=BF), torch.rand(16, device=DEV), torch.randn(16, 11, 96, device=DEV, dtype=BF)
        _lib.TRACE = []
        out = m(x, t, y)
        names = [k for _, k, _ in _lib.TRACE]
        _lib.TRACE = None
        monkeypatch.setattr(embed, "USE_EMBED_KERNELS", False)
        ref = m(x, t, y)
    assert names.count("patch_embed") == 1 and names.count("timestep_embed") == 1 and names.count("final_layer") == 1
    assert names.count("skinny_linear_mfma") == 3              # timestep MLP (2) + the adaLN modulation of all blocks (1)
    assert torch.isfinite(out).all() and rel_err(N(out), N(ref)) < 1e-2
