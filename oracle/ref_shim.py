"""Import shims that let the UNMODIFIED reference (/root/reference) run on CPU.

TEST INFRASTRUCTURE ONLY.  Used by oracle/make_golden.py (in the build container,
where /root/reference exists) to generate the golden fixtures under tests/golden/, and by
bench.py's `cpu_baseline` leg to time the reference's own CPU path.  Nothing in the product
path imports this file.  The reference is Python and cannot travel: on the GPU box there is no
/root/reference and this module refuses to install (bench.py then times the numpy port, kind "port").

Recipe = SURVEY.md §8c:
  1. stand-in modules `causal_conv1d_cuda` / `selective_scan_cuda` backed by the
     reference's own pure-torch `causal_conv1d_ref` / `selective_scan_ref`
     (dis_causal_conv1d/causal_conv1d/causal_conv1d_interface.py:49-65,
      dis_mamba/mamba_ssm/ops/selective_scan_interface.py:86-152);
  2. bypass `dis_mamba/mamba_ssm/__init__.py` (it drags in transformers-4.36-only names);
  3. `timm.models.vision_transformer` shim (PatchEmbed, Mlp);
  4. the Triton `layernorm.py` replaced by an fp32 torch restatement of its forward
     semantics (layernorm.py:86-120,126-177,406-422);
  5. `create_block` wrapped so that `zzvideo_X` reaches Mamba as `video_X`
     (reference bug, SURVEY.md §8a notes) without touching any reference file.
"""
import importlib
import importlib.util
import sys
import types

import torch
import torch.nn as nn
import torch.nn.functional as F

import os

REF = "/root/reference"
_installed = False


def available():
    """True where the reference checkout is mounted (the build container)."""
    return os.path.isdir(REF)


def _mod(name, path=None):
    m = types.ModuleType(name)
    if path is not None:
        m.__path__ = [path]
    sys.modules[name] = m
    return m


def install():
    """Make `import model_zigma`, `import transport`, ... resolve to the reference."""
    global _installed
    if _installed:
        return
    if not available():
        raise RuntimeError("oracle.ref_shim: /root/reference is not mounted here (build container only)")
    _installed = True
    import matplotlib

    matplotlib.use("Agg")
    sys.path.insert(0, REF)
    sys.path.insert(0, REF + "/dis_causal_conv1d")

    # -- 1. extension stand-ins (filled in after the refs are importable) -----------
    cc = _mod("causal_conv1d_cuda")
    ss = _mod("selective_scan_cuda")

    # -- 2. package skeletons that skip dis_mamba/mamba_ssm/__init__.py --------------
    _mod("dis_mamba", REF + "/dis_mamba")
    _mod("dis_mamba.mamba_ssm", REF + "/dis_mamba/mamba_ssm")
    _mod("dis_mamba.mamba_ssm.ops", REF + "/dis_mamba/mamba_ssm/ops")
    _mod("dis_mamba.mamba_ssm.modules", REF + "/dis_mamba/mamba_ssm/modules")
    tri = _mod("dis_mamba.mamba_ssm.ops.triton", REF + "/dis_mamba/mamba_ssm/ops/triton")

    # -- 4. fp32 torch restatement of the Triton fused add+norm forward ---------------
    ln = _mod("dis_mamba.mamba_ssm.ops.triton.layernorm")

    def _norm_fwd(x, weight, bias, residual, eps, prenorm, residual_in_fp32, is_rms):
        xf = x.float()
        if residual is not None:
            xf = xf + residual.float()
            res_dtype = residual.dtype
        else:
            res_dtype = torch.float32 if residual_in_fp32 else x.dtype
        if is_rms:
            rstd = torch.rsqrt(xf.square().mean(-1, keepdim=True) + eps)
            y = xf * rstd * weight.float()
            if bias is not None:
                y = y + bias.float()
        else:
            y = F.layer_norm(xf, xf.shape[-1:], None if weight is None else weight.float(),
                             None if bias is None else bias.float(), eps)
        y = y.to(x.dtype)
        return (y, xf.to(res_dtype)) if prenorm else y

    def rms_norm_fn(x, weight, bias, residual=None, prenorm=False, residual_in_fp32=False, eps=1e-6):
        return _norm_fwd(x, weight, bias, residual, eps, prenorm, residual_in_fp32, True)

    def layer_norm_fn(x, weight, bias, residual=None, eps=1e-6, prenorm=False,
                      residual_in_fp32=False, is_rms_norm=False):
        return _norm_fwd(x, weight, bias, residual, eps, prenorm, residual_in_fp32, is_rms_norm)

    class RMSNorm(nn.Module):
        def __init__(self, hidden_size, eps=1e-5, device=None, dtype=None):
            super().__init__()
            self.eps = eps
            self.weight = nn.Parameter(torch.ones(hidden_size, device=device, dtype=dtype))
            self.register_parameter("bias", None)

        def forward(self, x, residual=None, prenorm=False, residual_in_fp32=False):
            return rms_norm_fn(x, self.weight, self.bias, residual=residual, eps=self.eps,
                               prenorm=prenorm, residual_in_fp32=residual_in_fp32)

    ln.rms_norm_fn, ln.layer_norm_fn, ln.RMSNorm = rms_norm_fn, layer_norm_fn, RMSNorm
    tri.layernorm = ln

    # -- 3. timm shim -------------------------------------------------------------------
    _mod("timm")
    _mod("timm.models")
    vt = _mod("timm.models.vision_transformer")

    class PatchEmbed(nn.Module):
        def __init__(self, img_size=224, patch_size=16, in_chans=3, embed_dim=768, bias=True):
            super().__init__()
            self.img_size = (img_size, img_size)
            self.patch_size = (patch_size, patch_size)
            self.grid_size = (img_size // patch_size, img_size // patch_size)
            self.num_patches = self.grid_size[0] * self.grid_size[1]
            self.proj = nn.Conv2d(in_chans, embed_dim, kernel_size=patch_size,
                                  stride=patch_size, bias=bias)

        def forward(self, x):
            return self.proj(x).flatten(2).transpose(1, 2)

    class Mlp(nn.Module):
        def __init__(self, in_features, hidden_features=None, out_features=None,
                     act_layer=nn.GELU, drop=0.0):
            super().__init__()
            out_features = out_features or in_features
            hidden_features = hidden_features or in_features
            self.fc1 = nn.Linear(in_features, hidden_features)
            self.act = act_layer()
            self.fc2 = nn.Linear(hidden_features, out_features)

        def forward(self, x):
            return self.fc2(self.act(self.fc1(x)))

    vt.PatchEmbed, vt.Mlp = PatchEmbed, Mlp

    # -- 1b. back the extension stand-ins with the reference's own *_ref functions ----
    cci = importlib.import_module("causal_conv1d.causal_conv1d_interface")
    ssi = importlib.import_module("dis_mamba.mamba_ssm.ops.selective_scan_interface")

    def causal_conv1d_fwd(x, weight, bias, silu):
        return cci.causal_conv1d_ref(x, weight, bias, "silu" if silu else None)

    def scan_fwd(u, delta, A, B, C, D, z, delta_bias, delta_softplus):
        out, last = ssi.selective_scan_ref(u, delta, A, B, C, D, z=None, delta_bias=delta_bias,
                                           delta_softplus=delta_softplus, return_last_state=True)
        n = A.shape[1]
        x = torch.zeros(u.shape[0], u.shape[1], 1, 2 * n, dtype=torch.float32)
        x[:, :, 0, 1::2] = last
        if z is None:
            return [out, x]
        out_z = (out.float() * F.silu(z.float())).to(u.dtype)
        return [out, x, out_z]

    cc.causal_conv1d_fwd = causal_conv1d_fwd
    ss.fwd = scan_fwd

    # -- 5. zzvideo_ -> video_ at the Mamba boundary --------------------------------------
    mz = importlib.import_module("model_zigma")
    _orig_create_block = mz.create_block

    def create_block(*a, scan_type="none", **kw):
        if scan_type.startswith("zzvideo_"):
            scan_type = "video_" + scan_type[len("zzvideo_"):]
        return _orig_create_block(*a, scan_type=scan_type, **kw)

    mz.create_block = create_block


def reference_modules():
    """Returns (model_zigma, selective_scan_interface, causal_conv1d_interface, utils_zigzag)."""
    install()
    return (importlib.import_module("model_zigma"),
            importlib.import_module("dis_mamba.mamba_ssm.ops.selective_scan_interface"),
            importlib.import_module("causal_conv1d.causal_conv1d_interface"),
            importlib.import_module("utils.utils_zigzag"))


def reference_layernorm():
    """The reference's own pure-torch `layer_norm_ref` / `rms_norm_ref` (layernorm.py:18-46), taken from its source
    file WITHOUT importing the module (its top level needs triton): the two function definitions are compiled as they
    stand into a namespace that only holds torch and torch.nn.functional."""
    import ast
    path = REF + "/dis_mamba/mamba_ssm/ops/triton/layernorm.py"
    tree = ast.parse(open(path).read(), filename=path)
    keep = [n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name in ("layer_norm_ref", "rms_norm_ref")]
    assert len(keep) == 2
    ns = {"torch": torch, "F": F}
    exec(compile(ast.Module(body=keep, type_ignores=[]), path, "exec"), ns)
    return types.SimpleNamespace(layer_norm_ref=ns["layer_norm_ref"], rms_norm_ref=ns["rms_norm_ref"])
