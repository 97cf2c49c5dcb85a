"""Seeded inputs of the backward parity cases — shared by oracle/make_golden_bwd.py (which runs the reference on them)
and tests/ (which run the HIP kernels and the numpy oracle on the SAME numbers).  torch's CPU generator is
bit-reproducible across machines, so the large inputs need not be stored in the fixtures.
TEST INFRASTRUCTURE ONLY."""
import torch

SCAN_CASES = {   # name: (batch, dim, seqlen, dstate, has_z, has_D, has_bias, softplus, seed, variable_BC)
    "bwd_scan_full": (2, 64, 100, 16, True, True, True, True, 1, True),
    "bwd_scan_plain": (2, 64, 48, 16, False, False, False, False, 2, True),
    "bwd_scan_n8": (3, 128, 33, 8, True, True, True, True, 3, True),
    "bwd_scan_long": (1, 64, 2300, 16, True, True, True, True, 4, True),     # crosses the 2048-step chunk boundary
    "bwd_scan_constbc": (2, 8, 40, 4, True, True, True, True, 5, False),
}
LONG_KEEP = list(range(0, 40)) + list(range(2030, 2070)) + list(range(2280, 2300))   # positions stored for the long case


def scan_inputs(name, dtype=torch.float32):
    Bsz, Dm, L, N, has_z, has_D, has_bias, softplus, seed, var_bc = SCAN_CASES[name]
    g = torch.Generator().manual_seed(seed)
    r = lambda *s: torch.randn(*s, generator=g, dtype=dtype)
    u, delta = r(Bsz, Dm, L), 0.5 * torch.rand(Bsz, Dm, L, generator=g, dtype=dtype)
    A = -0.5 * torch.rand(Dm, N, generator=g, dtype=dtype) - 0.05
    Bm = r(Bsz, N, L) if var_bc else r(Dm, N)
    Cm = r(Bsz, N, L) if var_bc else r(Dm, N)
    Dp = r(Dm) if has_D else None
    z = r(Bsz, Dm, L) if has_z else None
    db = 0.5 * torch.rand(Dm, generator=g, dtype=dtype) if has_bias else None
    if has_bias:
        db[0] = 25.0                          # softplus pass-through branch
    dout = r(Bsz, Dm, L)
    return dict(u=u, delta=delta, A=A, B=Bm, C=Cm, D=Dp, z=z, delta_bias=db, dout=dout, softplus=softplus)
