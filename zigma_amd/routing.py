"""ONE routing table for the dense projections of the block loop.

    route(role, tokens, n, k) -> Route(kernel, fuse_add, row)

role   = which projection of the reference's Block is asked for — it fixes the epilogue the call carries:
           "in_proj"  x @ W_in^T            (mamba_simple.py:290-294), no epilogue
           "out_proj" y @ W_out^T           (selective_scan_interface.py:365), optionally the block's gated add n + gate * (.)
           "to_q"     xa @ W_q^T            (model_zigma.py:104-106), no epilogue
           "to_out"   o @ W_o^T + bias      (model_zigma.py:128-135), bias + the block's gated add h + gate * (.)
kernel = which kernel family of zigma_linear_fwd serves it (KERNELS below), or "library" (F.linear = hipBLASLt) — every cell that falls
         to the library is an explicit row of the table or the explicit last row, and tests/test_host_cpu.py::test_routing_table sweeps
         E x tokens and compares the library cells with a reviewed list.
fuse_add = the caller hands the gated residual to the kernel's epilogue (True) or leaves it to the next norm kernel (False).

The table holds POLICY (what is fastest where, with the measurement that decided it); the shape LIMITS of each kernel family are the
`serves_*` functions, mirrored from the C side (csrc/linear_ws.hip linear_ws_panel, csrc/linear_sm.hip linear_sm_blocks, csrc/linear4w.hip
linear4w_variant, zigma_linear_fwd) — a row never returns a kernel whose limits the shape does
not meet.  Pointer / stride alignment is checked on the tensors by linear.py's `*_eligible` predicates when the call is made; a refusal there
lands on the library and is counted in REFUSED (never silent: tests assert it stays empty on the shipped shapes).

Knobs (tools / tests; no environment variable of their own — ZIGMA_KNOBS="routing.POLICY=off,routing.DISABLED=in_proj.ws+to_q.sm"):
  POLICY    "auto" the table;  "all" every projection the tiled kernels can serve runs on them (fused epilogues on);  "off" library only
  DISABLED  row ids skipped ("+"-separated string or a set): A/B of one row against what the table holds below it
"""
from collections import namedtuple

from . import _knobs

Route = namedtuple("Route", "kernel fuse_add row")
Row = namedtuple("Row", "id role k n tokens kernel fuse_add why")

KERNELS = ("ws", "ws128", "sm", "tiled", "tiled_halves", "library")
#   ws           csrc/linear_ws.hip, 256-feature weight panels resident in registers (k = 512 / 640)
#   ws128        the same kernel's 128-feature-panel form (k = 1280 / 1536)
#   sm           csrc/linear_sm.hip, one 128-token x n/4-feature tile per workgroup (few tokens)
#   tiled        zigma_linear_fwd's default: the generated 4-wave kernel (csrc/linear4w.hip) from 256 tiles on, the 8-wave kernel (csrc/linear.hip) below
#   tiled_halves two launches of `tiled`, one per half of the output columns
#   library      F.linear

POLICY = "auto"
DISABLED = ""
INF = 1 << 40
REFUSED = []          # (role, tokens, n, k, kernel) of calls a table row chose and the tensor-level check of linear.py turned down (the last 64 of them)


# ---- shape limits of the kernel families (C side mirrored) ---------------------------------------------------------------------------------
def serves_tiled(tokens, n, k):
    return tokens > 0 and tokens % 8 == 0 and n % 128 == 0 and k % 64 == 0


def tiles_4w(tokens, n):
    return (tokens // 256) * -(-n // 256)


def serves_4w(tokens, n, k):
    """the generated one-wave-per-SIMD kernel takes the call (linear4w_variant): whole 256-token tiles, at least one tile per CU"""
    return tokens % 256 == 0 and n % 128 == 0 and k % 64 == 0 and k >= 192 and tiles_4w(tokens, n) >= 256


def ws_panel_width(k):
    return 256 if k in (512, 640) else 128 if k in (1280, 1536) else 0


def serves_ws(tokens, n, k):
    """linear_ws_panel: instantiated k, whole panels, at most 32 of them, every workgroup of an XCD owns a 512-token tile"""
    pw = ws_panel_width(k)
    if not pw or n % pw or n > 8192 or n // pw > 32 or tokens % 512:
        return False
    return tokens // 512 >= 32 // (n // pw)


def serves_sm(tokens, n, k):
    return k % 64 == 0 and k >= 128 and tokens % 128 == 0 and tokens >= 128 and n % 128 == 0


_SERVES = {"ws": lambda t, n, k: serves_ws(t, n, k) and ws_panel_width(k) == 256,
           "ws128": lambda t, n, k: serves_ws(t, n, k) and ws_panel_width(k) == 128,
           "sm": serves_sm, "tiled": serves_tiled,
           "tiled_halves": lambda t, n, k: n % 512 == 0 and serves_tiled(t, n // 2, k),
           "library": lambda t, n, k: True}


# ---- the table: first matching enabled row whose kernel serves the shape wins ----------------------------------------------------------------
# k / n: a set of values, or (lo, hi) inclusive; tokens: (lo, hi) inclusive.
TABLE = (
    # -- in_proj: E -> 4E, no epilogue
    Row("in_proj.ws", "in_proj", {512, 640}, (1024, 8192), (8192, INF), "ws", False,
        "W_in panels resident in registers, only tokens stream: 33.5 / 52 / 93 / 186 us at 8192 ... 65 536 tokens against 41.5 / 72 / 106 / 200 (library), "
        "profiles/r04_g_linear_ws_probe.jsonl, r05_e_bench_kernel_stats.csv"),
    Row("in_proj.library_k768", "in_proj", {768}, (2048, INF), (65536, INF), "library", False,
        "E = 768 at >= 65 536 tokens: hipBLASLt 262-270 us against 276-303 for the 4-wave kernel (weights of k = 768 do not fit the weight-stationary form: 384 registers per "
        "lane); inside config 3y's forward 23.44-23.55 ms per evaluation with the library against 24.16-24.27 (one launch) / 24.16-24.19 (halves): -3.0 %, round 6 A/B on one box "
        "(DESIGN.md §0).  A plain GEMM without epilogue: the one place the library is the faster kernel"),
    Row("in_proj.tiled_wide_k", "in_proj", (704, INF), (2048, INF), (8192, INF), "tiled", False,
        "E = 768 (every shipped yaml): ONE launch of the 4-wave kernel, 49 / 71 / 139 / 268 us against 60 / 72 / 137 / 263 (library) and 58 / 93 / 141 / 275 as halves, "
        "profiles/r05_b_shapes_probe.jsonl; needs serves_4w (checked below)"),
    Row("in_proj.tiled_narrow", "in_proj", (192, INF), (128, 1024), (8192, INF), "tiled", False,
        "E <= 256: n <= 1024 is where the 4-wave kernel at least ties the library (as to_q); needs serves_4w"),
    Row("in_proj.halves", "in_proj", (64, INF), (2048, INF), (32768, INF), "tiled_halves", False,
        "a half's weight panel stays in an XCD's L2: 2 x 100 us against 215-222 as one launch and 190-200 for the library at E = 640 (round 3)"),
    Row("in_proj.library", "in_proj", (1, INF), (1, INF), (1, INF), "library", False, "below 8192 tokens the library ties or wins (4096: a tie)"),
    # -- out_proj: 2E -> E, optional gated add
    Row("out_proj.sm", "out_proj", (128, INF), (128, INF), (2048, 8192), "sm", False,
        "one round of 128 x n/4 tiles: 18.9 us at 8192 tokens (E = 640) against 22.4 library / 25.3 ws128; E = 768 24.2 against 25.9 / 33.8; its gated add stays in the "
        "next norm kernel (fused: config 5 +2 %), profiles/r05_k_/r05_l_shapes_probe*.jsonl, r05_p_few_token_fuse_ab.jsonl"),
    Row("out_proj.tiled", "out_proj", (192, INF), (128, INF), (16384, INF), "tiled", True,
        "from 256 tiles on the 4-wave kernel carries the gated add: 117-122 us at 65 536 tokens (E = 640) against 116 + the add in the norm kernel; needs serves_4w"),
    Row("out_proj.ws128", "out_proj", {1280, 1536}, (128, 4096), (8193, 32767), "ws128", False,
        "16 384 tokens: two rounds of the few-token kernel's 144 KB workgroups lose in the forward (7.05 vs 6.83 ms); 128-feature panels 38 us, add in the next norm"),
    Row("out_proj.library", "out_proj", (1, INF), (1, INF), (1, INF), "library", False, "shapes none of the above holds (e.g. k = 1024 at 16 384 tokens)"),
    # -- to_q: E -> 512, no epilogue
    Row("to_q.sm", "to_q", (128, INF), (128, INF), (2048, 8192), "sm", False,
        "11.5 us at 8192 tokens against 19.8 (library) / 17.2 (ws) / 20.9 (8-wave), profiles/r05_l_shapes_probe_linear_sm_128.jsonl"),
    Row("to_q.tiled", "to_q", (64, INF), (128, INF), (8192, INF), "tiled", False,
        ">= 32 768 tokens the 4-wave kernel ties the library (46-47 vs 44-47 us at 65 536); 16 384: the 8-wave kernel 22.7-23.7 vs 20.9-21.9 — taken so no library "
        "GEMM is left in the block loop (B = 16 forward 7.17 vs 7.16-7.35 ms)"),
    Row("to_q.library", "to_q", (1, INF), (1, INF), (1, INF), "library", False, "below 2048 tokens"),
    # -- to_out: 512 -> E, bias + gated add
    Row("to_out.sm", "to_out", (128, INF), (128, INF), (2048, 8192), "sm", True, "bias + gated add in the few-token kernel's epilogue (round 5)"),
    Row("to_out.tiled", "to_out", (64, INF), (128, 4096), (8, INF), "tiled", True,
        "the library cannot fuse bias + gated add: 54 us (67 in the forward) against 70 + the add (round 3)"),
    Row("to_out.library", "to_out", (1, INF), (1, INF), (1, INF), "library", False, "shapes the tiled kernels do not serve"),
)
ROLES = ("in_proj", "out_proj", "to_q", "to_out")

_knobs.apply(globals(), "routing")


def _in(v, spec):
    return v in spec if isinstance(spec, (set, frozenset)) else spec[0] <= v <= spec[1]


def _disabled():
    return DISABLED if isinstance(DISABLED, (set, frozenset)) else {s for s in str(DISABLED).split("+") if s}


def route(role, tokens, n, k):
    """the table's decision for a bf16 inference call of `role` with `tokens` rows, n output and k input features"""
    if role not in ROLES:
        raise ValueError(f"routing: unknown role {role!r}")
    if POLICY == "off":
        return Route("library", False, "policy.off")
    if POLICY == "all":
        if serves_tiled(tokens, n, k):
            return Route("tiled", role in ("out_proj", "to_out"), "policy.all")
        return Route("library", False, "policy.all")
    off = _disabled()
    for row in TABLE:
        if row.role != role or row.id in off or not (_in(k, row.k) and _in(n, row.n) and _in(tokens, row.tokens)):
            continue
        if not _SERVES[row.kernel](tokens, n, k):
            continue
        if row.id in ("in_proj.tiled_wide_k", "in_proj.tiled_narrow", "out_proj.tiled") and not serves_4w(tokens, n, k):
            continue              # (these two rows are only faster than what follows them on the 4-wave kernel)
        return Route(row.kernel, row.fuse_add, row.id)
    raise AssertionError("routing: the table's last row of every role matches everything")


def kernel_name(route_or_kernel, tokens=None, n=None, k=None):
    """what zigma_last_kernel() reports for a call served by this route (prefix for the families with several tile shapes)"""
    kern = route_or_kernel.kernel if isinstance(route_or_kernel, Route) else route_or_kernel
    if kern in ("tiled", "tiled_halves"):
        if tokens is None:
            return "linear"
        nn = n // 2 if kern == "tiled_halves" else n
        return ("linear4w_256x256+128" if nn % 256 else "linear4w_256x256") if serves_4w(tokens, nn, k) else "linear_tn_"
    return {"ws": "linear_ws", "ws128": "linear_ws_128", "sm": "linear_sm_", "library": "library"}[kern]
