"""Stress of the weight-stationary projection kernel (round 6: a one-off 2.5e-2 error in tests/test_gpu_backward.py::test_linear_train_fn_on_the_own_kernels
[16384-1280-640] inside a full-suite run; 7 other runs of that test passed).  The test's own sequence — forward product and dX of four shapes through
wgrad.LinearTrainFn, fresh operands every round, allocator churn in between — repeated, every result compared with the 8-wave tiled kernel
(bit-identical by construction: same MFMA, same accumulation order).  Prints one JSON line per shape."""
import os, sys, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import zigma_amd.wgrad as wg
from zigma_amd.linear import linear
dev = "cuda"
ROUNDS = int(os.environ.get("ROUNDS", 60))
shapes = [(65536, 640, 2560, False), (16384, 1280, 640, False), (8192, 512, 640, True), (4096, 640, 512, False), (16384, 1536, 768, False), (24576, 1280, 640, False)]
bad = {s: [] for s in shapes}
g = torch.Generator().manual_seed(0)
for rnd in range(ROUNDS):
    for s in shapes:
        M, K, N, bias = s
        x = torch.randn(M, K, generator=g).to(dev, torch.bfloat16).requires_grad_(True)
        w = (torch.randn(N, K, generator=g) * K ** -0.5).to(dev, torch.bfloat16).requires_grad_(True)
        b = (torch.randn(N, generator=g) * 0.1).to(dev, torch.bfloat16).requires_grad_(True) if bias else None
        dy = torch.randn(M, N, generator=g).to(dev, torch.bfloat16)
        y = wg.linear_train(x, w, b)
        y.backward(dy)
        with torch.no_grad():
            ref = linear(x.detach(), w.detach(), None if b is None else b.detach(), _probe_flags=0x2000)
            refdx = linear(dy, w.detach().t().contiguous(), _probe_flags=0x2000)
        if not torch.equal(y.detach(), ref):
            rows = (y.detach() != ref).any(1).nonzero().flatten()
            bad[s].append(("fwd", rnd, int(rows.numel()), rows[:6].tolist(), float((y.detach().float() - ref.float()).abs().max())))
        if not torch.equal(x.grad, refdx):
            rows = (x.grad != refdx).any(1).nonzero().flatten()
            bad[s].append(("dx", rnd, int(rows.numel()), rows[:6].tolist(), float((x.grad.float() - refdx.float()).abs().max())))
        del x, w, b, dy, y, ref, refdx
        if rnd % 5 == 0:
            torch.cuda.empty_cache()
for s in shapes:
    print(json.dumps(dict(shape=s, rounds=ROUNDS, mismatching=len(bad[s]), first=bad[s][:6])), flush=True)
