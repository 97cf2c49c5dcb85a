"""Weight-gradient GEMMs dW = dY^T X (K = tokens = 65536): which operand layout does the library run fastest?"""
import torch
dev, dt = "cuda", torch.bfloat16
M = 65536
shapes = {"in_proj": (2560, 640), "out_proj": (640, 1280), "to_q": (512, 640), "to_out": (640, 512)}
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / iters * 1e3
for name, (N, K) in shapes.items():
    dy = (torch.randn(M, N, device=dev) * 0.1).to(dt); x = (torch.randn(M, K, device=dev) * 0.3).to(dt)
    a = timeit(lambda: dy.t().mm(x))                      # what autograd does for F.linear: (N, M) @ (M, K)
    b = timeit(lambda: x.t().mm(dy).t())                  # (K, M) @ (M, N), transposed view back
    dyt, xt = dy.t().contiguous(), x.t().contiguous()
    c = timeit(lambda: dyt.mm(x))                         # pre-transposed dY (cost of the transpose excluded)
    d = timeit(lambda: torch.mm(dyt, xt.t()))
    fl = 2.0 * M * N * K
    print(f"{name:9s} dY^T@X {a:7.1f} us ({fl/a/1e9:.2f} PF/s)   (X^T@dY)^T {b:7.1f} us   pre-transposed dY {c:7.1f} us   both pre-transposed {d:7.1f} us", flush=True)
