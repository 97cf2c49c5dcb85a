"""hipBLASLt vs rocBLAS for the layer's GEMM shapes, on model-like (smooth) data."""
import json, torch, torch.nn.functional as F
dev, dt = "cuda", torch.bfloat16
M = 65536
shapes = {"in_proj": (640, 2560), "out_proj": (1280, 640), "q_proj": (640, 640), "x_proj": (1280, 72)}
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / iters * 1e3
for lib in ("cublaslt", "cublas"):
    torch.backends.cuda.preferred_blas_library(lib)
    for name, (K, N) in shapes.items():
        torch.manual_seed(0)
        x = (torch.randn(M, K, device=dev) * 0.3).to(dt); W = (torch.randn(N, K, device=dev) * K ** -0.5).to(dt)
        print(lib, name, round(timeit(lambda: F.linear(x, W)), 1), "us", flush=True)
