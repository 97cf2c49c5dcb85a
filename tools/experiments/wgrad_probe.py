import ctypes as C, os, torch
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libwgrad.so"))
lib.wgrad.argtypes = [C.c_void_p] * 4 + [C.c_int] * 3 + [C.c_int64, C.c_int64, C.c_int, C.c_void_p]
dev, dt = "cuda", torch.bfloat16
def timeit(fn, iters=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / iters * 1e3
M = 65536
for name, (N, K, splits) in {"to_out": (640, 512, 64), "to_q": (512, 640, 64), "out_proj": (640, 1280, 32), "in_proj": (2560, 640, 16)}.items():
    torch.manual_seed(0)
    dy = (torch.randn(M, N, device=dev) * 0.1).to(dt); x = (torch.randn(M, K, device=dev) * 0.3).to(dt)
    part = torch.empty(splits, N, K, device=dev); dw = torch.empty(N, K, device=dev)
    run = lambda: lib.wgrad(dy.data_ptr(), x.data_ptr(), part.data_ptr(), dw.data_ptr(), M, N, K, dy.stride(0), x.stride(0), splits, torch.cuda.current_stream().cuda_stream)
    rc = run(); torch.cuda.synchronize()
    ref = dy.float().t() @ x.float()
    err = ((dw - ref).norm() / ref.norm()).item()
    t1, t2 = timeit(run), timeit(lambda: dy.t().mm(x))
    fl = 2.0 * M * N * K
    print(f"{name:9s} rc={rc} wgrad {t1:7.1f} us ({fl / t1 / 1e9:.2f} PF/s)   library {t2:7.1f} us ({fl / t2 / 1e9:.2f} PF/s)   rel err {err:.2e}", flush=True)
