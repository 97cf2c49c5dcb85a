import ctypes as C, os, torch, torch.nn.functional as F
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libgemm_rowstream.so"))
lib.gemm_rowstream.argtypes = [C.c_void_p] * 4 + [C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_void_p]
dev, dt = "cuda", torch.bfloat16
def timeit(fn, iters=20):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / iters * 1e3
M = 65536
for K in (512, 1280):
    torch.manual_seed(0)
    a = (torch.randn(M, K, device=dev) * 0.3).to(dt); w = (torch.randn(640, K, device=dev) * K ** -0.5).to(dt)
    c = torch.empty(M, 640, device=dev, dtype=dt)
    run = lambda: lib.gemm_rowstream(a.data_ptr(), w.data_ptr(), None, c.data_ptr(), M, K, a.stride(0), w.stride(0), c.stride(0), torch.cuda.current_stream().cuda_stream)
    run(); torch.cuda.synchronize()
    ref = F.linear(a, w)
    err = ((c.float() - ref.float()).norm() / ref.float().norm()).item()
    t1, t2 = timeit(run), timeit(lambda: F.linear(a, w))
    fl = 2.0 * M * K * 640
    print(f"K={K}: rowstream {t1:.1f} us ({fl / t1 / 1e9:.2f} PF/s)  hipBLASLt {t2:.1f} us ({fl / t2 / 1e9:.2f} PF/s)  rel err vs library {err:.2e}")
