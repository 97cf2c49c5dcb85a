import ctypes as C, os, torch
lib = C.CDLL(os.path.join(os.path.dirname(os.path.abspath(__file__)), "libstore_patterns.so"))
lib.run.argtypes = [C.c_int, C.c_void_p, C.c_void_p]
o = torch.empty(65536, 1280, device="cuda", dtype=torch.bfloat16)
def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / iters * 1e3
names = ["P0 linear 16 B", "P1 dt_proj shipped (32 tok x 64 ch, 4-B stores)", "P2 wide (16 tok x 128 ch, 16-B stores)", "P3 whole rows per wave"]
for w, nm in enumerate(names):
    t = timeit(lambda: lib.run(w, o.data_ptr(), torch.cuda.current_stream().cuda_stream))
    print(f"{nm:55s} {t:6.1f} us  {o.numel() * 2 / t / 1e6:.2f} TB/s")
