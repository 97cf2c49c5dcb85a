import os, sys, time, json, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
F = torch.nn.functional
dev, dt = "cuda", torch.bfloat16
B, L, E, Di, R, N = 64, 1024, 640, 1280, 40, 16
x = torch.randn(B, L, E, device=dev, dtype=dt); u = torch.randn(B, L, Di, device=dev, dtype=dt)
q = torch.randn(B, L, 512, device=dev, dtype=dt)
shapes = {"in_proj": (x, torch.randn(2 * Di, E, device=dev, dtype=dt)), "out_proj": (u, torch.randn(E, Di, device=dev, dtype=dt)),
          "x_proj": (u, torch.randn(R + 2 * N, Di, device=dev, dtype=dt)), "to_q": (x, torch.randn(512, E, device=dev, dtype=dt)),
          "to_out": (q, torch.randn(E, 512, device=dev, dtype=dt))}
def timeit(fn, iters=30):
    for _ in range(5): fn()
    torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) / iters * 1e3
base = {k: timeit(lambda a=a, w=w: F.linear(a, w)) for k, (a, w) in shapes.items()}
t0 = time.time()
torch.cuda.tunable.enable(True); torch.cuda.tunable.set_max_tuning_duration(200); torch.cuda.tunable.set_max_tuning_iterations(20)
tuned = {k: timeit(lambda a=a, w=w: F.linear(a, w)) for k, (a, w) in shapes.items()}
print(json.dumps(dict(base_us=base, tuned_us=tuned, tuning_s=time.time() - t0)))
