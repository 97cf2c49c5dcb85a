"""Does running the per-GPU batch as two half-batches on two HIP streams overlap the VALU-bound scan with the MFMA-bound
GEMMs / HBM-bound norms of the other half?  (stagger = head start of stream 0, in microseconds)"""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import WORKLOADS, build_model, make_inputs
dev = torch.device("cuda", 0)
wl = WORKLOADS["readme_text_b64"]
m = build_model(wl["model"], dev, torch.bfloat16)
x, t, y = make_inputs(wl, 64, dev, 0)
def timeit(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n * 1e3
with torch.no_grad():
    print("single stream B=64", round(timeit(lambda: m(x, t, y)), 2), "ms")
    parts = 2
    streams = [torch.cuda.Stream() for _ in range(parts)]
    xs, ts, ys = x.chunk(parts), t.chunk(parts), y.chunk(parts)
    for stagger_us in (0, 200, 400, 650, 900):
        def multi():
            cur = torch.cuda.current_stream()
            outs = []
            for i, (s, a, b, c) in enumerate(zip(streams, xs, ts, ys)):
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    if i and stagger_us: torch.cuda._sleep(int(stagger_us * 2400))
                    outs.append(m(a, b, c))
            for s in streams: cur.wait_stream(s)
            return outs
        g = torch.cuda.CUDAGraph()
        multi(); torch.cuda.synchronize()
        with torch.cuda.graph(g):
            outs = multi()
        print(f"2 streams x B=32 graphed, stagger {stagger_us} us:", round(timeit(g.replay), 2), "ms", flush=True)
