import sys, time, torch
sys.path.insert(0, "/root/repo")
import bench
wl = bench.WORKLOADS["readme_text_b64"]
m = bench.build_model(wl["model"], "cuda", torch.bfloat16)
x, t, y = bench.make_inputs(wl, 64, "cuda", 1)
with torch.no_grad():
    for _ in range(3): m(x, t, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): m(x, t, y)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print("eager: host issue ms/fwd", t_issue / 5 * 1e3, "wall ms/fwd", t_all / 5 * 1e3)
    # graph
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(2): m(x, t, y)
    torch.cuda.current_stream().wait_stream(s)
    with torch.cuda.graph(g):
        out = m(x, t, y)
    torch.cuda.synchronize()
    ref = m(x, t, y)
    g.replay(); torch.cuda.synchronize()
    print("graph vs eager max diff", (out - ref).abs().max().item())
    t0 = time.perf_counter()
    for _ in range(10): g.replay()
    torch.cuda.synchronize()
    print("graph: wall ms/fwd", (time.perf_counter() - t0) / 10 * 1e3)
