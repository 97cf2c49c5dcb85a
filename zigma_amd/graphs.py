"""hipGraph capture of one denoiser forward.

At small batch a ZigMa forward is ~700 short launches and host-bound; the ODE loop calls the same
forward with the same shapes num_steps times.  `GraphedForward` captures it once (torch.cuda.CUDAGraph; the HIP
kernels of libzigma_hip.so launch on torch's current stream, so they are captured like any other node) and replays
it per function evaluation.  Drop-in for `model.forward` in `Sampler.sample_ode(...)(z, model_fn, **kw)`.
"""
import torch


class GraphedForward:
    def __init__(self, model, x, t, y=None, warmup=2):
        self.model = model
        self.sx, self.st = x.clone(), t.clone()
        self.sy = None if y is None else y.clone()
        self.key = self._key(x, t, y)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            for _ in range(warmup):                       # library handles / autotune / lazy caches off the graph
                model(self.sx, self.st, self.sy)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph), torch.no_grad():
            self.out = model(self.sx, self.st, self.sy)

    @staticmethod
    def _key(x, t, y):
        return (tuple(x.shape), x.dtype, tuple(t.shape), t.dtype, None if y is None else (tuple(y.shape), y.dtype))

    def __call__(self, x, t, y=None):
        if self._key(x, t, y) != self.key:
            raise RuntimeError("GraphedForward: shapes / dtypes differ from the captured ones")
        self.sx.copy_(x)
        self.st.copy_(t)
        if y is not None:
            self.sy.copy_(y)
        self.graph.replay()
        return self.out.clone()


class DualStreamForward:
    """One denoiser forward captured as ONE hipGraph whose two halves of the batch run on two HIP streams, the second
    with a head-start delay for the first (a few hundred microseconds), so that the VALU-bound scan of one half
    overlaps the MFMA-bound GEMMs / HBM-bound norms of the other.  Same outputs as `model.forward` (samples are independent; the
    library GEMMs may pick another tile for the half-size M).  Measured on MI355X, README model, B=64: 23.9 -> 22.5 ms (tools/overlap_probe.py)."""

    def __init__(self, model, x, t, y=None, stagger_us=300, warmup=2):
        if x.shape[0] % 2:
            raise RuntimeError("DualStreamForward needs an even batch")
        self.sx, self.st = x.clone(), t.clone()
        self.sy = None if y is None else y.clone()
        self.key = GraphedForward._key(x, t, y)
        streams = [torch.cuda.Stream(), torch.cuda.Stream()]
        halves = lambda v: (None, None) if v is None else v.chunk(2)

        def run():
            cur = torch.cuda.current_stream()
            outs = []
            for i, (s, a, b, c) in enumerate(zip(streams, halves(self.sx), halves(self.st), halves(self.sy))):
                s.wait_stream(cur)
                with torch.cuda.stream(s):
                    if i and stagger_us:
                        torch.cuda._sleep(int(stagger_us * 2400))
                    outs.append(model(a, b, c))
            for s in streams:
                cur.wait_stream(s)
            return torch.cat(outs, 0)

        with torch.no_grad():
            for _ in range(warmup):
                run()
            torch.cuda.synchronize()
            self.graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self.graph):
                self.out = run()

    def __call__(self, x, t, y=None):
        if GraphedForward._key(x, t, y) != self.key:
            raise RuntimeError("DualStreamForward: shapes / dtypes differ from the captured ones")
        self.sx.copy_(x)
        self.st.copy_(t)
        if y is not None:
            self.sy.copy_(y)
        self.graph.replay()
        return self.out.clone()
