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
