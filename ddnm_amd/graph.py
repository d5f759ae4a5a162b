"""hipGraph replay of a noise-predictor forward, with the batch split over two concurrent graph branches.

Why (measured on MI355X, ADM UNet fp16 path, B = 4): a forward is ~680 kernel launches of 5-400 us; eager, the host
needs ~15 us of Python / ctypes per launch, which is the same order as the GPU time once the convolutions run at
> 1 PFLOP/s.  And every launch is a bulk-synchronous phase: all 256 CUs run the MFMA main loop together (HBM idle)
and then all write their output tiles together (matrix pipes idle); the low-resolution layers cannot fill 256 CUs
at all.  Images of a batch are independent, so the forward is captured ONCE per batch shape as
a graph with two branches -- the two halves of the batch on two streams -- and replayed: the hardware interleaves
workgroups of both branches, so one half's epilogues / small launches overlap the other half's main loops.

Only plumbing lives here (torch.cuda.CUDAGraph = hipGraph on ROCm, streams, static I/O buffers); the kernels are the
same C-ABI launches, enqueued on `torch.cuda.current_stream()` and therefore captured like any other stream work.
"""
import torch


class GraphedForward:
    def __init__(self, fn, two_streams=True):
        self.fn, self.two_streams = fn, two_streams
        self.entries = {}
        self.side = None

    def reset(self):
        self.entries = {}

    def _split_forward(self, x, t, y):
        B = x.shape[0]
        if not self.two_streams or B < 2 or B % 2:
            return self.fn(x, t, y)
        if self.side is None:
            self.side = [torch.cuda.Stream(device=x.device), torch.cuda.Stream(device=x.device)]
        cur = torch.cuda.current_stream()
        h = B // 2
        outs = []
        for i, s in enumerate(self.side):
            s.wait_stream(cur)                       # fork
            with torch.cuda.stream(s):
                sl = slice(i * h, (i + 1) * h)
                outs.append(self.fn(x[sl], t[sl], None if y is None else y[sl]))
        for s in self.side:
            cur.wait_stream(s)                       # join
        return torch.cat(outs, 0)

    def __call__(self, x, t, y=None):
        key = (tuple(x.shape), x.device, y is not None)
        ent = self.entries.get(key)
        if ent is None:
            sx = x.detach().float().contiguous().clone()
            st = t.detach().to(device=x.device, dtype=torch.float32).contiguous().clone()
            sy = None if y is None else y.detach().to(device=x.device, dtype=torch.int64).contiguous().clone()
            warm = torch.cuda.Stream(device=x.device)
            warm.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(warm):            # settles workspaces / allocator pools before the capture
                for _ in range(2):
                    self._split_forward(sx, st, sy)
            torch.cuda.current_stream().wait_stream(warm)
            torch.cuda.synchronize()
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                out = self._split_forward(sx, st, sy)
            ent = (g, sx, st, sy, out)
            self.entries[key] = ent
        g, sx, st, sy, out = ent
        sx.copy_(x)
        st.copy_(t)
        if sy is not None:
            sy.copy_(y)
        g.replay()
        return out.clone()             # the static output buffer is overwritten by the next replay
