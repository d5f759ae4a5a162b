"""DDNM reverse-diffusion loop on MI355X.

Drop-in for `functions/svd_ddnm.py::ddnm_diffusion` (:19-78): same name, same
positional arguments `(x, model, b, eta, A_funcs, y, cls_fn, classes, config)`,
same return value `([x_0], [x0_pred_last])`.

What differs from the reference is everything the reference does per step on the
host: it rebuilds alpha-bar with a cumprod every step (:10-13,43-44), ping-pongs
every intermediate through CPU memory (:45,67-68,72,76 -- two implicit device
syncs per step) and launches ~60 tiny ATen kernels for the projection.  Here
alpha-bar is tabulated once on the host, all state stays in HBM, and a step is:
UNet forward (ddnm_amd.guided_diffusion.models) + ONE fused HIP kernel (or a
short fixed chain for the SVD-free forms of sr_bicubic / cs_walshhadamard).

Reference quirks kept on purpose: float timesteps
990, 980, ...; classifier guidance evaluated on the initial noise `x` with the
hard-coded class 951 (:7,49-52); time travel re-noises the UN-projected x0
prediction (:72-74).

`noise` (optional) is an explicit tape: tensor [n_iters, B, 3, H, W] or a list of
tensors, consumed one per loop iteration (the parity path).  Without it the
N(0, I) draw of the reference's `torch.randn_like(x)` (svd_ddnm.py:65,74) happens
INSIDE the step kernels (`ops.PhiloxNoise`: Philox4x32-10 + Box-Muller, counter =
(element, iteration, global image index), key = the torch seed and a per-call
counter): no noise tensor is written and read back, and no ATen kernel runs in
the loop.  `noise=ops.PhiloxNoise(seed, image_base)` pins seed and the global
index of the batch's first image (sharded runs); DDNM_NOISE=torch restores the
ATen draw from the device generator.

`return_cpu` (default True) matches the reference, which hands back CPU tensors
(`xs[-1]`, `x0_preds[-1]` were `.to('cpu')`, svd_ddnm.py:67-68,76-78); the runner, the
benchmark and the parity tests pass False and keep the result in HBM.  `record(k, name, tensor)`
is an optional probe called with the device tensors `x0_t` / `xt_next` after iteration k.
"""
import contextlib

import torch

from .. import ops
from .svd_operators import A_functions

class_num = 951      # svd_ddnm.py:7


def compute_alpha(beta, t):
    """alpha_bar_t as in svd_ddnm.py:10-13 (fp32 cumprod over [0, beta]); t = -1 -> 1."""
    beta = torch.cat([torch.zeros(1).to(beta.device), beta], dim=0)
    return (1 - beta).cumprod(dim=0).index_select(0, t + 1).view(-1, 1, 1, 1)


def get_schedule_jump(T_sampling, travel_length, travel_repeat):
    """RePaint-style jump schedule (svd_ddnm.py:167-206), with the reference's sanity checks."""
    budget = {j: travel_repeat - 1 for j in range(0, T_sampling - travel_length, travel_length)}
    t, ts = T_sampling, []
    while t >= 1:
        t -= 1
        ts.append(t)
        if budget.get(t, 0) > 0:
            budget[t] -= 1
            for _ in range(travel_length):
                t += 1
                ts.append(t)
    ts.append(-1)
    assert ts[0] > ts[1], (ts[0], ts[1])
    assert ts[-1] == -1, ts[-1]
    for a, b in zip(ts[:-1], ts[1:]):
        assert abs(a - b) == 1, (a, b)
    for v in ts:
        assert -1 <= v <= T_sampling, (v, T_sampling)
    return ts


class _AlphaTable:
    """alpha-bar for t in [-1, T) tabulated once on the host in fp32 (same arithmetic as compute_alpha)."""

    def __init__(self, b):
        bc = b.detach().float().cpu()
        full = torch.cat([torch.zeros(1), bc], dim=0)
        self.ab = (1 - full).cumprod(dim=0)

    def __call__(self, t):
        return self.ab[t + 1]


def _guided_eps(et, grad, coef):
    """eps[:, :3] - sqrt(1 - abar_t) * cls_fn(x, t, y)   (svd_ddnm.py:51-52; note the guidance is evaluated on the
    INITIAL noise x, not on x_t -- reference quirk kept)."""
    import ctypes  # noqa: F401
    from .. import _lib
    n = et.shape[0]
    chw = 3 * et.shape[2] * et.shape[3]
    out = torch.empty(n, 3, et.shape[2], et.shape[3], dtype=torch.float32, device=et.device)
    grad = grad.float().contiguous()
    _lib.check(_lib.lib().ddnm_axpby_strided_f32(et.data_ptr(), et.stride(0), grad.data_ptr(), out.data_ptr(), n, chw,
                                                 1.0, -coef, ops._stream()), "ddnm_axpby_strided_f32")
    return out


def _step_tables(times, skip, n, device, with_classes):
    """Per-run device constants of the loop: one row [n] of the float timestep per distinct reverse-step time (the
    reference builds `torch.ones(n) * i` every step, svd_ddnm.py:40) and the constant class vector (:50).  ONE host-to-
    device copy per run; the loop then only takes views -- no fill kernel (and no other ATen arithmetic) per step."""
    uniq = sorted({a * skip for a, c in zip(times[:-1], times[1:]) if c < a})
    table = torch.tensor([[float(v)] * n for v in uniq], dtype=torch.float32).to(device, non_blocking=False) if uniq else None
    index = {v: k for k, v in enumerate(uniq)}
    cls = torch.tensor([class_num] * n, dtype=torch.long).to(device) if with_classes else None
    return (lambda i: table[index[i]]), cls


_SIDE_STREAMS = {}


def _side_stream(device, priority):
    key = (torch.device(device).index if torch.device(device).index is not None else torch.cuda.current_device(), priority)
    st = _SIDE_STREAMS.get(key)
    if st is None:
        st = _SIDE_STREAMS[key] = torch.cuda.Stream(device=device, priority=priority)
    return st


class _GuidanceAhead:
    """Classifier guidance on its own HIP stream, ahead of the UNet.

    The reference evaluates `cls_fn(x, t, classes)` on the INITIAL noise `x` (svd_ddnm.py:49-52; quirk kept), so the
    guidance term of a step depends only on (x, t, class) -- not on x_t, i.e. not on the UNet chain.  Its ~600 launches
    (classifier forward + explicit input-gradient backward, many of them latency-bound with a handful of workgroups)
    therefore run on a second stream while the first stream runs the UNet; the main stream waits on an event right before
    it combines the two.  Since round 4 the guidance terms of G consecutive reverse steps (DDNM_CLS_GROUP, default 4) are
    evaluated in one pass over the batch [x; ...; x] with the timestep vector [t_k ... t_k, t_k+1 ... t_k+1, ...] (the
    classifier conditions per sample, so this IS the reference's computation for each of the steps): the low-resolution
    half of the classifier is a chain of launches whose time does not depend on the batch, and one chain now serves G
    steps.  Same inputs, same arithmetic; the launch plans of a batch of G n differ from those of n in summation order
    only (fp32: ~1e-7).  DDNM_CLS_GROUP=1 evaluates step by step (bit-identical to the serial order), DDNM_CLS_OVERLAP=0
    restores the serial order on the main stream."""

    def __init__(self, cls_fn, x, n, t_values, t_of=None, cls=None):
        import collections
        import os
        self.cls_fn, self.x, self.n = cls_fn, x, n
        self.t_of, self.cls = t_of, cls
        self.t_values, self.pos = t_values, 0
        self.serial = os.environ.get("DDNM_CLS_OVERLAP") == "0"
        # steps per guidance pass: DDNM_CLS_GROUP (default 4: c5 at B = 8 on one MI355X 2.92 / 3.00 / 3.03 / 3.04 images/s
        # for 1 / 2 / 3 / 4, same box; 1 or 0 = step by step, bit-identical to the serial order).  ONLY the engine's own cond_fn is grouped (`make_cond_fn` marks it): a foreign callable
        # -- e.g. the reference's torch-autograd closure -- sees exactly the reference's calls (n images, one timestep),
        # because grouping multiplies its activation memory by G and assumes a strictly per-sample classifier.
        grp = int(os.environ.get("DDNM_CLS_GROUP", "4"))
        if grp < 1 or getattr(cls_fn, "ddnm_engine", None) is None:
            grp = 1
        # ... capped so that the replicated batch stays within what one convolution launch can address (2 GiB per
        # tensor): 32 images for the fp32-tensor engines (128 channels at 256 x 256), 64 for the fp16-activation one
        cap = int(getattr(getattr(cls_fn, "ddnm_engine", None), "max_group_batch", 32))
        self.group = max(1, min(grp, len(t_values), max(1, cap // n))) if (not self.serial and n > 0) else 1
        self.queue = collections.deque()
        if self.group > 1:
            # per-run constants of the grouped evaluation, built on the main stream before the side stream forks: the
            # replicated batch, its class vector and ONE host-to-device table of the grouped timestep vectors
            G = self.group
            self.xg = torch.cat([x] * G, 0)
            rows = []
            for i in range(0, len(t_values), G):
                ts = list(t_values[i:i + G])
                ts = ts + [ts[-1]] * (G - len(ts))                   # a ragged last group uses the first len(ts) * n rows only
                rows.append([float(tv) for tv in ts for _ in range(n)])
            self.tg = torch.tensor(rows, dtype=torch.float32).to(x.device)
            self.clsg = torch.tensor([class_num] * (G * n), dtype=torch.long).to(x.device)
        if not self.serial:
            self.main = torch.cuda.current_stream()
            # high priority: the guidance pass is a long chain of small dependent launches (latency-bound), the UNet a
            # sequence of chip-filling ones -- the chain must not queue behind them, the big kernels soak up the rest
            # ONE side stream per (device, priority) for the life of the process: the convolution / GroupNorm workspaces
            # are keyed by the stream handle, a fresh stream per call would grow them by one set per batch
            # (measured on c5, one MI355X: high priority 2.85 -> 2.99 images/s for step-by-step evaluation; with four steps
            # per pass the chain has four UNet steps to hide behind and normal priority is ahead, 3.034 vs 3.011)
            self.side = _side_stream(x.device, -1 if self.group == 1 else 0)
            self.side.wait_stream(self.main)             # x (and the operator's set-up) are complete
            self._launch()

    def _launch(self):
        if self.pos >= len(self.t_values):
            return
        tv = self.t_values[self.pos]
        G = self.group
        if G > 1:
            g = min(G, len(self.t_values) - self.pos)            # steps served by this pass
            m = g * self.n
            with torch.cuda.stream(self.side):
                # (the engine's cond_fn is told that the batch is g copies of x: t-independent layers run once)
                gg = self.cls_fn(self.xg[:m], self.tg[self.pos // G][:m], self.clsg[:m], replicas=g)
                ev = torch.cuda.Event()
                ev.record(self.side)
            for j in range(g):
                self.queue.append((self.t_values[self.pos + j], gg[j * self.n:(j + 1) * self.n], ev, gg))
            self.pos += g
            return
        self.pos += 1
        with torch.cuda.stream(self.side):
            if self.t_of is not None:        # views of the per-run tables (built on the main stream before this one forked)
                t, cls = self.t_of(tv), self.cls
            else:
                t = torch.full((self.n,), float(tv), device=self.x.device, dtype=torch.float32)
                cls = torch.full((self.n,), class_num, dtype=torch.long, device=self.x.device)
            g = self.cls_fn(self.x, t, cls)
            ev = torch.cuda.Event()
            ev.record(self.side)
        self.queue.append((tv, g, ev, g))

    def grad(self, tv, t, cls):
        """Gradient of the reverse step at timestep `tv` (called AFTER the UNet forward of that step was enqueued)."""
        if self.serial:
            return self.cls_fn(self.x, t, cls)
        tq, g, ev, owner = self.queue.popleft()
        assert tq == tv, (tq, tv)
        self.main.wait_event(ev)
        owner.record_stream(self.main)                    # allocated on the side stream, consumed on the main one
        if len(self.queue) <= max(0, self.group - 1):
            # the next evaluation starts under this step's tail; a grouped pass (G times the work) is launched when the
            # FIRST term of the previous group is consumed, so it has G UNet steps to hide behind
            self._launch()
        return g

    def close(self):
        if not self.serial:
            self.main.wait_stream(self.side)


_PHILOX_CALLS = [0]


def _philox_for_call(like):
    """Key of an un-pinned run (`noise=None`): a 64-bit hash of the device generator's FULL seed (torch.manual_seed /
    torch.cuda.manual_seed_all set it, like the reference's main.py:139-143), a per-call counter -- consecutive restorations
    draw different noise -- and the rank of the calling process, so that the ranks of a sharded programmatic run do not draw
    identical noise for different images.  Callers that want rank-count-independent results pin the generator:
    `noise=ops.PhiloxNoise(seed, image_base=<global index of image 0>)`, as the runner does; DDNM_NOISE=torch = ATen draws."""
    seed = int(torch.cuda.initial_seed() if like.is_cuda else torch.initial_seed())
    _PHILOX_CALLS[0] += 1
    rank = torch.distributed.get_rank() if (torch.distributed.is_available() and torch.distributed.is_initialized()) else 0
    z = (seed * 0x9E3779B97F4A7C15 + _PHILOX_CALLS[0] * 0xBF58476D1CE4E5B9 + (rank + 1) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return ops.PhiloxNoise(z ^ (z >> 31))


def _noise_source(noise, like):
    """draw(k) -> the noise tensor of loop iteration k, or None when the step kernels draw it themselves (`.philox`)."""
    import os
    if noise is None and os.environ.get("DDNM_NOISE") == "torch":
        def draw(k):
            return torch.randn_like(like)
        draw.philox = None
        return draw
    if noise is None or isinstance(noise, ops.PhiloxNoise):
        ph = _philox_for_call(like) if noise is None else noise

        def draw(k):
            return None
        draw.philox = ph
        return draw

    def take(k):
        n = noise[k]
        if n.device != like.device or n.dtype != torch.float32 or not n.is_contiguous():
            n = n.to(device=like.device, dtype=torch.float32).contiguous()
        return n
    take.philox = None
    return take


def _finish(xt, x0_t, return_cpu):
    if return_cpu:
        return [xt.to("cpu")], [x0_t.to("cpu")]
    return [xt], [x0_t]


def ddnm_diffusion(x, model, b, eta, A_funcs, y, cls_fn=None, classes=None, config=None, noise=None, return_cpu=True,
                   record=None):
    if not x.is_cuda:
        raise RuntimeError("ddnm_amd.ddnm_diffusion runs on the GPU only (no CPU fallback); got a CPU tensor")
    skip = config.diffusion.num_diffusion_timesteps // config.time_travel.T_sampling
    n = x.size(0)
    times = get_schedule_jump(config.time_travel.T_sampling, config.time_travel.travel_length,
                              config.time_travel.travel_repeat)
    alpha = _AlphaTable(b)
    x = x.float().contiguous()
    y = y.reshape(n, -1).float().contiguous()
    draw = _noise_source(noise, x)
    fused = isinstance(A_funcs, A_functions)
    if hasattr(A_funcs, "begin_run"):
        A_funcs.begin_run(y)           # per-run constants of the operator (e.g. A^+ y); never cached across runs

    xt = x
    x0_t = torch.empty_like(x)
    bufs = [torch.empty_like(x), torch.empty_like(x)]
    have_x0 = False
    guide = None
    with torch.no_grad(), contextlib.ExitStack() as _stack:
        t_of, cls_const = _step_tables(times, skip, n, x.device, cls_fn is not None)
        if cls_fn is not None:
            guide = _GuidanceAhead(cls_fn, x, n, [a * skip for a, c in zip(times[:-1], times[1:]) if c < a], t_of, cls_const)
            _stack.callback(guide.close)         # also when the loop raises: the main stream re-joins the side stream
        for k, (i, j) in enumerate(zip(times[:-1], times[1:])):
            i, j = i * skip, j * skip
            if j < 0:
                j = -1
            at_next = alpha(j)
            out = bufs[k & 1]
            if j < i:      # reverse step
                at = alpha(i)
                t = t_of(i)
                if cls_fn is None:
                    et = model(xt, t)
                else:
                    cls = cls_const
                    eps = model(xt, t, cls)
                    et = _guided_eps(eps, guide.grad(i, t, cls), float((1 - at).sqrt()))
                if et.size(1) == 6:
                    et = et[:, :3]
                s = ops.step_scalars(at, at_next, eta)
                if draw.philox is not None:
                    draw.philox.stamp(s, k)          # the step kernel draws its own noise (noise pointer NULL)
                if fused:
                    A_funcs.ddnm_step(xt, et, draw(k), y, s, x0_t, out)
                else:          # foreign operator object: its own A / A_pinv, our elementwise kernels
                    ops.step_x0(xt, et, s, out=x0_t)
                    proj = A_funcs.A_pinv(A_funcs.A(x0_t.reshape(n, -1)) - y).reshape(*x0_t.size())
                    ops.step_combine(x0_t, proj.float().contiguous(), None, draw(k), et, s, out=out)
                have_x0 = True
            else:          # time-travel back (svd_ddnm.py:70-76)
                assert have_x0
                nz = draw(k) if draw.philox is None else draw.philox.tensor(k, x0_t)
                ops.renoise(x0_t, nz, float(at_next.sqrt()), float((1 - at_next).sqrt()), out=out)
            xt = out
            if record is not None:
                record(k, "x0_t", x0_t)
                record(k, "xt_next", xt)
    return _finish(xt, x0_t, return_cpu)


def ddnm_plus_diffusion(x, model, b, eta, A_funcs, y, sigma_y, cls_fn=None, classes=None, config=None, noise=None,
                        return_cpu=True, record=None):
    """DDNM+ for noisy measurements: drop-in for `functions/svd_ddnm.py::ddnm_plus_diffusion` (:80-164).

      x0|t  = (x_t - eps*sqrt(1-abar_t)) / sqrt(abar_t)                                  (Eq. 12)
      x0^   = x0|t - Lambda( A^+ (A x0|t - y) )                                          (Eq. 17)
      x_t-1 = sqrt(abar') x0^ + Lambda_noise( N(0,I), eps )                              (Eq. 51)

    with the operator's `Lambda` / `Lambda_noise` (spectral lambda_t and noise mixing).  Every product is a
    HIP kernel; per step: UNet forward, x0 kernel, A / A^+ kernels, Lambda, Lambda_noise, combine."""
    if not x.is_cuda:
        raise RuntimeError("ddnm_amd.ddnm_plus_diffusion runs on the GPU only (no CPU fallback)")
    skip = config.diffusion.num_diffusion_timesteps // config.time_travel.T_sampling
    n = x.size(0)
    times = get_schedule_jump(config.time_travel.T_sampling, config.time_travel.travel_length,
                              config.time_travel.travel_repeat)
    alpha = _AlphaTable(b)
    x = x.float().contiguous()
    y = y.reshape(n, -1).float().contiguous()
    draw = _noise_source(noise, x)
    from .svd_operators import _axpby
    xt = x
    x0_t = torch.empty_like(x)
    bufs = [torch.empty_like(x), torch.empty_like(x)]
    have_x0 = False
    guide = None
    with torch.no_grad(), contextlib.ExitStack() as _stack:
        t_of, cls_const = _step_tables(times, skip, n, x.device, cls_fn is not None)
        if cls_fn is not None:
            guide = _GuidanceAhead(cls_fn, x, n, [a * skip for a, c in zip(times[:-1], times[1:]) if c < a], t_of, cls_const)
            _stack.callback(guide.close)         # also when the loop raises: the main stream re-joins the side stream
        for k, (i, j) in enumerate(zip(times[:-1], times[1:])):
            i, j = i * skip, j * skip
            if j < 0:
                j = -1
            at_next = alpha(j)
            out = bufs[k & 1]
            if j < i:
                at = alpha(i)
                t = t_of(i)
                if cls_fn is None:
                    et = model(xt, t)
                else:
                    cls = cls_const
                    eps = model(xt, t, cls)
                    et = _guided_eps(eps, guide.grad(i, t, cls), float((1 - at).sqrt()))
                if et.size(1) == 6:
                    et = et[:, :3].contiguous()
                a, sigma_t = at_next.sqrt(), (1 - at_next).sqrt()
                s = ops.step_scalars(at, at_next, eta)
                ops.step_x0(xt, et, s, out=x0_t)
                resid = _axpby(A_funcs.A(x0_t), y, 1.0, -1.0)
                corr = A_funcs.Lambda(A_funcs.A_pinv(resid), a, sigma_y, sigma_t, eta).reshape(x.shape)
                eps_k = draw(k) if draw.philox is None else draw.philox.tensor(k, x0_t)
                nz = A_funcs.Lambda_noise(eps_k, a, sigma_y, sigma_t, eta, et).reshape(x.shape)
                s.c1, s.c2, s.lam = 1.0, 0.0, 1.0        # x_t-1 = sqrt(abar') (x0 - corr) + 1 * nz
                ops.step_combine(x0_t, corr, None, nz, et, s, out=out)
                have_x0 = True
            else:
                assert have_x0
                nz = draw(k) if draw.philox is None else draw.philox.tensor(k, x0_t)
                ops.renoise(x0_t, nz, float(at_next.sqrt()), float((1 - at_next).sqrt()), out=out)
            xt = out
            if record is not None:
                record(k, "x0_t", x0_t)
                record(k, "xt_next", xt)
    return _finish(xt, x0_t, return_cpu)
