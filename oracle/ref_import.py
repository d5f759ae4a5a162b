"""Import the real reference (wyhuai/DDNM @ /root/reference) on CPU.

TEST INFRASTRUCTURE ONLY.  Available only where ``/root/reference`` exists
(the build container); the GPU box never has it, so nothing executed by the
``-m gpu`` tests, ``smoke()`` or ``bench.py`` may call into this module.

Shims:
  (i)   stub modules for cv2 / torchvision / tensorboard, which the reference
        imports but never uses on the hot path
        (functions/svd_operators.py:2, functions/svd_ddnm.py:3-4, main.py:10);
  (ii)  ``Tensor.to('cuda')`` / ``torch.device('cuda')`` mapped to CPU, because the
        sampler hard-codes 'cuda' (functions/svd_ddnm.py:45,49,72);
  (iii) a noise tape replacing ``torch.randn_like`` so that reference and engine
        consume identical Gaussian draws (functions/svd_ddnm.py:65,74).
"""
import contextlib
import os
import sys
import types

import torch

REF_ROOT = os.environ.get("DDNM_REFERENCE_ROOT", "/root/reference")


def available() -> bool:
    return os.path.isdir(os.path.join(REF_ROOT, "functions"))


def _stub(name, **attrs):
    if name in sys.modules:
        return sys.modules[name]
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


_loaded = {}


def load():
    """Returns a namespace with the reference modules of the hot path."""
    if _loaded:
        return _loaded["ns"]
    if not available():
        raise RuntimeError(f"reference tree not found at {REF_ROOT}")
    _stub("cv2")
    tv = _stub("torchvision")
    tv.utils = _stub("torchvision.utils", save_image=lambda *a, **k: None)
    tv.transforms = _stub("torchvision.transforms")
    tv.transforms.functional = _stub("torchvision.transforms.functional")
    tv.datasets = _stub("torchvision.datasets")
    tv.datasets.utils = _stub("torchvision.datasets.utils")
    _stub("tensorboard")
    if "torch.utils.tensorboard" not in sys.modules:
        _stub("torch.utils.tensorboard", SummaryWriter=object)
    # the reference has top-level packages called `functions`, `datasets`,
    # `guided_diffusion`; import them by path with REF_ROOT in front.
    sys.path.insert(0, REF_ROOT)
    try:
        import importlib

        ns = types.SimpleNamespace()
        ns.svd_operators = importlib.import_module("functions.svd_operators")
        ns.svd_ddnm = importlib.import_module("functions.svd_ddnm")
        ns.models = importlib.import_module("guided_diffusion.models")
        ns.unet = importlib.import_module("guided_diffusion.unet")
        ns.script_util = importlib.import_module("guided_diffusion.script_util")
        ns.nn = importlib.import_module("guided_diffusion.nn")
    finally:
        sys.path.remove(REF_ROOT)
    _loaded["ns"] = ns
    return ns


@contextlib.contextmanager
def cuda_is_cpu():
    """Shim (ii): make `.to('cuda')` and `device=torch.device('cuda')` no-ops on CPU."""
    orig_to = torch.Tensor.to
    orig_ones = torch.ones

    def _is_cuda(d):
        return (isinstance(d, str) and d.startswith("cuda")) or (
            isinstance(d, torch.device) and d.type == "cuda")

    def to(self, *args, **kwargs):
        args = tuple("cpu" if _is_cuda(a) else a for a in args)
        if "device" in kwargs and _is_cuda(kwargs["device"]):
            kwargs["device"] = "cpu"
        return orig_to(self, *args, **kwargs)

    def ones(*args, **kwargs):
        if "device" in kwargs and _is_cuda(kwargs["device"]):
            kwargs["device"] = "cpu"
        return orig_ones(*args, **kwargs)

    torch.Tensor.to = to
    torch.ones = ones
    try:
        yield
    finally:
        torch.Tensor.to = orig_to
        torch.ones = orig_ones


@contextlib.contextmanager
def noise_tape(tape, on_call=None):
    """Shim (iii): every `torch.randn_like(x)` pops the next tensor of `tape`.

    `tape` is a list of tensors shaped like x; an exhausted tape raises.  `on_call(k, x)` (optional) sees the
    argument of the k-th call -- in `ddnm_diffusion` that is the current x0|t (svd_ddnm.py:65,74).
    """
    it = iter(tape)
    orig = torch.randn_like
    count = [0]

    def randn_like(x, *a, **k):
        try:
            n = next(it)
        except StopIteration:
            raise RuntimeError("noise tape exhausted")
        assert n.shape == x.shape, (n.shape, x.shape)
        if on_call is not None:
            on_call(count[0], x)
        count[0] += 1
        return n.to(dtype=x.dtype, device=x.device)

    torch.randn_like = randn_like
    try:
        yield
    finally:
        torch.randn_like = orig
