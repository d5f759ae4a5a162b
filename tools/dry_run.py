#!/usr/bin/env python
"""Host-only dry run of an engine path (development tool, no GPU): the real shared library is loaded, its PLANNING entry
points (`*_supported`, `*_stats_tiles`, `*_workspace_floats`, `ddnm_gn_nchunk`, ...) run as usual -- they are host code --
and every LAUNCHING entry point (last argument = stream) is replaced by a stub that returns 0.  Tensors live on the CPU
and hold garbage; what is exercised is the Python plumbing: shapes, dtypes, descriptor marshalling, workspace sizes,
dictionary keys.  Usage:  python tools/dry_run.py [classifier|adm|celeba] ...
"""
import ctypes
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ddnm_amd import _lib, ops  # noqa: E402

HOST_ONLY = ("_supported", "_stats_tiles", "_workspace_floats", "_nchunk", "_tile_n", "_fuses_skip", "_fuses_fin",
             "_act_scale", "ddnm_version", "ddnm_error_string", "ddnm_build_digest", "ddnm_sizeof")


class DryLib:
    def __init__(self, real):
        self._real = real
        self.calls = {}

    def __getattr__(self, name):
        fn = getattr(self._real, name)
        if any(name.endswith(s) or name == s for s in HOST_ONLY):
            return fn

        def stub(*args):
            self.calls[name] = self.calls.get(name, 0) + 1
            res, argtypes = _lib.PROTOTYPES[name]
            assert len(args) == len(argtypes), f"{name}: {len(args)} arguments for {len(argtypes)} parameters"
            for a, t in zip(args, argtypes):      # what ctypes would reject at call time
                if t in (ctypes.c_int32, ctypes.c_int64):
                    assert isinstance(a, int), f"{name}: {a!r} passed for an integer parameter"
                elif t is ctypes.c_float:
                    assert isinstance(a, (int, float)), f"{name}: {a!r} passed for a float parameter"
            return 0
        return stub


def install():
    real = _lib.lib()
    dry = DryLib(real)
    _lib._lib = dry
    ops._stream = lambda: 0
    ok = lambda t, name: t          # noqa: E731
    f32c, f16c = ops._f32c, ops._f16c

    def f32(t, name):
        assert t.dtype == torch.float32 and t.is_contiguous(), (name, t.dtype, t.is_contiguous())
        return t

    def f16(t, name):
        assert t.dtype == torch.float16 and t.is_contiguous(), (name, t.dtype, t.is_contiguous())
        return t
    ops._f32c, ops._f16c = f32, f16
    del ok, f32c, f16c
    return dry


def classifier(B=2, size=256, kind="h16"):
    from ddnm_amd.guided_diffusion.classifier import classifier_defaults, create_classifier, make_cond_fn
    kw = classifier_defaults()
    kw["image_size"] = size
    clf = create_classifier(**kw)
    clf.device = torch.device("cpu")
    g = torch.Generator().manual_seed(1)
    clf.load_state_dict({k: torch.randn(v, generator=g) * 0.05 for k, v in clf.state_dict_shapes().items()})
    if kind != "fp32":
        os.environ["DDNM_CLS_GEN1"] = "1" if kind == "gen1" else "0"
        clf.convert_to_fp16()
    x = torch.randn(B, 3, size, size)
    t = torch.full((B,), 500.0)
    y = torch.full((B,), 951, dtype=torch.long)
    logits = clf(x, t)
    assert logits.shape == (B, 1000)
    grad = make_cond_fn(clf, 1.0)(x, t, y)
    assert grad.shape == x.shape and grad.dtype == torch.float32, (grad.shape, grad.dtype)
    return clf


if __name__ == "__main__":
    dry = install()
    what = sys.argv[1] if len(sys.argv) > 1 else "classifier"
    if what == "classifier":
        for kind in ("h16", "gen1", "fp32"):
            for size, B in ((256, 2), (64, 3), (256, 32)):
                dry.calls.clear()
                classifier(B=B, size=size, kind=kind)
                n = sum(dry.calls.values())
                print(f"classifier {kind} {size}px B={B}: {n} launches; " +
                      ", ".join(f"{k.replace('ddnm_', '')} x{v}" for k, v in sorted(dry.calls.items(), key=lambda kv: -kv[1])[:8]))
