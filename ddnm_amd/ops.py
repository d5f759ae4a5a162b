"""Tensor-level wrappers over the C ABI (ddnm_amd/_lib.py).

PyTorch-ROCm is used here only as the owner of device memory and of the HIP
stream; every computation below is a hand-written HIP kernel from
ddnm_amd/csrc.  Nothing in this module falls back to torch ops.
"""
import ctypes

import torch

from . import _lib
from ._lib import Conv16Desc, ConvDesc, GemmDesc, StepScalars, check


def _p(t):
    return None if t is None else t.data_ptr()


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _f32c(t, name):
    if t.dtype != torch.float32 or not t.is_contiguous() or not t.is_cuda:
        raise ValueError(f"{name}: expected a contiguous float32 device tensor, got {t.dtype} "
                         f"contiguous={t.is_contiguous()} device={t.device}")
    return t


# ----------------------------------------------------------------------------- kernel timing hook
class KernelTimer:
    """Optional per-launch timing of the convolution kernel with HIP events on the launch stream
    (used by bench.py for the roofline figure; off in normal operation)."""

    def __init__(self):
        self.records = []          # (variant, flops, start_event, end_event)
        self.shapes = []           # per record: (B, Ho, Wo, Cin, Cout, k, stride, ups, skipC, gn, res)

    def summary(self):
        torch.cuda.synchronize()
        out = {}
        for variant, flops, e0, e1 in self.records:
            r = out.setdefault(variant, {"launches": 0, "flops": 0.0, "ms": 0.0})
            r["launches"] += 1
            r["flops"] += flops
            r["ms"] += e0.elapsed_time(e1)
        return out


_timer = None


def set_kernel_timer(t):
    global _timer
    _timer = t


def max_launch_batch(per_image_bytes, margin=2):
    """Largest batch whose biggest activation stays `margin` times below the 2 GiB one convolution launch can address
    (32-bit byte offsets of the raw buffer descriptors, csrc/conv_common.h::conv_sizes_addressable): the models split a
    larger batch into chunks of this size INSIDE forward() instead of refusing it (the reference: "you may increase the
    batch_size to accelerate evaluation", README.md:89).  Margin 2 = half of the addressable limit (32 images for the celeba
    `Model` and the fp16-activation ADM UNet at 256 x 256, 16 for the fp32 ADM UNet): the 32-image BASELINE configurations run
    as ONE launch per layer (round 5's margin of 4 split them into two chunks plus a concat copy)."""
    return max(1, (1 << 31) // int(per_image_bytes) // margin)


# ----------------------------------------------------------------------------- convolution
CONV_COUT_ALIGN = 128


_conv_ws = {}


def _conv_workspace(device, floats):
    """Split-K scratch, one buffer per (device, stream), grown on demand: stream order makes reuse safe within a
    stream, and the two half-batch streams of a captured forward (ddnm_amd/graph.py) must not share one."""
    key = (device, _stream())
    ws = _conv_ws.get(key)
    if ws is None or ws.numel() < floats:
        ws = torch.empty(int(floats), dtype=torch.float32, device=device)
        _conv_ws[key] = ws
    return ws


def pack_conv_weight(w, cin_pad=None):
    """OIHW [Cout,Cin,k,k] -> (O, ky, kx, I) with Cout padded to 128 and Cin to `cin_pad`."""
    cout, cin, kh, kw = w.shape
    cin_pad = cin if cin_pad is None else cin_pad
    cout_pad = (cout + CONV_COUT_ALIGN - 1) // CONV_COUT_ALIGN * CONV_COUT_ALIGN
    out = torch.zeros(cout_pad, kh * kw, cin_pad, dtype=torch.float32, device=w.device)
    out[:cout, :, :cin] = w.float().permute(0, 2, 3, 1).reshape(cout, kh * kw, cin)
    return out.contiguous()


import os as _os
_NO_FUSED_GN = False      # module attribute (A/B profiling from a script): convolutions emit no GroupNorm partials
FORCE_ONE_TILE = False    # module attribute (A/B timing from a script): every split-fp16 3x3 launch on the one-tile kernel


class Act:
    """An NHWC activation plus, when its producer could emit them, the GroupNorm partials of it
    (per-(M tile, channel) sum / sum of squares written by the convolution epilogue)."""
    __slots__ = ("t", "stats", "tiles", "gn")

    def __init__(self, t, stats=None, tiles=0, gn=None):
        self.t, self.stats, self.tiles = t, stats, tiles
        # (scale, shift, name): the affine of the consumer GroupNorm `name`, already finalized by the producing launch
        # (split-K reduction pass, ddnm_conv16_desc::fin_*); valid until the next finalize reuses the workspace
        self.gn = gn


def conv_fuses_skip(B, H, W, cin, cout):
    """True when a 3x3 / stride-1 conv of this shape runs on the halo kernel, i.e. can take a fused shortcut."""
    d = ConvDesc()
    d.B, d.Hin, d.Win, d.C0, d.C1, d.Cout = B, H, W, cin, 0, cout
    d.ksize, d.stride, d.pad, d.Ho, d.Wo = 3, 1, 1, H, W
    return _lib.lib().ddnm_conv2d_f32_fuses_skip(ctypes.byref(d)) == 1


def pack_skip_weight(w, f16=False):
    """1x1 shortcut weights [Cout, Cin, 1, 1] -> [Cout_pad][Cin] (fp32, or fp16 for the fp16-operand kernel)."""
    p = pack_conv_weight(w).reshape(-1, w.shape[1]).contiguous()
    return p.to(torch.float16) if f16 else p


def pack_conv_weight_f16(w, cin_pad=None):
    """OIHW fp32/fp16 -> (O, ky, kx, I) IEEE fp16, Cout padded to 128 (operands of ddnm_conv3x3_f16_f32)."""
    return pack_conv_weight(w, cin_pad=cin_pad).to(torch.float16).contiguous()


def s16_weight_scale(*ws):
    """The power of two 2^s that brings max|W| of a launch's weight tensors (3x3 kernel + fused shortcut: they share
    one accumulator) into [2^13, 2^14): fp16 `hi` parts stay far from 65504 and the `lo` parts of all but negligible
    weights are normal fp16 numbers (include/ddnm_hip.h::ddnm_conv3x3_s16_f32)."""
    import math
    m = max(float(w.abs().max()) for w in ws)
    if not (m > 0.0) or not math.isfinite(m):
        return 1.0
    return 2.0 ** (13 - math.floor(math.log2(m)))


def pack_conv_weight_s16(w, scale):
    """OIHW fp32 -> the split packing of ddnm_conv3x3_s16_f32: (O, ky, kx, I) with Cout padded to 128, every 32-channel
    chunk stored as 32 hi halfs then 32 lo halfs of scale * W (hi = rn16, lo = rn16(residual)); returned as an fp16
    tensor [Cout_pad][taps][Cin/32][2][32] (= the byte size of the fp32 packing)."""
    p = pack_conv_weight(w) * float(scale)                      # [Cout_pad][taps][Cin], power-of-two scaling: exact
    cp, taps, cin = p.shape
    if cin % 32:
        raise ValueError("split packing needs Cin % 32 == 0")
    hi = p.to(torch.float16)
    lo = (p - hi.float()).to(torch.float16)
    hi, lo = hi.view(cp, taps, cin // 32, 1, 32), lo.view(cp, taps, cin // 32, 1, 32)
    return torch.cat([hi, lo], 3).contiguous()


def conv_runs_s16(B, H, W, cin, cout, ups=False):
    """True when a 3x3 / stride-1 conv of this OUTPUT shape takes the split-fp16 kernel (given split-packed weights)."""
    d = ConvDesc()
    d.B, d.Hin, d.Win, d.C0, d.C1, d.Cout = B, H, W, cin, 0, cout
    d.ksize, d.stride, d.pad, d.Ho, d.Wo = 3, 1, 1, H, W
    return _lib.lib().ddnm_conv3x3_s16_supported(ctypes.byref(d)) == 1


def conv_runs_s16_gather(B, H, W, cin, cout, ksize=3, stride=1):
    """True when a conv of this INPUT shape takes the gather form of the split-fp16 arithmetic (no fused shortcut)."""
    if _NO_S16_GATHER:
        return False
    d = ConvDesc()
    d.B, d.Hin, d.Win, d.C0, d.C1, d.Cout = B, H, W, cin, 0, cout
    d.ksize, d.stride, d.pad, d.Ho, d.Wo = ksize, stride, (ksize // 2 if stride == 1 else 0), H // stride, W // stride
    return _lib.lib().ddnm_conv_gather_s16_supported(ctypes.byref(d)) == 1


_S16_ACT_SCALE = None
_NO_S16_GATHER = False      # module attribute (A/B from a script): gather-form launches stay on the fp32 MFMA kernel


def _s16_act_scale():
    global _S16_ACT_SCALE
    if _S16_ACT_SCALE is None:
        _S16_ACT_SCALE = float(_lib.lib().ddnm_conv3x3_s16_act_scale())
    return _S16_ACT_SCALE


def conv_runs_f16(B, H, W, cin, cout, ksize=3):
    """True when a stride-1 conv of this shape takes the fp16-operand kernel (given packed fp16 weights)."""
    d = ConvDesc()
    d.B, d.Hin, d.Win, d.C0, d.C1, d.Cout = B, H, W, cin, 0, cout
    d.ksize, d.stride, d.pad, d.Ho, d.Wo = ksize, 1, ksize // 2, H, W
    L = _lib.lib()
    return (L.ddnm_conv3x3_f16_supported if ksize == 3 else L.ddnm_conv1x1_f16_supported)(ctypes.byref(d)) == 1


_F16_PREPASS_MIN_COUT = 256
_f16_scratch_buf = {}


def _f16_scratch(device, numel, slot=0):
    """fp16 activation scratch of the GroupNorm pre-pass (slot 0) / the im2col matrix (slot 1); stream order makes
    reuse safe."""
    key = (device, slot, _stream())
    buf = _f16_scratch_buf.get(key)
    if buf is None or buf.numel() < numel:
        buf = torch.empty(numel, dtype=torch.float16, device=device)
        _f16_scratch_buf[key] = buf
    return buf


AMAX_N = 32          # include/ddnm_hip.h::DDNM_AMAX_N
_S16_CHECK = False      # module attribute (debugging): assert finiteness after every split-fp16 launch (synchronises)


def amax_bound(a0, a1=None):
    """[B][AMAX_N] per-image upper bounds of |concat(a0, a1)| for the operand-range guard of the split-fp16 kernels
    (ddnm_conv_desc::amax_in): from the producers' GroupNorm partials when the `Act`s carry them (a few KB ... 2 MB read),
    otherwise from the tensors themselves.  One small launch, no host synchronisation."""
    def src(a):
        if a is None:
            return None, 0, 0
        if isinstance(a, Act) and a.stats is not None:
            B = a.t.shape[0]
            return a.stats, a.stats.numel() // B, 1
        t = a.t if isinstance(a, Act) else a
        return _f32c(t, "amax source"), t.numel() // t.shape[0], 0
    t0 = a0.t if isinstance(a0, Act) else a0
    B = t0.shape[0]
    p0, n0, k0 = src(a0)
    p1, n1, k1 = src(a1)
    out = torch.empty(B * AMAX_N, dtype=torch.float32, device=t0.device)
    check(_lib.lib().ddnm_amax_bound_f32(_p(p0), n0, k0, _p(p1), n1, k1, _p(out), B, _stream()), "ddnm_amax_bound_f32")
    return out


def conv2d(src0, weight, cout, ksize, *, src1=None, bias=None, badd=None, badd_stride=0, res=None, res_ups=False,
           gn=None, gn_silu=True, stride=1, pad=None, ups=False, out=None, out_nchw=False, out_hw=None, tile=0,
           emit_stats=False, weight_f16=None, skip=None, skip_weight=None, skip_weight_f16=None, weight_s16=None,
           raw_amax=None, one_tile=False):
    """NHWC implicit-GEMM convolution; see include/ddnm_hip.h::ddnm_conv_desc.
    one_tile=True: the split-fp16 3x3 launch runs the one-tile-per-workgroup kernel where the persistent form would apply
    (same results bit for bit; A/B timing and the bit-identity tests).
    With emit_stats=True returns an `Act` (tensor + GroupNorm partials when the launch can produce them).
    weight_s16 = (packed, scale, packed_skip or None) from pack_conv_weight_s16: 3x3 / stride-1 launches whose shape
    qualifies then run the split-fp16 kernel (fp32-grade products on the fp16 matrix pipe) instead of the fp32 one.
    Operands such a launch reads RAW (no GroupNorm: the main operand when gn is None, the fused shortcut's input always)
    are range-guarded: `raw_amax` = their [B][AMAX_N] bound (amax_bound / group_norm_affine(want_amax=True)); when it is
    not supplied it is computed here from the producers' partials or the tensors."""
    a_src0, a_src1 = src0, src1                 # (possibly `Act`s: their partials feed the operand bound)
    src0 = src0.t if isinstance(src0, Act) else src0
    src1 = src1.t if isinstance(src1, Act) else src1
    res = res.t if isinstance(res, Act) else res
    B, Hs, Ws, C0 = src0.shape
    C1 = 0 if src1 is None else src1.shape[3]
    Hin, Win = (2 * Hs, 2 * Ws) if ups else (Hs, Ws)
    if pad is None:
        pad = ksize // 2
    if out_hw is None:
        Ho, Wo = (Hin // stride, Win // stride)
    else:
        Ho, Wo = out_hw
    if out is None:
        shape = (B, cout, Ho, Wo) if out_nchw else (B, Ho, Wo, cout)
        out = torch.empty(shape, dtype=torch.float32, device=src0.device)
    d = ConvDesc()
    d.src0, d.src1, d.weight = _p(_f32c(src0, "src0")), _p(src1), _p(_f32c(weight, "weight"))
    d.bias, d.badd, d.res = _p(bias), _p(badd), _p(res)
    d.gn_scale, d.gn_shift = (None, None) if gn is None else (_p(gn[0]), _p(gn[1]))
    d.out = _p(out)
    d.B, d.Hin, d.Win, d.C0, d.C1, d.Cout = B, Hin, Win, C0, C1, cout
    d.ksize, d.stride, d.pad, d.Ho, d.Wo = ksize, stride, pad, Ho, Wo
    d.ups, d.gn_silu, d.out_nchw = int(ups), int(gn_silu), int(out_nchw)
    d.badd_stride, d.tile, d.res_ups = badd_stride, tile, int(res_ups)
    d.flags = 1 if (one_tile or FORCE_ONE_TILE) else 0
    L = _lib.lib()
    if out_nchw and cout <= 4 and skip is None and weight_f16 is None and L.ddnm_conv3x3_small_cout_f32_supported(ctypes.byref(d)) == 1:
        # the network's 3-channel output convolution: HBM-bound vector-ALU kernel (csrc/conv_small_f32.hip)
        if _timer is None:
            check(L.ddnm_conv3x3_small_cout_f32(ctypes.byref(d), _stream()), "ddnm_conv3x3_small_cout_f32")
        else:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            check(L.ddnm_conv3x3_small_cout_f32(ctypes.byref(d), _stream()), "ddnm_conv3x3_small_cout_f32")
            e1.record()
            _timer.records.append(("conv3x3_small_cout_f32", 2.0 * B * Ho * Wo * cout * 9 * C0, e0, e1))
            _timer.shapes.append((B, Ho, Wo, C0, cout, 3, 1, 0, 0, gn is not None, False))
        return Act(out, None, 0) if emit_stats else out
    # fp16-operand MFMA path (the reference's use_fp16 torso) when packed fp16 weights are supplied and the
    # shape qualifies; everything else runs the exact-fp32 kernels
    f16, f16_1x1 = False, False
    if skip is not None:
        # the whole shortcut part of the descriptor is filled BEFORE the support queries below, so that every planner
        # sees the descriptor the launch will see (some key on the pointer, some on the channel counts);
        # skip_weight is chosen further down, once the kernel family is known
        s0 = skip[0].t if isinstance(skip[0], Act) else skip[0]
        s1 = None if skip[1] is None else (skip[1].t if isinstance(skip[1], Act) else skip[1])
        d.skip0, d.skip1 = _p(s0), _p(s1)
        d.SC0, d.SC1 = s0.shape[3], (0 if s1 is None else s1.shape[3])
    s16 = (weight_s16 is not None and weight_f16 is None and ksize == 3 and stride == 1
           and (skip is None or (weight_s16[2] is not None and gn is not None and d.SC0 % 32 == 0 and d.SC1 % 32 == 0))
           and L.ddnm_conv3x3_s16_supported(ctypes.byref(d)) == 1)
    # ... and the per-tap gather form of the same arithmetic for 1x1 / strided / 8x8-level launches (no fused shortcut)
    s16g = (not s16 and weight_s16 is not None and weight_f16 is None and skip is None and not _NO_S16_GATHER
            and L.ddnm_conv_gather_s16_supported(ctypes.byref(d)) == 1)
    if s16 or s16g:
        d.weight = weight_s16[0].data_ptr()
        d.acc_scale = 1.0 / (float(weight_s16[1]) * _s16_act_scale())
        if gn is None or skip is not None:
            # operand-range guard: a raw operand may have ANY fp32 magnitude; the kernel scales it per image by a power
            # of two derived on the device from this bound (include/ddnm_hip.h::ddnm_conv_desc::amax_in)
            if raw_amax is None:
                raw_amax = amax_bound(*skip) if skip is not None else amax_bound(a_src0, a_src1)
            d.amax_in = raw_amax.data_ptr()
    if weight_f16 is not None:
        if ksize == 3:
            f16 = L.ddnm_conv3x3_f16_supported(ctypes.byref(d)) == 1
            if (not f16 and stride == 1 and not ups and skip is None and not out_nchw and not res_ups
                    and Hs * Ws < 256 and (C0 | C1) % 8 == 0):
                # lowest-resolution level (a 256-pixel tile would span several images): im2col with the GroupNorm
                # prologue, then ONE fp16 GEMM with K = 9*Cin -- the (O,ky,kx,I)-packed weights are its matrix
                d.ksize, d.pad, d.C0, d.C1, d.gn_scale, d.gn_shift = 1, 0, 9 * (C0 + C1), 0, None, None
                if L.ddnm_conv1x1_f16_supported(ctypes.byref(d)) == 1:
                    col = _f16_scratch(src0.device, B * Hs * Ws * 9 * (C0 + C1), slot=1)
                    check(L.ddnm_im2col3x3_f16(_p(src0), _p(src1), None if gn is None else _p(gn[0]),
                                               None if gn is None else _p(gn[1]), col.data_ptr(), B, Hs, Ws, C0, C1,
                                               int(gn_silu), _stream()), "ddnm_im2col3x3_f16")
                    d.src0, d.src1, d.src_f16 = col.data_ptr(), None, 1
                    f16 = f16_1x1 = True
                    gn = None
                else:
                    d.ksize, d.pad, d.C0, d.C1 = 3, pad, C0, C1
                    d.gn_scale, d.gn_shift = (None, None) if gn is None else (_p(gn[0]), _p(gn[1]))
        elif ksize == 1 and skip is None:
            d.gn_scale, d.gn_shift = None, None         # the GEMM kernel has no fused prologue: always the pre-pass
            f16 = f16_1x1 = L.ddnm_conv1x1_f16_supported(ctypes.byref(d)) == 1
            if not f16 and gn is not None:
                d.gn_scale, d.gn_shift = _p(gn[0]), _p(gn[1])
    if f16:
        d.weight = weight_f16.data_ptr()
        if gn is not None and (f16_1x1 or cout >= _F16_PREPASS_MIN_COUT):
            # GroupNorm (+ swish) once per element into an fp16 scratch tensor instead of once per
            # (128-output-channel tile x halo overlap) inside the conv's loader
            h16 = _f16_scratch(src0.device, B * Hs * Ws * (C0 + C1))
            check(L.ddnm_gn_apply_f16(_p(src0), _p(src1), _p(gn[0]), _p(gn[1]), h16.data_ptr(), B, Hs * Ws, C0, C1,
                                      int(gn_silu), _stream()), "ddnm_gn_apply_f16")
            d.src0, d.src1, d.C0, d.C1 = h16.data_ptr(), None, C0 + C1, 0
            d.gn_scale, d.gn_shift, d.src_f16 = None, None, 1
    if skip is not None:
        # fused 1x1 shortcut; the caller has checked `conv_fuses_skip` (3x3 halo launch)
        d.skip_weight = _p(skip_weight_f16) if f16 else (weight_s16[2].data_ptr() if s16 else _p(skip_weight))
        if f16 and skip_weight_f16 is None:
            raise ValueError("fp16 launch with a fused shortcut needs skip_weight_f16")
    if f16_1x1:
        fn_run, fn_tiles, fn_ws = L.ddnm_conv1x1_f16_f32, L.ddnm_conv1x1_f16_stats_tiles, L.ddnm_conv1x1_f16_workspace_floats
    elif f16:
        fn_run, fn_tiles, fn_ws = L.ddnm_conv3x3_f16_f32, L.ddnm_conv3x3_f16_stats_tiles, L.ddnm_conv3x3_f16_workspace_floats
    elif s16:
        fn_run, fn_tiles, fn_ws = L.ddnm_conv3x3_s16_f32, L.ddnm_conv3x3_s16_stats_tiles, L.ddnm_conv3x3_s16_workspace_floats
    elif s16g:
        fn_run, fn_tiles, fn_ws = (L.ddnm_conv_gather_s16_f32, L.ddnm_conv_gather_s16_stats_tiles,
                                   L.ddnm_conv_gather_s16_workspace_floats)
    else:
        fn_run, fn_tiles, fn_ws = L.ddnm_conv2d_f32, L.ddnm_conv2d_f32_stats_tiles, L.ddnm_conv2d_f32_workspace_floats
    stats, tiles = None, 0
    if emit_stats and not _NO_FUSED_GN:
        tiles = fn_tiles(ctypes.byref(d))
        if tiles > 0:
            stats = torch.empty(B * tiles * cout * 2, dtype=torch.float32, device=src0.device)
            d.stats_out = stats.data_ptr()
    need = fn_ws(ctypes.byref(d))
    if need > 0:
        ws = _conv_workspace(src0.device, need)
        d.workspace, d.workspace_floats = ws.data_ptr(), ws.numel()
    if _timer is None:
        check(fn_run(ctypes.byref(d), _stream()), "ddnm_conv2d")
    else:
        if f16_1x1:
            variant = "conv1x1_f16<256x128>"
        elif f16:
            variant = "conv3x3_halo_f16<256x128>"
        elif s16:
            variant = "conv3x3_s16_persist<256x128>" if L.ddnm_conv3x3_s16_persistent(ctypes.byref(d)) == 1 else "conv3x3_halo_s16<256x128>"
        elif s16g:
            variant = "conv_gather_s16"
        else:
            tn = L.ddnm_conv2d_f32_tile_n(ctypes.byref(d))
            kind = "conv3x3_halo_f32" if (ksize == 3 and stride == 1) else "conv_gather_f32"
            variant = kind + {128: "<128x128>", 64: "<64x64>", 32: "<128x32>"}[tn]
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(fn_run(ctypes.byref(d), _stream()), "ddnm_conv2d")
        e1.record()
        flops = 2.0 * B * Ho * Wo * cout * (ksize * ksize * (C0 + C1) + d.SC0 + d.SC1)      # + fused 1x1 shortcut
        _timer.records.append((variant, flops, e0, e1))
        _timer.shapes.append((B, Ho, Wo, C0 + C1, cout, ksize, stride, int(ups), d.SC0 + d.SC1, gn is not None,
                              res is not None))
    if _S16_CHECK and (s16 or s16g) and not bool(torch.isfinite(out).all()):
        raise FloatingPointError(f"split-fp16 launch produced non-finite values (B={B}, {Hin}x{Win}, Cin={C0 + C1}, "
                                 f"Cout={cout}, k={ksize}): an operand left fp16 range; DDNM_CONV_F32=mfma32 selects the "
                                 "all-fp32 engine")
    if emit_stats:
        return Act(out, stats, tiles)
    return out


# ----------------------------------------------------------------------------- fp16-activation path (csrc/conv16.hip)
CONV16_COUT_ALIGN = 256


def _f16c(t, name):
    if t.dtype != torch.float16 or not t.is_contiguous() or not t.is_cuda:
        raise ValueError(f"{name}: expected a contiguous float16 device tensor, got {t.dtype} "
                         f"contiguous={t.is_contiguous()} device={t.device}")
    return t


def pack_conv_weight16(w, cin_pad=None):
    """OIHW (or OI for Conv1d k=1 / Linear-like) -> fp16 (O, ky, kx, I) with Cout padded to 256, Cin to `cin_pad`."""
    if w.dim() == 2:
        w = w[:, :, None, None]
    cout, cin, kh, kw = w.shape
    cin_pad = cin if cin_pad is None else cin_pad
    cout_pad = (cout + CONV16_COUT_ALIGN - 1) // CONV16_COUT_ALIGN * CONV16_COUT_ALIGN
    out = torch.zeros(cout_pad, kh * kw, cin_pad, dtype=torch.float16, device=w.device)
    out[:cout, :, :cin] = w.float().permute(0, 2, 3, 1).reshape(cout, kh * kw, cin).to(torch.float16)
    return out.contiguous()


def _conv16_desc(src, weight, cout, ksize, H, W, ups, skip, skip_weight, bias, res, res_ups, src1=None, gn=None,
                 gn_silu=True):
    B = src.shape[0]
    d = Conv16Desc()
    d.src, d.weight, d.bias, d.res = _p(_f16c(src, "src")), _p(_f16c(weight, "weight")), _p(bias), _p(res)
    cin = src.shape[-1]
    if src1 is not None:
        d.src1, d.C0 = _p(_f16c(src1, "src1")), cin
        cin += src1.shape[-1]
    if gn is not None:
        d.gn_scale, d.gn_shift, d.gn_silu = _p(gn[0]), _p(gn[1]), int(gn_silu)
    d.B, d.H, d.W, d.Cin, d.Cout, d.ksize = B, H, W, cin, cout, ksize
    d.ups, d.res_ups = int(ups), int(res_ups)
    if skip is not None:
        s0, s1 = skip
        d.skip0, d.skip1, d.skip_weight = _p(_f16c(s0, "skip0")), _p(s1), _p(_f16c(skip_weight, "skip_weight"))
        d.SC0, d.SC1 = s0.shape[-1], (0 if s1 is None else s1.shape[-1])
    return d


def conv16_supported(B, H, W, cin, cout, ksize, ups=False):
    d = Conv16Desc()
    d.B, d.H, d.W, d.Cin, d.Cout, d.ksize, d.ups = B, H, W, cin, cout, ksize, int(ups)
    return _lib.lib().ddnm_conv16_supported(ctypes.byref(d)) == 1


def conv16(src, weight, cout, ksize, *, src1=None, gn=None, gn_silu=True, bias=None, res=None, res_ups=False, ups=False,
           skip=None, skip_weight=None, emit_stats=True, out=None, fin=None):
    """fp16 NHWC convolution of the `use_fp16` torso (include/ddnm_hip.h::ddnm_conv16_desc).  The operand is
    concat_c(src, src1) [B,Hs,Ws,Cin] fp16: raw with `gn` = (scale, shift) (3x3 only: GroupNorm affine + swish fused,
    applied in LDS) or already activated.  Returns an `Act` whose tensor is fp16 [B,H,W,cout] and whose GroupNorm
    partials (when the launch can emit them) describe exactly those rounded values."""
    src = src.t if isinstance(src, Act) else src
    src1 = src1.t if isinstance(src1, Act) else src1
    res = res.t if isinstance(res, Act) else res
    B, Hs, Ws, _ = src.shape
    H, W = (2 * Hs, 2 * Ws) if ups else (Hs, Ws)
    if skip is not None:
        skip = tuple(None if s is None else (s.t if isinstance(s, Act) else s) for s in skip)
    d = _conv16_desc(src, weight, cout, ksize, H, W, ups, skip, skip_weight, bias, res, res_ups, src1, gn, gn_silu)
    L = _lib.lib()
    if out is None:
        out = torch.empty(B, H, W, cout, dtype=torch.float16, device=src.device)
    d.out = _p(_f16c(out, "out"))
    fused_fin = False
    if fin is not None and emit_stats:
        # fin = (name, gamma, beta, film or None, film_stride, eps, workspace): the GroupNorm that will consume `out`
        name, gamma, beta, film, film_stride, eps, ws = fin
        d.fin_gamma, d.fin_beta, d.fin_film = _p(gamma), _p(beta), _p(film)
        d.fin_scale, d.fin_shift = _p(ws.scale), _p(ws.shift)
        d.fin_eps, d.fin_film_stride, d.fin_groups = eps, film_stride, 32
        fused_fin = ws.scale.numel() >= B * cout and L.ddnm_conv16_fuses_fin(ctypes.byref(d)) == 1
        if not fused_fin:
            d.fin_gamma = None
        else:
            ws.generation += 1
    stats, tiles = None, 0
    if emit_stats:
        tiles = L.ddnm_conv16_stats_tiles(ctypes.byref(d))
        if tiles < 0:
            check(tiles, "ddnm_conv16_stats_tiles")
        if tiles > 0:
            stats = torch.empty(B * tiles * cout * 2, dtype=torch.float32, device=src.device)
            d.stats_out = stats.data_ptr()
    need = L.ddnm_conv16_workspace_floats(ctypes.byref(d))
    if need < 0:
        check(int(need), "ddnm_conv16_workspace_floats")
    if need > 0:
        ws = _conv_workspace(src.device, need)
        d.workspace, d.workspace_floats = ws.data_ptr(), ws.numel()
    if _timer is None:
        check(L.ddnm_conv16(ctypes.byref(d), _stream()), "ddnm_conv16")
    else:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(L.ddnm_conv16(ctypes.byref(d), _stream()), "ddnm_conv16")
        e1.record()
        flops = 2.0 * B * H * W * cout * (ksize * ksize * d.Cin + d.SC0 + d.SC1)
        _timer.records.append((f"conv16<{ksize}x{ksize}>", flops, e0, e1))
        _timer.shapes.append((B, H, W, d.Cin, cout, ksize, 1, int(ups), d.SC0 + d.SC1, False, res is not None))
    return Act(out, stats, tiles, gn=(fin[6].scale, fin[6].shift, fin[0], fin[6], fin[6].generation) if fused_fin else None)


def conv16_out(src, weight, cout, bias=None, gn=None, gn_silu=True):
    """The network's output convolution on the fp16 path: 3x3, cout <= 32, fp32 NCHW result (unet.py:627-631,664);
    `gn` = (scale, shift) fuses the GroupNorm affine + swish of `out.0/out.1` into its loader like `conv16`."""
    src = src.t if isinstance(src, Act) else src
    B, H, W, _ = src.shape
    d = _conv16_desc(src, weight, cout, 3, H, W, False, None, None, bias, None, False, None, gn, gn_silu)
    d.out_nchw_f32 = 1
    out = torch.empty(B, cout, H, W, dtype=torch.float32, device=src.device)
    d.out = out.data_ptr()
    if _timer is None:
        check(_lib.lib().ddnm_conv16(ctypes.byref(d), _stream()), "ddnm_conv16(out)")
    else:
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(_lib.lib().ddnm_conv16(ctypes.byref(d), _stream()), "ddnm_conv16(out)")
        e1.record()
        _timer.records.append(("conv16<out>", 2.0 * B * H * W * cout * 9 * d.Cin, e0, e1))
        _timer.shapes.append((B, H, W, d.Cin, cout, 3, 1, 0, 0, False, False))
    return out


def gn_apply16(src0, src1, gn, silu, pool=False):
    """fp16 operand of the next convolution: act(concat_c(src0, src1) * scale + shift), optionally 2x2 average-pooled
    (gn=None: plain concat / pooling of the raw tensors)."""
    src0 = src0.t if isinstance(src0, Act) else src0
    src1 = src1.t if isinstance(src1, Act) else src1
    B, H, W, C0 = src0.shape
    C1 = 0 if src1 is None else src1.shape[3]
    Ho, Wo = (H // 2, W // 2) if pool else (H, W)
    out = torch.empty(B, Ho, Wo, C0 + C1, dtype=torch.float16, device=src0.device)
    sc, sh = (None, None) if gn is None else gn
    check(_lib.lib().ddnm_gn_apply_h16(_p(_f16c(src0, "src0")), _p(src1), _p(sc), _p(sh), _p(out), B, H, W, C0, C1,
                                       int(silu), int(pool), _stream()), "ddnm_gn_apply_h16")
    return out


def im2col16(src0, src1, gn, silu):
    src0 = src0.t if isinstance(src0, Act) else src0
    src1 = src1.t if isinstance(src1, Act) else src1
    B, H, W, C0 = src0.shape
    C1 = 0 if src1 is None else src1.shape[3]
    out = torch.empty(B, H, W, 9 * (C0 + C1), dtype=torch.float16, device=src0.device)
    sc, sh = (None, None) if gn is None else gn
    check(_lib.lib().ddnm_im2col3x3_h16(_p(_f16c(src0, "src0")), _p(src1), _p(sc), _p(sh), _p(out), B, H, W, C0, C1,
                                        int(silu), _stream()), "ddnm_im2col3x3_h16")
    return out


def nchw_to_nhwc16(x, cpad):
    B, C, H, W = x.shape
    out = torch.empty(B, H, W, cpad, dtype=torch.float16, device=x.device)
    check(_lib.lib().ddnm_nchw_to_nhwc_h16(_p(_f32c(x, "x")), _p(out), B, C, H * W, cpad, _stream()),
          "ddnm_nchw_to_nhwc_h16")
    return out


def gn_stats16(t):
    """GroupNorm partials of an fp16 NHWC tensor whose producer could not emit them."""
    B, H, W, C = t.shape
    hw = H * W
    tiles = 1
    for cand in (64, 32, 16, 8, 4, 2):
        if hw % cand == 0 and hw // cand >= 4:
            tiles = cand
            break
    stats = torch.empty(B * tiles * C * 2, dtype=torch.float32, device=t.device)
    check(_lib.lib().ddnm_gn_stats_h16(_p(_f16c(t, "t")), _p(stats), B, hw, C, tiles, _stream()), "ddnm_gn_stats_h16")
    return Act(t, stats, tiles)


def attn16(qkv, C, lse=None):
    """Fused multi-head attention (head dim 64) over the legacy-ordered fp16 qkv tensor [B,H,W,3C] -> [B,H,W,C].
    `lse` (fp32 [B, C/64, H*W], optional) receives the per-query base-2 log-sum-exp the backward pass needs."""
    B, H, W, C3 = qkv.shape
    assert C3 == 3 * C
    out = torch.empty(B, H, W, C, dtype=torch.float16, device=qkv.device)
    if lse is None:
        check(_lib.lib().ddnm_attn16_d64(_p(_f16c(qkv, "qkv")), _p(out), B, H * W, C, _stream()), "ddnm_attn16_d64")
    else:
        if lse.dtype != torch.float32 or lse.numel() != B * (C // 64) * H * W or not lse.is_contiguous():
            raise ValueError("attn16: lse must be a contiguous float32 tensor [B, C/64, H*W]")
        check(_lib.lib().ddnm_attn16_d64_lse(_p(_f16c(qkv, "qkv")), _p(out), _p(lse), B, H * W, C, _stream()),
              "ddnm_attn16_d64_lse")
    return out


def attn16_bwd(qkv, o, dO, lse):
    """Input gradient of `attn16`: dqkv fp16 [B,H,W,3C] from qkv, the forward output o, its gradient dO and lse; the
    probabilities are recomputed tile by tile (csrc/attn16_bwd.hip), nothing of size [T][T] is stored."""
    B, H, W, C3 = qkv.shape
    C = C3 // 3
    dqkv = torch.empty_like(qkv)
    dsum = torch.empty_like(lse)
    check(_lib.lib().ddnm_attn16_d64_bwd(_p(_f16c(qkv, "qkv")), _p(_f16c(o, "o")), _p(_f16c(dO, "dO")), _p(lse), _p(dsum),
                                         _p(dqkv), B, H * W, C, _stream()), "ddnm_attn16_d64_bwd")
    return dqkv


# ----------------------------------------------------------------------------- GroupNorm
class GroupNormWorkspace:
    """Scratch shared by every GroupNorm of a forward pass (stream order makes reuse safe)."""

    def __init__(self, device, max_batch, max_channels, max_partial_doubles):
        self.partial = torch.empty(max_partial_doubles, dtype=torch.float64, device=device)
        self.scale = torch.empty(max_batch * max_channels, dtype=torch.float32, device=device)
        self.shift = torch.empty(max_batch * max_channels, dtype=torch.float32, device=device)
        # bumped by every launch that writes scale / shift (finalize kernels, the fused `fin` pass of conv16): a consumer
        # that was handed "already finalized" buffers checks that nobody has overwritten them since (Act.gn)
        self.generation = 0


def gn_nchunk(hw, c):
    n = _lib.lib().ddnm_gn_nchunk(hw, c)
    if n <= 0:
        check(n, "ddnm_gn_nchunk")
    return n


def group_norm_affine(src0, src1, gamma, beta, eps, ws, groups=32, film=None, film_stride=0, keep=None, want_amax=False):
    """(scale, shift) [B][C] such that GN(x)[b,:,c] = x*scale + shift; no normalised tensor is written.
    `film` (rows [s | t], row stride film_stride) folds the FiLM modulation GN(x)*(1+s)+t into the affine.
    src0 / src1 may be tensors or `Act`s; when every source carries conv-epilogue partials the statistics
    come from those (no pass over the activation), otherwise from the stand-alone statistics kernel."""
    # keep: optional dict that receives private copies of (scale, shift, mean_rstd) for a backward pass
    # want_amax: also return the [B][AMAX_N] operand bound of concat(src0, src1) (a third value), emitted by the same
    # finalize launch when the statistics come from tile partials (groups == AMAX_N), by amax_bound otherwise
    a0 = src0 if isinstance(src0, Act) else Act(src0)
    a1 = None if src1 is None else (src1 if isinstance(src1, Act) else Act(src1))
    if a0.t.dtype == torch.float16:          # fp16-activation path: statistics always come as tile partials
        if a0.stats is None:
            a0 = gn_stats16(a0.t)
        if a1 is not None and a1.stats is None:
            a1 = gn_stats16(a1.t)
    src0, src1 = a0.t, (None if a1 is None else a1.t)
    B, H, W, C0 = src0.shape
    C1 = 0 if src1 is None else src1.shape[3]
    C, HW = C0 + C1, H * W
    sc_buf, sh_buf, mr = ws.scale, ws.shift, None
    if keep is None:
        ws.generation += 1
    if keep is not None:
        sc_buf = torch.empty(B * C, dtype=torch.float32, device=src0.device)
        sh_buf = torch.empty(B * C, dtype=torch.float32, device=src0.device)
        mr = torch.empty(B * groups * 2, dtype=torch.float32, device=src0.device)
        keep.update(scale=sc_buf, shift=sh_buf, mean_rstd=mr, groups=groups)
    if a0.stats is not None and (a1 is None or a1.stats is not None):
        if sc_buf.numel() < B * C:
            raise ValueError("GroupNorm workspace too small")
        amax = None
        if want_amax and groups == AMAX_N:
            amax = torch.empty(B * AMAX_N, dtype=torch.float32, device=src0.device)
        check(_lib.lib().ddnm_gn_finalize_tiles_amax_f32(_p(a0.stats), a0.tiles, C0, None if a1 is None else _p(a1.stats),
                                                         0 if a1 is None else a1.tiles, C1, _p(gamma), _p(beta), B, HW,
                                                         groups, eps, _p(sc_buf), _p(sh_buf), _p(film), film_stride,
                                                         _p(mr), _p(amax), _stream()), "ddnm_gn_finalize_tiles_amax_f32")
        if want_amax:
            return sc_buf, sh_buf, (amax if amax is not None else amax_bound(a0, a1))
        return sc_buf, sh_buf
    nchunk = gn_nchunk(HW, C)
    need = B * nchunk * groups * 2
    if ws.partial.numel() < need or sc_buf.numel() < B * C:
        raise ValueError("GroupNorm workspace too small")
    L = _lib.lib()
    check(L.ddnm_gn_stats_f32(_p(src0), _p(src1), B, HW, C0, C1, groups, _p(ws.partial), nchunk, _stream()),
          "ddnm_gn_stats_f32")
    check(L.ddnm_gn_finalize_f32(_p(ws.partial), nchunk, _p(gamma), _p(beta), B, HW, C, groups, eps, _p(sc_buf),
                                 _p(sh_buf), _p(film), film_stride, _p(mr), _stream()), "ddnm_gn_finalize_f32")
    if want_amax:
        return sc_buf, sh_buf, amax_bound(a0, a1)
    return sc_buf, sh_buf


# ----------------------------------------------------------------------------- GEMM / softmax / linear
def bgemm(A, Bm, C, M, N, K, *, lda, ldb, ldc, transb, batch=1, inner=1, sA=(0, 0), sB=(0, 0), sC=(0, 0),
          D=None, ldd=0, sD=(0, 0), alpha=1.0, beta=0.0, transa=False):
    d = GemmDesc()
    d.A, d.Bm, d.D, d.C = _p(A), _p(Bm), _p(D), _p(C)
    d.M, d.N, d.K, d.lda, d.ldb, d.ldc, d.ldd = M, N, K, lda, ldb, ldc, ldd
    d.transb, d.batch, d.inner = int(transb), batch, inner
    d.sAo, d.sAi, d.sBo, d.sBi, d.sCo, d.sCi, d.sDo, d.sDi = sA[0], sA[1], sB[0], sB[1], sC[0], sC[1], sD[0], sD[1]
    d.alpha, d.beta, d.transa = alpha, beta, int(transa)
    if _timer is not None:           # FLOP accounting of an instrumented pass (bench.py counts the classifier's work)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        check(_lib.lib().ddnm_bgemm_f32(ctypes.byref(d), _stream()), "ddnm_bgemm_f32")
        e1.record()
        _timer.records.append(("bgemm_f32", 2.0 * M * N * K * batch, e0, e1))
        _timer.shapes.append((batch, M, N, K, 0, 1, 1, 0, 0, False, False))
        return C
    check(_lib.lib().ddnm_bgemm_f32(ctypes.byref(d), _stream()), "ddnm_bgemm_f32")
    return C


def attn_fused_supported(T, C):
    return _lib.lib().ddnm_attn_fused_supported(int(T), int(C)) == 1


def attn_fused(qkv, B, T, C, s_qk, s_v, scale, out=None):
    """Single-head self-attention over qkv [B, T, 3C] fp32 (q | k | v per token) in ONE launch (csrc/attn_d512.hip): the two
    torch.bmm and the softmax of guided_diffusion/models.py:171-185.  s_qk / s_v: power-of-two operand scales (see
    models.Model: a static bound per checkpoint)."""
    if out is None:
        out = torch.empty(B, T, C, dtype=torch.float32, device=qkv.device)
    check(_lib.lib().ddnm_attn_fused_f32(_p(_f32c(qkv, "qkv")), _p(out), B, T, C, float(s_qk), float(s_v), float(scale), _stream()),
          "ddnm_attn_fused_f32")
    return out


def softmax_rows_(x, rows, n, ld, scale):
    check(_lib.lib().ddnm_softmax_rows_f32(_p(x), rows, n, ld, scale, _stream()), "ddnm_softmax_rows_f32")
    return x


def linear(x, W, bias, silu_in=False, out=None):
    B, K = x.shape
    N = W.shape[0]
    if out is None:
        out = torch.empty(B, N, dtype=torch.float32, device=x.device)
    check(_lib.lib().ddnm_linear_f32(_p(x), _p(W), _p(bias), _p(out), B, K, N, int(silu_in), _stream()),
          "ddnm_linear_f32")
    return out


def timestep_embedding(t, freq, order):
    B, half = t.shape[0], freq.shape[0]
    emb = torch.empty(B, 2 * half, dtype=torch.float32, device=t.device)
    check(_lib.lib().ddnm_timestep_embedding_f32(_p(t), _p(freq), _p(emb), B, half, order, _stream()),
          "ddnm_timestep_embedding_f32")
    return emb


def avgpool2_nhwc(x, gn=None, silu=False):
    B, H, W, C = x.shape
    out = torch.empty(B, H // 2, W // 2, C, dtype=torch.float32, device=x.device)
    sc, sh = (None, None) if gn is None else gn
    check(_lib.lib().ddnm_avgpool2_nhwc_f32(_p(_f32c(x, "x")), _p(sc), _p(sh), int(silu), _p(out), B, H // 2, W // 2, C,
                                            _stream()), "ddnm_avgpool2_nhwc_f32")
    return out


def embedding_add_(emb, table, idx):
    idx = idx.to(device=emb.device, dtype=torch.int64).contiguous()
    if idx.numel() != emb.shape[0]:
        raise ValueError("embedding_add_: one index per row expected")
    check(_lib.lib().ddnm_embedding_add_f32(_p(emb), _p(table), _p(idx), emb.shape[0], emb.shape[1], table.shape[0],
                                            _stream()), "ddnm_embedding_add_f32")
    return emb


def nchw_to_nhwc_pad(x, cpad):
    B, C, H, W = x.shape
    out = torch.empty(B, H, W, cpad, dtype=torch.float32, device=x.device)
    check(_lib.lib().ddnm_nchw_to_nhwc_pad_f32(_p(_f32c(x, "x")), _p(out), B, C, H * W, cpad, _stream()),
          "ddnm_nchw_to_nhwc_pad_f32")
    return out


def nchw_im2col3x3_pad(x, cpad):
    """[B,C,H,W] -> NHWC [B,H,W,cpad] holding the 9*C taps of the 3x3 / pad 1 input convolution (C = 3: one K chunk)."""
    B, C, H, W = x.shape
    out = torch.empty(B, H, W, cpad, dtype=torch.float32, device=x.device)
    check(_lib.lib().ddnm_nchw_im2col3x3_pad_f32(_p(_f32c(x, "x")), _p(out), B, C, H, W, cpad, _stream()),
          "ddnm_nchw_im2col3x3_pad_f32")
    return out


def pack_conv_in_weight_im2col(w, cpad):
    """conv_in weights [Cout, C, 3, 3] -> the 1x1 form over the im2col'ed input: [Cout_pad][1][cpad], entry
    (ky*3+kx)*C + c (the order ddnm_nchw_im2col3x3_pad_f32 writes)."""
    cout, cin = w.shape[0], w.shape[1]
    flat = w.float().permute(0, 2, 3, 1).reshape(cout, 9 * cin)
    cout_pad = (cout + CONV_COUT_ALIGN - 1) // CONV_COUT_ALIGN * CONV_COUT_ALIGN
    out = torch.zeros(cout_pad, 1, cpad, dtype=torch.float32, device=w.device)
    out[:cout, 0, :9 * cin] = flat
    return out.contiguous()


# ----------------------------------------------------------------------------- sampler step
class PhiloxNoise:
    """In-kernel noise of one sampling run (include/ddnm_hip.h::ddnm_step_scalars::rng_*): instead of a tensor, the step
    kernels get (seed, loop iteration, global index of the batch's first image) and draw N(0, I) themselves -- Philox4x32-10
    + Box-Muller, counter = (element / 4, iteration, image), so the values do not depend on batch size or rank count.
    `tensor(k, B, shape)` materialises the SAME values (x_T: iteration 0xFFFFFFFF; time-travel re-noise; DDNM+)."""
    XT_ITER = 0xFFFFFFFF

    def __init__(self, seed, image_base=0):
        self.seed_lo, self.seed_hi = int(seed) & 0xFFFFFFFF, (int(seed) >> 32) & 0xFFFFFFFF
        self.image_base = int(image_base)

    def stamp(self, s, k):
        s.rng_on, s.rng_seed_lo, s.rng_seed_hi, s.rng_iter, s.rng_image_base = 1, self.seed_lo, self.seed_hi, int(k), self.image_base
        return s

    def tensor(self, k, like):
        B = like.shape[0]
        out = torch.empty(like.shape, dtype=torch.float32, device=like.device)
        check(_lib.lib().ddnm_randn_philox_f32(_p(out), B, out.numel() // B, self.seed_lo, self.seed_hi, int(k) & 0xFFFFFFFF,
                                               self.image_base, _stream()), "ddnm_randn_philox_f32")
        return out


def step_scalars(at, at_next, eta, lam=1.0, gamma=1.0):
    """Host-side scalar terms of one reverse step, evaluated in fp32 exactly like the reference
    (functions/svd_ddnm.py:57,63-65): `at`, `at_next` are fp32 torch scalars (alpha-bar)."""
    s = StepScalars()
    at = at.float()
    at_next = at_next.float()
    s.sqrt_1m_at = float((1 - at).sqrt())
    s.sqrt_at = float(at.sqrt())
    s.sqrt_at_next = float(at_next.sqrt())
    c1 = (1 - at_next).sqrt() * eta
    c2 = (1 - at_next).sqrt() * ((1 - eta ** 2) ** 0.5)
    if gamma != 1.0:
        c1, c2 = gamma * c1, gamma * c2
    s.c1, s.c2, s.lam = float(c1), float(c2), float(lam)
    return s


def _et_args(et):
    """et may be a [B,3,H,W] view of a [B,6,H,W] learn_sigma output (svd_ddnm.py:54-55)."""
    if et.dim() != 4 or et.stride(3) != 1 or et.stride(2) != et.shape[3] or et.stride(1) != et.shape[2] * et.shape[3]:
        raise ValueError("et must be channel-contiguous NCHW")
    return et.data_ptr(), et.stride(0)


def step_x0(xt, et, s, out=None):
    B = xt.shape[0]
    chw = xt.numel() // B
    out = torch.empty_like(xt) if out is None else out
    ep, es = _et_args(et)
    check(_lib.lib().ddnm_step_x0_f32(_p(xt), ep, es, _p(out), B, chw, ctypes.byref(s), _stream()), "ddnm_step_x0_f32")
    return out


def step_combine(x0, proj, apy, noise, et, s, out=None):
    B = x0.shape[0]
    chw = x0.numel() // B
    out = torch.empty_like(x0) if out is None else out
    ep, es = _et_args(et)
    check(_lib.lib().ddnm_step_combine_f32(_p(x0), _p(proj), _p(apy), _p(noise), ep, es, _p(out), B, chw,
                                           ctypes.byref(s), _stream()), "ddnm_step_combine_f32")
    return out


def fill_(t, value=0.0):
    """t[...] = value for a contiguous fp32 tensor (or contiguous slice) through ddnm_fill_f32."""
    if t.numel():
        check(_lib.lib().ddnm_fill_f32(_p(_f32c(t, "t")), t.numel(), float(value), _stream()), "ddnm_fill_f32")
    return t


def renoise(x0, noise, a, b, out=None):
    out = torch.empty_like(x0) if out is None else out
    check(_lib.lib().ddnm_renoise_f32(_p(x0), _p(noise), _p(out), x0.numel(), a, b, _stream()), "ddnm_renoise_f32")
    return out


def finalize_psnr(x, x_orig=None, want_img=True):
    """inverse_data_transform + per-image PSNR (datasets/__init__.py:218-227, diffusion.py:599-602)."""
    B = x.shape[0]
    chw = x.numel() // B
    img = torch.empty_like(x) if want_img else None
    sse = torch.empty(B, dtype=torch.float64, device=x.device) if x_orig is not None else None
    check(_lib.lib().ddnm_finalize_psnr_f32(_p(x), _p(x_orig), _p(img), _p(sse), B, chw, _stream()),
          "ddnm_finalize_psnr_f32")
    psnr = None
    if sse is not None:
        psnr = 10.0 * torch.log10(1.0 / (sse / chw))
    return img, psnr
