"""Split-fp16 3x3 convolution (ddnm_conv3x3_s16_f32) on the GPU: the fp32 celeba path carries every operand as hi + lo
fp16 halves and forms a product from three fp16 MFMAs.  The claim tested here is "fp32-grade": against an fp64 evaluation
of the same layer the kernel must be at least as close as the fp32 MFMA kernel (ddnm_conv2d_f32), including the emitted
GroupNorm partials, for every fused feature of the layer (GroupNorm + swish prologue, concat, x2 upsample, fused 1x1
shortcut, bias / per-sample addend / residual, split-K), for badly scaled operands, and at the model level."""
import pytest
import torch
import torch.nn.functional as F

pytestmark = pytest.mark.gpu

DEV = "cuda"


def _make(B, C0, C1, Cout, H, ups, gn, res, skip, badd=False, seed=0, wscale=None, ascale=1.5):
    g = torch.Generator(device=DEV).manual_seed(seed)
    rn = lambda *s: torch.randn(*s, device=DEV, generator=g)  # noqa: E731
    Ho = 2 * H if ups else H
    cin = C0 + C1
    t = dict(a=rn(B, H, H, C0) * ascale, b=rn(B, H, H, C1) * ascale if C1 else None,
             w=rn(Cout, cin, 3, 3) * (wscale if wscale is not None else 1.0 / (3.0 * cin ** 0.5)), bias=rn(Cout),
             sc=rn(B, cin) * 0.3 + 1.0 if gn else None, sh=rn(B, cin) * 0.3 if gn else None,
             r=rn(B, Ho, Ho, Cout) * 2.0 if res else None, sk=rn(B, H, H, 64) * 2.0 if skip else None,
             wsk=rn(Cout, 64, 1, 1) * 0.1 if skip else None, badd=rn(B, Cout) if badd else None, ups=ups, Cout=Cout)
    return t


def _ref64(t):
    x = t["a"] if t["b"] is None else torch.cat([t["a"], t["b"]], 3)
    x = x.double()
    if t["sc"] is not None:
        x = x * t["sc"].double()[:, None, None, :] + t["sh"].double()[:, None, None, :]
        x = x * torch.sigmoid(x)
    x = x.permute(0, 3, 1, 2)
    if t["ups"]:
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    y = F.conv2d(x, t["w"].double(), t["bias"].double(), padding=1)
    if t["sk"] is not None:
        y = y + F.conv2d(t["sk"].double().permute(0, 3, 1, 2), t["wsk"].double())
    y = y.permute(0, 2, 3, 1)
    if t["badd"] is not None:
        y = y + t["badd"].double()[:, None, None, :]
    if t["r"] is not None:
        y = y + t["r"].double()
    return y


def _run(t, split):
    from ddnm_amd import ops
    w32 = ops.pack_conv_weight(t["w"])
    wsk32 = ops.pack_skip_weight(t["wsk"]) if t["wsk"] is not None else None
    s16 = None
    if split:
        scale = ops.s16_weight_scale(*([t["w"]] + ([t["wsk"]] if t["wsk"] is not None else [])))
        s16 = (ops.pack_conv_weight_s16(t["w"], scale), scale,
               ops.pack_conv_weight_s16(t["wsk"], scale) if t["wsk"] is not None else None)
    gn = None if t["sc"] is None else (t["sc"], t["sh"])
    B = t["a"].shape[0]
    return ops.conv2d(t["a"], w32, t["Cout"], 3, src1=t["b"], bias=t["bias"], res=t["r"], gn=gn, gn_silu=True,
                      badd=t["badd"], badd_stride=(t["Cout"] if t["badd"] is not None else 0),
                      ups=bool(t["ups"]), emit_stats=True, weight_s16=s16,
                      skip=None if t["sk"] is None else (t["sk"], None), skip_weight=wsk32), B


def _errors(t):
    y = _ref64(t)
    out = {}
    for split in (False, True):
        act, B = _run(t, split)
        o = act.t.double()
        rel = ((o - y).norm() / y.norm()).item()
        serr = 0.0
        if act.stats is not None:
            st = act.stats.view(B, act.tiles, -1, 2).double().sum(1)
            s1, s2 = o.sum((1, 2)), (o * o).sum((1, 2))
            serr = max(((st[..., 0] - s1).abs().max() / s1.abs().max()).item(),
                       ((st[..., 1] - s2).abs().max() / s2.abs().max()).item())
        out[split] = (rel, serr, act)
    return out


# B, C0, C1, Cout, H, ups, gn, res, skip, badd
CASES = [
    (2, 128, 0, 128, 32, 0, 1, 1, 0, 1),      # ResnetBlock conv1 form: GroupNorm + swish, temb addend, residual
    (2, 128, 128, 128, 32, 0, 1, 0, 0, 0),    # up path: concat of two sources
    (2, 128, 0, 256, 32, 0, 1, 0, 1, 0),      # conv2 with the fused 1x1 shortcut
    (2, 128, 0, 128, 16, 1, 0, 0, 0, 0),      # Upsample conv: nearest x2 inside the loader (output 32 x 32)
    (2, 256, 0, 256, 16, 0, 1, 1, 0, 0),      # 16 x 16: one 256-pixel tile per image, split-K with the statistics pass
    (1, 512, 512, 512, 16, 0, 1, 0, 0, 0),    # deepest concat, split-K 16
    (2, 160, 0, 128, 64, 0, 1, 1, 0, 0),      # Cin = 5 chunks of 32 (not a multiple of 64)
    (3, 128, 0, 128, 48, 0, 0, 0, 0, 0),      # 48 x 48: 16-wide tiles, plain operands
]


@pytest.mark.parametrize("case", CASES)
def test_split_kernel_is_at_least_as_close_to_fp64_as_the_fp32_kernel(case):
    from ddnm_amd import ops
    B, C0, C1, Cout, H, ups, gn, res, skip, badd = case
    Ho = 2 * H if ups else H
    assert ops.conv_runs_s16(B, Ho, Ho, C0 + C1, Cout), "case must exercise the split kernel"
    e = _errors(_make(*case))
    rel32, s32, a32 = e[False]
    rel16, s16, a16 = e[True]
    assert rel16 < 8e-7, (rel16, rel32)
    assert rel16 <= 1.25 * rel32 + 2e-8, (rel16, rel32)       # fp32-grade: not worse than the fp32 MFMA kernel
    assert s16 < 2e-6 and a16.stats is not None and a16.tiles > 0               # partials describe the tensor written
    assert ((a16.t - a32.t).norm() / a32.t.norm()).item() < 1.5e-6


@pytest.mark.parametrize("wscale,ascale", [(40.0, 1.5), (3e-5, 1.5), (0.05, 300.0), (0.05, 8000.0), (0.05, 0.15)])
def test_split_kernel_badly_scaled_operands(wscale, ascale):
    # weights far from 1 are brought into fp16 range by the per-launch power of two; activations are split as they are:
    # large ones (up to the fp16 maximum, 65504) and ordinary ones are carried to fp32 grade
    t = _make(2, 128, 0, 128, 32, 0, 0, 0, 0, seed=3, wscale=wscale, ascale=ascale)
    t["bias"].zero_()
    e = _errors(t)
    assert e[True][0] < 8e-7, e[True][0]
    assert e[True][0] <= 1.25 * e[False][0] + 2e-8


@pytest.mark.parametrize("ascale,bound", [(0.02, 3e-6), (2e-4, 3e-4)])
def test_split_kernel_uniformly_tiny_operand_degrades_gracefully(ascale, bound):
    # a tensor that is tiny everywhere sits in the subnormal range of `lo`: the absolute error stays <= 2^-25 per
    # element, the relative error of the result grows accordingly (documented domain: csrc/conv_common.h)
    t = _make(2, 128, 0, 128, 32, 0, 0, 0, 0, seed=3, wscale=0.05, ascale=ascale)
    t["bias"].zero_()
    e = _errors(t)
    assert e[True][0] < bound, e[True][0]


def test_fp16_mfma_honours_subnormal_inputs():
    # small operands rely on it: lo = rn16(v - hi) is subnormal for |v| < 0.25
    from ddnm_amd import ops
    sa = ops._s16_act_scale()
    t = _make(1, 128, 0, 128, 16, 0, 0, 0, 0, seed=5, ascale=1.0)
    v = 2.0 ** -6 * (1 + 2.0 ** -12) / sa                      # pre-scaled: hi = 2^-6, lo = 2^-18 (subnormal in fp16)
    t["a"].fill_(v)
    t["w"].fill_(0.0)
    t["w"][:, :, 1, 1] = 1.0 / 128                              # centre tap: out = mean over channels = the value itself
    t["bias"].zero_()
    act, _ = _run(t, True)
    assert abs(act.t[0, 8, 8, 0].item() - v) / v < 1e-6         # 2.4e-4 if `lo` were flushed


def test_shapes_outside_the_split_kernel_fall_back_to_fp32_mfma():
    from ddnm_amd import ops
    assert not ops.conv_runs_s16(8, 8, 8, 512, 512)             # 8 x 8 level: no 256-pixel tile inside an image
    assert not ops.conv_runs_s16(2, 32, 32, 128, 96)            # Cout % 128
    t = _make(2, 512, 0, 160, 8, 0, 1, 1, 0)                    # Cout % 64: neither the halo nor the gather form
    assert not ops.conv_runs_s16_gather(2, 8, 8, 512, 160)
    e = _errors(t)
    assert e[True][0] == e[False][0]                            # same kernel ran both times


def test_celeba_model_split_vs_fp32_mfma_paths():
    from oracle import cases
    from ddnm_amd.guided_diffusion.models import Model
    cfg = cases.weights.celeba_config(resolution=64, ch=128, ch_mult=(1, 2, 2), attn_resolutions=(16,))
    a, b = Model(cfg, device=DEV, split16=True), Model(cfg, device=DEV, split16=False)
    sd = a.random_state_dict(seed=7)
    a.load_state_dict(sd)
    b.load_state_dict(sd)
    assert any(k.endswith(".s16") for k in a.w) and not any(k.endswith(".s16") for k in b.w)
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(4, 3, 64, 64, device=DEV, generator=g)
    t = torch.tensor([999.0, 500.0, 37.0, 0.0], device=DEV)
    ea, eb = a(x, t), b(x, t)
    assert ((ea - eb).double().norm() / eb.double().norm()).item() < 3e-6


# ------------------------------------------------------------------ gather form (1x1, strided, 8 x 8 level)
# B, C0, C1, Cout, H, ksize, stride, gn, gn_silu, res
GATHER_CASES = [
    (2, 128, 0, 128, 32, 3, 2, 0, 0, 0),      # Downsample: pad (0,1,0,1), stride 2 (models.py:61-71)
    (2, 256, 0, 256, 16, 3, 2, 0, 0, 0),
    (2, 512, 0, 1536, 16, 1, 1, 1, 0, 0),     # attention qkv: GroupNorm affine without swish
    (2, 512, 0, 512, 16, 1, 1, 0, 0, 1),      # attention proj_out: residual
    (2, 256, 256, 256, 16, 1, 1, 0, 0, 0),    # un-fused nin_shortcut over a concat
    (8, 512, 0, 512, 8, 3, 1, 1, 1, 1),       # 8 x 8 level: M tile = one image, split-K
    (4, 512, 512, 512, 8, 3, 1, 1, 1, 0),
    (1, 160, 0, 192, 8, 3, 1, 1, 1, 0),       # Cout = 3 x 64, Cin = 5 chunks
]


@pytest.mark.parametrize("case", GATHER_CASES)
def test_gather_split_kernel_vs_fp64_and_fp32_kernel(case):
    from ddnm_amd import ops
    B, C0, C1, Cout, H, k, stride, gn, silu, res = case
    g = torch.Generator(device=DEV).manual_seed(11)
    rn = lambda *s: torch.randn(*s, device=DEV, generator=g)  # noqa: E731
    cin = C0 + C1
    Ho = H // stride
    a, b = rn(B, H, H, C0) * 1.5, (rn(B, H, H, C1) * 1.5 if C1 else None)
    w = rn(Cout, cin, k, k) / (k * cin ** 0.5)
    bias = rn(Cout)
    sc, sh = (rn(B, cin) * 0.3 + 1.0, rn(B, cin) * 0.3) if gn else (None, None)
    r = rn(B, Ho, Ho, Cout) * 2.0 if res else None
    assert ops.conv_runs_s16_gather(B, H, H, cin, Cout, ksize=k, stride=stride)
    assert not (k == 3 and stride == 1 and ops.conv_runs_s16(B, H, H, cin, Cout))
    # fp64 evaluation
    x = (a if b is None else torch.cat([a, b], 3)).double()
    if gn:
        x = x * sc.double()[:, None, None, :] + sh.double()[:, None, None, :]
        if silu:
            x = x * torch.sigmoid(x)
    x = x.permute(0, 3, 1, 2)
    if stride == 2:
        x = F.pad(x, (0, 1, 0, 1))
        y = F.conv2d(x, w.double(), bias.double(), stride=2)
    else:
        y = F.conv2d(x, w.double(), bias.double(), padding=k // 2)
    y = y.permute(0, 2, 3, 1)
    if res:
        y = y + r.double()
    w32 = ops.pack_conv_weight(w)
    s = ops.s16_weight_scale(w)
    errs = {}
    for split in (False, True):
        act = ops.conv2d(a, w32, Cout, k, src1=b, bias=bias, res=r, gn=None if not gn else (sc, sh), gn_silu=bool(silu),
                         stride=stride, pad=(0 if stride == 2 else k // 2), out_hw=(Ho, Ho), emit_stats=True,
                         weight_s16=(ops.pack_conv_weight_s16(w, s), s, None) if split else None)
        o = act.t.double()
        errs[split] = ((o - y).norm() / y.norm()).item()
        if act.stats is not None:
            st = act.stats.view(B, act.tiles, -1, 2).double().sum(1)
            s1, s2 = o.sum((1, 2)), (o * o).sum((1, 2))
            assert ((st[..., 0] - s1).abs().max() / s1.abs().max()).item() < 2e-6
            assert ((st[..., 1] - s2).abs().max() / s2.abs().max()).item() < 2e-6
    assert errs[True] < 8e-7, errs
    assert errs[True] <= 1.25 * errs[False] + 2e-8, errs
