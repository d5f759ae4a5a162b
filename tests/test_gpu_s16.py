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
             wsk=rn(Cout, 64, 1, 1) * 0.1 if skip else None, badd=rn(B, Cout) if badd else None, ups=ups, Cout=Cout,
             amax=None)
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
                      skip=None if t["sk"] is None else (t["sk"], None), skip_weight=wsk32,
                      raw_amax=t["amax"] if split else None), B


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


# Launches of >= 2 tiles per CU (>= 512 tiles, no split-K, no fused shortcut) run the PERSISTENT form of the kernel
# (csrc/conv_s16_persist.hip, round 6): tiles walked by one workgroup per CU, the next tile's halo / weights prefetched in
# the last chunk, the epilogue's stores and residual loads spread over the neighbouring chunks.
# B, C0, C1, Cout, H, ups, gn, res, skip, badd
PERSIST_CASES = [
    (8, 128, 0, 128, 128, 0, 1, 1, 0, 1),     # 512 tiles (2 per CU): GroupNorm + swish, temb addend, residual, 4 chunks
    (2, 128, 128, 128, 256, 0, 1, 0, 0, 0),   # 256 x 256: concat of two sources, 8 chunks
    (2, 128, 0, 128, 128, 1, 0, 0, 0, 0),     # Upsample conv (raw operand: the operand-scale instance), output 256 x 256
    (4, 160, 0, 256, 128, 0, 1, 1, 0, 1),     # Cin = 5 chunks, two channel tiles (the weight stream changes between tiles)
    (3, 64, 0, 128, 256, 0, 1, 1, 0, 0),      # TWO chunks (FIRST directly followed by LAST), 768 tiles: 3 per workgroup
    (9, 96, 0, 128, 128, 0, 0, 0, 0, 0),      # ragged: 576 tiles over 256 workgroups (2 or 3 each), raw operand, no residual
    (5, 128, 0, 128, 128, 0, 1, 0, 0, 1),     # 320 tiles: NOT eligible, stays on the one-tile kernel either way (control)
]


@pytest.mark.parametrize("case", PERSIST_CASES)
def test_persistent_split_kernel_equals_the_one_tile_kernel(case):
    """conv3x3_s16_persist_kernel: same products, same summation order, same epilogue expression as the one-tile kernel --
    output AND GroupNorm partials must be BIT-IDENTICAL to it (`one_tile=True` = ddnm_conv_desc::flags & DDNM_CONV_ONE_TILE
    selects the one-tile kernel for the same descriptor); fp32 grade against fp64 on top."""
    from ddnm_amd import ops
    B, C0, C1, Cout, H, ups, gn, res, skip, badd = case
    t = _make(*case, seed=11)
    if not gn:
        t["amax"] = ops.amax_bound(t["a"], t["b"])
    y = _ref64(t)
    outs = {}
    for one_tile in (True, False):
        scale = ops.s16_weight_scale(t["w"])
        s16 = (ops.pack_conv_weight_s16(t["w"], scale), scale, None)
        a = ops.conv2d(t["a"], ops.pack_conv_weight(t["w"]), Cout, 3, src1=t["b"], bias=t["bias"], res=t["r"],
                       gn=None if t["sc"] is None else (t["sc"], t["sh"]), gn_silu=True, badd=t["badd"],
                       badd_stride=(Cout if t["badd"] is not None else 0), ups=bool(ups), emit_stats=True, weight_s16=s16,
                       raw_amax=t["amax"], one_tile=one_tile)
        torch.cuda.synchronize()
        outs[one_tile] = (a.t.clone(), a.stats.clone(), a.tiles)
    o = outs[False][0].double()
    assert ((o - y).norm() / y.norm()).item() < 8e-7
    st = outs[False][1].view(B, outs[False][2], -1, 2).double().sum(1)
    s1, s2 = o.sum((1, 2)), (o * o).sum((1, 2))
    assert ((st[..., 0] - s1).abs().max() / s1.abs().max()).item() < 2e-6
    assert ((st[..., 1] - s2).abs().max() / s2.abs().max()).item() < 2e-6
    assert outs[True][2] == outs[False][2]
    assert torch.equal(outs[True][0], outs[False][0]), "persistent and one-tile split kernels must agree bit for bit"
    assert torch.equal(outs[True][1], outs[False][1]), "GroupNorm partials of the two kernels must agree bit for bit"
    # repeated launches are bit-identical (no atomics, fixed orders, a deterministic request stream)
    a2 = ops.conv2d(t["a"], ops.pack_conv_weight(t["w"]), Cout, 3, src1=t["b"], bias=t["bias"], res=t["r"],
                    gn=None if t["sc"] is None else (t["sc"], t["sh"]), gn_silu=True, badd=t["badd"],
                    badd_stride=(Cout if t["badd"] is not None else 0), ups=bool(ups), emit_stats=True, weight_s16=s16,
                    raw_amax=t["amax"])
    assert torch.equal(a2.t, outs[False][0]) and torch.equal(a2.stats, outs[False][1])


@pytest.mark.parametrize("wscale,ascale", [(40.0, 1.5), (3e-5, 1.5), (0.05, 300.0), (0.05, 8000.0), (0.05, 0.15)])
def test_split_kernel_badly_scaled_operands(wscale, ascale):
    # weights far from 1 are brought into fp16 range by the per-launch power of two; activations are split as they are:
    # large ones (up to the fp16 maximum, 65504) and ordinary ones are carried to fp32 grade
    t = _make(2, 128, 0, 128, 32, 0, 0, 0, 0, seed=3, wscale=wscale, ascale=ascale)
    t["bias"].zero_()
    e = _errors(t)
    assert e[True][0] < 8e-7, e[True][0]
    assert e[True][0] <= 1.25 * e[False][0] + 2e-8


# ------------------------------------------------------------------ operand-range guard (ddnm_conv_desc::amax_in, ABI 5)
@pytest.mark.parametrize("ascale", [0.02, 2e-4, 1e-5, 1e-9, 1e5, 3e7, 1e12])
def test_raw_operand_of_any_magnitude_stays_fp32_grade(ascale):
    # fp16 carries |v| < 65504 and loses relative precision below ~2^-14; the reference runs this network in fp32, where
    # neither limit exists.  A raw operand is therefore scaled per launch and image by a power of two derived on the
    # device from its bound: uniformly tiny tensors (which kept 1e-4 ... 1e-6 until round 3) and tensors far beyond the
    # fp16 range (inf until round 3) both match fp64 like the fp32 MFMA kernel does.
    t = _make(2, 128, 0, 128, 32, 0, 0, 0, 0, seed=3, wscale=0.05, ascale=ascale)
    t["bias"].zero_()
    e = _errors(t)
    assert torch.isfinite(e[True][2].t).all()
    assert e[True][0] < 8e-7, e[True][0]
    assert e[True][0] <= 1.25 * e[False][0] + 2e-8


def test_operand_scale_is_per_image():
    # image 0 ~ 1e5, image 1 ~ 1e-4 in ONE launch: each image gets its own power of two
    t = _make(2, 128, 0, 128, 32, 0, 0, 0, 0, seed=4, wscale=0.05, ascale=1.0)
    t["a"][0] *= 1e5
    t["a"][1] *= 1e-4
    t["bias"].zero_()
    y = _ref64(t)
    act, _ = _run(t, True)
    for b in range(2):
        assert ((act.t[b].double() - y[b]).norm() / y[b].norm()).item() < 8e-7


@pytest.mark.parametrize("form", ["upsample", "fused_shortcut_huge", "fused_shortcut_tiny", "downsample", "proj_out",
                                  "nin_shortcut_concat"])
@pytest.mark.parametrize("mag", [1e5, 1e-5])
def test_every_raw_operand_launch_form_is_guarded(form, mag):
    """The raw-operand launches of the celeba `Model` (models.py:47-51 Upsample, :61-71 Downsample, :109 nin_shortcut fused
    into conv2 or on its own, :183-189 proj_out) with an input of magnitude 1e5 / 1e-5: finite and fp32 grade against fp64.
    The bound comes from the producer's GroupNorm partials (an `Act`), from the finalize launch, or from the tensor."""
    from ddnm_amd import ops
    g = torch.Generator(device=DEV).manual_seed(21)
    rn = lambda *s: torch.randn(*s, device=DEV, generator=g)  # noqa: E731
    if form == "upsample":
        t = _make(2, 128, 0, 128, 16, 1, 0, 0, 0, seed=5, ascale=mag)
    elif form.startswith("fused_shortcut"):
        t = _make(2, 128, 0, 256, 32, 0, 1, 0, 1, seed=6)
        t["sk"] = t["sk"] * (mag if form.endswith("huge") else 1.0 / max(mag, 1.0 / mag))
    if form in ("upsample", "fused_shortcut_huge", "fused_shortcut_tiny"):
        e = _errors(t)
        assert torch.isfinite(e[True][2].t).all()
        assert e[True][0] < 8e-7 and e[True][0] <= 1.25 * e[False][0] + 2e-8, e
        return
    B, C0, C1, Cout, H, k, stride, res = {"downsample": (2, 128, 0, 128, 32, 3, 2, 0), "proj_out": (2, 512, 0, 512, 16, 1, 1, 1),
                                          "nin_shortcut_concat": (2, 256, 256, 256, 16, 1, 1, 0)}[form]
    a, b = rn(B, H, H, C0) * mag, (rn(B, H, H, C1) * mag if C1 else None)
    w = rn(Cout, C0 + C1, k, k) / (k * (C0 + C1) ** 0.5)
    r = rn(B, H // stride, H // stride, Cout) * mag if res else None
    x = (a if b is None else torch.cat([a, b], 3)).double().permute(0, 3, 1, 2)
    y = F.conv2d(F.pad(x, (0, 1, 0, 1)), w.double(), stride=2) if stride == 2 else F.conv2d(x, w.double(), padding=k // 2)
    y = y.permute(0, 2, 3, 1)
    if res:
        y = y + r.double()
    sw = ops.s16_weight_scale(w)
    # the downsample's input arrives as an `Act` with its producer's GroupNorm partials (what the network passes)
    src = a
    if form == "downsample":
        ident = torch.zeros(C0, C0, 1, 1, device=DEV)
        ident[torch.arange(C0), torch.arange(C0), 0, 0] = 1.0
        src = ops.conv2d(a, ops.pack_conv_weight(ident), C0, 1, emit_stats=True)          # fp32 kernel: a copy + partials
        assert src.stats is not None and torch.equal(src.t, a)
    act = ops.conv2d(src, ops.pack_conv_weight(w), Cout, k, src1=b, res=r, stride=stride, pad=(0 if stride == 2 else k // 2),
                     out_hw=(H // stride, H // stride), emit_stats=True, weight_s16=(ops.pack_conv_weight_s16(w, sw), sw, None))
    assert torch.isfinite(act.t).all()
    assert ((act.t.double() - y).norm() / y.norm()).item() < 8e-7


def test_operand_bound_kernels():
    """ddnm_amax_bound_f32 (tensor / partials form) and the bound emitted by ddnm_gn_finalize_tiles_amax_f32: upper
    bounds of max |x| per image, tight to the tile size (sqrt of a per-(tile, channel) sum of squares)."""
    from ddnm_amd import ops
    g = torch.Generator(device=DEV).manual_seed(8)
    B, H, C = 3, 32, 128
    x = torch.randn(B, H, H, C, device=DEV, generator=g) * torch.tensor([1e-3, 1.0, 4e4], device=DEV)[:, None, None, None]
    true = x.abs().amax((1, 2, 3))
    raw = ops.amax_bound(x).view(B, ops.AMAX_N).amax(1)
    assert torch.equal(raw, true)                                        # tensor form: the exact maximum
    ident = torch.zeros(C, C, 1, 1, device=DEV)
    ident[torch.arange(C), torch.arange(C), 0, 0] = 1.0
    act = ops.conv2d(x, ops.pack_conv_weight(ident), C, 1, emit_stats=True)
    assert act.stats is not None
    st = ops.amax_bound(act).view(B, ops.AMAX_N).amax(1)
    assert bool((st >= true).all()) and bool((st <= true * 16.1).all()), (st, true)      # tiles of <= 256 pixels
    both = ops.amax_bound(act, x * 3.0).view(B, ops.AMAX_N).amax(1)     # two sources, mixed forms
    assert bool((both >= 3.0 * true).all()) and bool((both <= true * 16.1).all())
    ws = ops.GroupNormWorkspace(DEV, B, C, 1)
    gam, bet = torch.ones(C, device=DEV), torch.zeros(C, device=DEV)
    sc, sh, am = ops.group_norm_affine(act, None, gam, bet, 1e-6, ws, want_amax=True)
    sc0, sh0 = (t.clone() for t in (sc[:B * C], sh[:B * C]))
    sc1, sh1 = ops.group_norm_affine(act, None, gam, bet, 1e-6, ws)
    assert torch.equal(sc0, sc1[:B * C]) and torch.equal(sh0, sh1[:B * C])       # the affine itself is unchanged
    fin = am.view(B, ops.AMAX_N)
    per_group = x.abs().view(B, H * H, 32, C // 32).amax((1, 3))
    assert bool((fin >= per_group).all()) and bool((fin.amax(1) <= true * 16.1).all())


def test_split_entry_points_refuse_a_raw_operand_without_bound():
    import ctypes
    from ddnm_amd import _lib, ops
    t = _make(1, 128, 0, 128, 16, 0, 0, 0, 0)
    sw = ops.s16_weight_scale(t["w"])
    w16 = ops.pack_conv_weight_s16(t["w"], sw)
    out = torch.empty(1, 16, 16, 128, device=DEV)
    for fn, k in ((_lib.lib().ddnm_conv3x3_s16_f32, 3), (_lib.lib().ddnm_conv_gather_s16_f32, 1)):
        d = _lib.ConvDesc()
        d.src0, d.weight, d.out = t["a"].data_ptr(), w16.data_ptr(), out.data_ptr()
        d.B, d.Hin, d.Win, d.C0, d.Cout, d.ksize, d.stride, d.pad, d.Ho, d.Wo = 1, 16, 16, 128, 128, k, 1, k // 2, 16, 16
        d.acc_scale = 1.0 / sw
        if k == 1:
            w1 = ops.pack_conv_weight_s16(t["w"][:, :, 1:2, 1:2].contiguous(), sw)
            d.weight = w1.data_ptr()
        assert fn(ctypes.byref(d), ops._stream()) == -1                   # DDNM_E_BADARG: no silent overflow path
        d.amax_in = ops.amax_bound(t["a"]).data_ptr()
        assert fn(ctypes.byref(d), ops._stream()) == 0
    torch.cuda.synchronize()


def test_fp16_mfma_honours_subnormal_inputs():
    # small operands rely on it: lo = rn16(v - hi) is subnormal for |v| < 0.25
    from ddnm_amd import ops
    sa = ops._s16_act_scale()
    t = _make(1, 128, 0, 128, 16, 0, 0, 0, 0, seed=5, ascale=1.0)
    v = 2.0 ** -6 * (1 + 2.0 ** -12) / sa                      # pre-scaled: hi = 2^-6, lo = 2^-18 (subnormal in fp16)
    t["a"].fill_(v)
    t["amax"] = torch.full((ops.AMAX_N,), 2.0 ** 14, device=DEV)   # a bound in [2^14, 2^15): operand scale 2^0
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


def test_normalised_operand_bound_is_checked_at_load_time():
    """GroupNorm'd operands carry no run-time bound because they have a static one (sqrt(n) * max|gamma| + max|beta|); a
    checkpoint whose bound does not fit fp16 sends that convolution to the exact-fp32 kernel at load_state_dict time."""
    from oracle import cases
    from ddnm_amd.guided_diffusion.models import Model
    cfg = cases.weights.celeba_config(resolution=64, ch=128, ch_mult=(1, 2, 2), attn_resolutions=(16,))
    a, b = Model(cfg, device=DEV, split16=True), Model(cfg, device=DEV, split16=False)
    sd = a.random_state_dict(seed=7)
    a.load_state_dict(sd)
    assert a.s16_dropped == []                                  # gamma ~ 1: nothing is dropped
    sd = dict(sd)
    sd["down.0.block.0.norm2.weight"] = sd["down.0.block.0.norm2.weight"] * 3000.0      # bound = sqrt(64*64*4) * 3300 = 4e5
    a.load_state_dict(sd)
    b.load_state_dict(sd)
    assert a.s16_dropped == ["down.0.block.0.conv2"]
    g = torch.Generator(device=DEV).manual_seed(1)
    x = torch.randn(2, 3, 64, 64, device=DEV, generator=g)
    t = torch.tensor([999.0, 37.0], device=DEV)
    ea, eb = a(x, t), b(x, t)
    assert torch.isfinite(ea).all()
    assert ((ea - eb).double().norm() / eb.double().norm()).item() < 3e-6
