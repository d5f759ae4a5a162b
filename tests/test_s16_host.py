"""CPU tests of the split-fp16 packing (ops.pack_conv_weight_s16 / s16_weight_scale) and of the arithmetic the split kernel
relies on: an fp32 value carried as hi + lo fp16 halves, a product formed as hi*hi' + hi*lo' + lo*hi' with exact fp16 x fp16
products and fp32 accumulation (ddnm_conv3x3_s16_f32, include/ddnm_hip.h)."""
import numpy as np
import torch

from ddnm_amd import ops


def test_weight_scale_is_a_power_of_two_that_normalises_the_largest_weight():
    for m in (3e-5, 0.02, 0.7, 1.0, 40.0, 1e4):
        w = torch.tensor([m, -m / 3, m / 1000])
        s = ops.s16_weight_scale(w)
        assert np.log2(s) == round(np.log2(s))
        assert 2.0 ** 13 <= m * s < 2.0 ** 14
    assert ops.s16_weight_scale(torch.zeros(4)) == 1.0
    # a launch's 3x3 kernel and fused shortcut share one accumulator, hence one scale over both
    assert ops.s16_weight_scale(torch.tensor([0.01]), torch.tensor([3.0])) == ops.s16_weight_scale(torch.tensor([3.0]))


def test_split_packing_layout_and_reconstruction():
    g = torch.Generator().manual_seed(0)
    w = torch.randn(96, 64, 3, 3, generator=g) * 0.03
    w[0, 0, 0, 0] = 0.4                                   # a large weight next to small ones
    w[1, 1, 1, 1] = 1e-7                                  # and a negligible one
    s = ops.s16_weight_scale(w)
    p = ops.pack_conv_weight_s16(w, s)
    assert p.dtype == torch.float16 and tuple(p.shape) == (128, 9, 2, 2, 32)      # Cout padded to 128, [hi | lo] x 32
    assert p.numel() * 2 == 128 * 9 * 64 * 4                                      # the byte size of the fp32 packing
    ref = ops.pack_conv_weight(w).double()                                        # [128][9][64], (O, ky, kx, I)
    hi, lo = p[:, :, :, 0].double().reshape(128, 9, 64), p[:, :, :, 1].double().reshape(128, 9, 64)
    rec = (hi + lo) / s
    big = ref.abs() > ref.abs().max() * 2.0 ** -12
    assert ((rec - ref).abs()[big] / ref.abs()[big]).max() <= 2.0 ** -21           # fp32-grade for every weight that matters
    assert (rec - ref).abs()[~big].max() <= ref.abs().max() * 2.0 ** -33          # and absolutely tiny for the rest
    assert (p[96:] == 0).all()
    assert hi.abs().max() < 2.0 ** 14
    # channel order inside a chunk: chunk c holds input channels 32c .. 32c+31
    assert hi[3, 4, 40].item() == float(torch.tensor(w[3, 40, 1, 1].item() * s, dtype=torch.float32).half())


def _split(v):
    hi = v.astype(np.float16)
    lo = (v - hi.astype(np.float32)).astype(np.float16)
    return hi.astype(np.float64), lo.astype(np.float64)


def test_three_product_sum_is_fp32_grade():
    rng = np.random.default_rng(0)
    M, K, N = 256, 1152, 128
    x = rng.standard_normal((M, K))
    a = (x / (1 + np.exp(-x))).astype(np.float32)                        # swish(GroupNorm-like) activations
    w = (rng.standard_normal((K, N)) * 0.03).astype(np.float32)
    s = np.float32(ops.s16_weight_scale(torch.from_numpy(w)))
    ref = a.astype(np.float64) @ w.astype(np.float64)
    ah, al = _split(a)
    wh, wl = _split(w * s)
    y = ((ah @ wh + ah @ wl + al @ wh) / s).astype(np.float32).astype(np.float64)
    f32 = (a @ w).astype(np.float64)                                      # an fp32 dot product of the same length
    err = np.linalg.norm(y - ref) / np.linalg.norm(ref)
    err32 = np.linalg.norm(f32 - ref) / np.linalg.norm(ref)
    assert err < 1.5e-7 and err < err32
    # every fp16 x fp16 product is exact in fp32 (22 significant bits): the MFMA adds exact terms
    p = (ah[:8, :64].astype(np.float32)[:, :, None] * wh[:64, :8].astype(np.float32)[None]).astype(np.float64)
    assert (p == ah[:8, :64, None] * wh[None, :64, :8]).all()
    # without the two cross terms it is fp16 arithmetic
    assert np.linalg.norm(ah @ wh / s - ref) / np.linalg.norm(ref) > 1e-4
