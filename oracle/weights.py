"""Deterministic seeded state-dicts with the reference's key names
(TEST INFRASTRUCTURE, see oracle/__init__.py).

No pretrained checkpoints exist offline, so parity
runs on seeded random weights.  The key lists restate the constructors
  * guided_diffusion/models.py:192-299   (celeba `Model`)
  * guided_diffusion/unet.py:396-633     (ImageNet `UNetModel`, built by
                                          script_util.py:130-185)
and are pinned against the real constructors' `state_dict()` in
tests/test_oracle_pins.py (golden key/shape lists in tests/golden/).

Every tensor -- including those the reference zero-initialises with
`zero_module` (nn.py:68) -- gets non-trivial values, otherwise the ADM UNet
outputs exactly 0 and parity is vacuous.
"""
import types
from collections import OrderedDict

import torch


def celeba_config(ch=128, ch_mult=(1, 1, 2, 2, 4, 4), num_res_blocks=2, attn_resolutions=(16,),
                  resolution=256, in_channels=3, out_ch=3):
    """Namespace shaped like the YAML config (configs/celeba_hq.yml:14-40)."""
    model = types.SimpleNamespace(type="simple", in_channels=in_channels, out_ch=out_ch, ch=ch,
                                  ch_mult=list(ch_mult), num_res_blocks=num_res_blocks,
                                  attn_resolutions=list(attn_resolutions), dropout=0.0,
                                  var_type="fixedsmall", ema_rate=0.999, ema=True,
                                  resamp_with_conv=True)
    data = types.SimpleNamespace(dataset="CelebA_HQ", image_size=resolution, channels=in_channels,
                                 logit_transform=False, uniform_dequantization=False,
                                 gaussian_dequantization=False, random_flip=True, rescaled=True,
                                 num_workers=0, out_of_dist=False, category="")
    diffusion = types.SimpleNamespace(beta_schedule="linear", beta_start=1e-4, beta_end=0.02,
                                      num_diffusion_timesteps=1000)
    sampling = types.SimpleNamespace(batch_size=1)
    tt = types.SimpleNamespace(T_sampling=100, travel_length=1, travel_repeat=1)
    return types.SimpleNamespace(model=model, data=data, diffusion=diffusion, sampling=sampling,
                                 time_travel=tt)


def _conv(shapes, name, cout, cin, k):
    shapes[name + ".weight"] = (cout, cin, k, k)
    shapes[name + ".bias"] = (cout,)


def _lin(shapes, name, cout, cin):
    shapes[name + ".weight"] = (cout, cin)
    shapes[name + ".bias"] = (cout,)


def _gn(shapes, name, c):
    shapes[name + ".weight"] = (c,)
    shapes[name + ".bias"] = (c,)


def _resblock(shapes, name, cin, cout, temb_ch):
    _gn(shapes, name + ".norm1", cin)
    _conv(shapes, name + ".conv1", cout, cin, 3)
    _lin(shapes, name + ".temb_proj", cout, temb_ch)
    _gn(shapes, name + ".norm2", cout)
    _conv(shapes, name + ".conv2", cout, cout, 3)
    if cin != cout:
        _conv(shapes, name + ".nin_shortcut", cout, cin, 1)


def _attn(shapes, name, c):
    _gn(shapes, name + ".norm", c)
    for p in ("q", "k", "v", "proj_out"):
        _conv(shapes, f"{name}.{p}", c, c, 1)


def celeba_shapes(config):
    """(name -> shape) in the registration order of models.py::Model.__init__."""
    m = config.model
    ch, mult, nrb = m.ch, tuple(m.ch_mult), m.num_res_blocks
    temb_ch = 4 * ch
    nres = len(mult)
    in_mult = (1,) + mult
    s = OrderedDict()
    _lin(s, "temb.dense.0", temb_ch, ch)
    _lin(s, "temb.dense.1", temb_ch, temb_ch)
    _conv(s, "conv_in", ch, m.in_channels, 3)
    res = config.data.image_size
    block_in = None
    for lvl in range(nres):
        block_in = ch * in_mult[lvl]
        block_out = ch * mult[lvl]
        names_attn = []
        for ib in range(nrb):
            _resblock(s, f"down.{lvl}.block.{ib}", block_in, block_out, temb_ch)
            block_in = block_out
            if res in m.attn_resolutions:
                names_attn.append(f"down.{lvl}.attn.{len(names_attn)}")
        # nn.Module registers `block` before `attn`: all blocks first, then attns
        for n in names_attn:
            _attn(s, n, block_in)
        if lvl != nres - 1:
            _conv(s, f"down.{lvl}.downsample.conv", block_in, block_in, 3)
            res //= 2
    _resblock(s, "mid.block_1", block_in, block_in, temb_ch)
    _attn(s, "mid.attn_1", block_in)
    _resblock(s, "mid.block_2", block_in, block_in, temb_ch)
    ups = {}
    for lvl in reversed(range(nres)):
        u = OrderedDict()
        block_out = ch * mult[lvl]
        skip_in = ch * mult[lvl]
        n_attn = 0
        for ib in range(nrb + 1):
            if ib == nrb:
                skip_in = ch * in_mult[lvl]
            _resblock(u, f"up.{lvl}.block.{ib}", block_in + skip_in, block_out, temb_ch)
            block_in = block_out
            if res in m.attn_resolutions:
                n_attn += 1
        for ia in range(n_attn):
            _attn(u, f"up.{lvl}.attn.{ia}", block_in)
        if lvl != 0:
            _conv(u, f"up.{lvl}.upsample.conv", block_in, block_in, 3)
            res *= 2
        ups[lvl] = u
    for lvl in range(nres):          # `self.up.insert(0, up)` -> stored by level index
        s.update(ups[lvl])
    _gn(s, "norm_out", block_in)
    _conv(s, "conv_out", m.out_ch, block_in, 3)
    return s


def fill(shapes, seed):
    """Seeded fp32 tensors: weights ~ N(0, 1/fan_in), GN affine ~ (1 + 0.1 N, 0.1 N),
    all biases (conv, linear, GN beta) ~ 0.05 N.  CPU mt19937 => identical on every box with this torch build."""
    g = torch.Generator().manual_seed(seed)
    sd = OrderedDict()
    for name, shape in shapes.items():
        if name.endswith(".weight") and len(shape) >= 2:
            fan_in = 1
            for d in shape[1:]:
                fan_in *= d
            t = torch.randn(shape, generator=g) * (1.0 / fan_in) ** 0.5
        elif name.endswith(".weight"):            # GroupNorm gamma
            t = 1.0 + 0.1 * torch.randn(shape, generator=g)
        else:
            t = 0.05 * torch.randn(shape, generator=g)
        sd[name] = t.float().contiguous()
    return sd


def celeba_state_dict(config, seed=1234):
    return fill(celeba_shapes(config), seed)


# ----------------------------------------------------------------------------------------------
# ADM UNet (guided_diffusion/unet.py::UNetModel built by script_util.create_model)
# ----------------------------------------------------------------------------------------------
def adm_config(image_size=256, num_channels=256, num_res_blocks=2, channel_mult="", attention_resolutions="32,16,8",
               num_head_channels=64, learn_sigma=True, class_cond=False, use_scale_shift_norm=True,
               resblock_updown=True):
    """Namespace shaped like configs/imagenet_256.yml:14-47 (`use_fp16` False: the oracle runs fp32)."""
    model = types.SimpleNamespace(type="openai", in_channels=3, out_channels=3, num_channels=num_channels,
                                  num_heads=4, num_res_blocks=num_res_blocks,
                                  attention_resolutions=attention_resolutions, dropout=0.0, resamp_with_conv=True,
                                  learn_sigma=learn_sigma, use_scale_shift_norm=use_scale_shift_norm, use_fp16=False,
                                  resblock_updown=resblock_updown, num_heads_upsample=-1, var_type="fixedsmall",
                                  num_head_channels=num_head_channels, image_size=image_size, class_cond=class_cond,
                                  use_new_attention_order=False, channel_mult=channel_mult)
    data = types.SimpleNamespace(dataset="ImageNet", image_size=image_size, channels=3, logit_transform=False,
                                 uniform_dequantization=False, gaussian_dequantization=False, random_flip=True,
                                 rescaled=True, num_workers=0, subset_1k=True, out_of_dist=False)
    diffusion = types.SimpleNamespace(beta_schedule="linear", beta_start=1e-4, beta_end=0.02,
                                      num_diffusion_timesteps=1000)
    sampling = types.SimpleNamespace(batch_size=1)
    tt = types.SimpleNamespace(T_sampling=100, travel_length=1, travel_repeat=1)
    return types.SimpleNamespace(model=model, data=data, diffusion=diffusion, sampling=sampling, time_travel=tt)


def adm_channel_mult(m):
    """script_util.py:148-160."""
    if m.channel_mult == "":
        return {512: (0.5, 1, 1, 2, 2, 4, 4), 256: (1, 1, 2, 2, 4, 4), 128: (1, 1, 2, 3, 4), 64: (1, 2, 3, 4)}[m.image_size]
    return tuple(int(v) for v in m.channel_mult.split(","))


def adm_blocks(config):
    """Block structure of UNetModel.__init__ (unet.py:470-617) as plain data:
    input[i] / middle / output[i] = list of layers, each ("conv", cin, cout) | ("res", cin, cout, mode)
    | ("attn", c) with mode in {"", "down", "up"}."""
    m = config.model
    mc = m.num_channels
    mult = adm_channel_mult(m)
    attn_ds = tuple(m.image_size // int(r) for r in m.attention_resolutions.split(","))
    ch = int(mult[0] * mc)
    inp = [[("conv", m.in_channels, ch)]]
    chans = [ch]
    ds = 1
    for level, mu in enumerate(mult):
        for _ in range(m.num_res_blocks):
            layers = [("res", ch, int(mu * mc), "")]
            ch = int(mu * mc)
            if ds in attn_ds:
                layers.append(("attn", ch))
            inp.append(layers)
            chans.append(ch)
        if level != len(mult) - 1:
            assert m.resblock_updown, "conv_resample Downsample path is not used by the ImageNet configs"
            inp.append([("res", ch, ch, "down")])
            chans.append(ch)
            ds *= 2
    mid = [("res", ch, ch, ""), ("attn", ch), ("res", ch, ch, "")]
    out = []
    for level, mu in list(enumerate(mult))[::-1]:
        for i in range(m.num_res_blocks + 1):
            ich = chans.pop()
            layers = [("res", ch + ich, int(mc * mu), "")]
            ch = int(mc * mu)
            if ds in attn_ds:
                layers.append(("attn", ch))
            if level and i == m.num_res_blocks:
                layers.append(("res", ch, ch, "up"))
                ds //= 2
            out.append(layers)
    return inp, mid, out, ch


def adm_shapes(config):
    m = config.model
    mc = m.num_channels
    ted = 4 * mc
    out_channels = 6 if m.learn_sigma else 3
    s = OrderedDict()
    _lin(s, "time_embed.0", ted, mc)
    _lin(s, "time_embed.2", ted, ted)
    if m.class_cond:
        s["label_emb.weight"] = (1000, ted)

    def layer(prefix, L):
        if L[0] == "conv":
            _conv(s, prefix, L[2], L[1], 3)
        elif L[0] == "res":
            _, cin, cout, _mode = L
            _gn(s, prefix + ".in_layers.0", cin)
            _conv(s, prefix + ".in_layers.2", cout, cin, 3)
            _lin(s, prefix + ".emb_layers.1", (2 * cout if m.use_scale_shift_norm else cout), ted)
            _gn(s, prefix + ".out_layers.0", cout)
            _conv(s, prefix + ".out_layers.3", cout, cout, 3)
            if cin != cout:
                _conv(s, prefix + ".skip_connection", cout, cin, 1)
        else:
            c = L[1]
            _gn(s, prefix + ".norm", c)
            s[prefix + ".qkv.weight"], s[prefix + ".qkv.bias"] = (3 * c, c, 1), (3 * c,)
            s[prefix + ".proj_out.weight"], s[prefix + ".proj_out.bias"] = (c, c, 1), (c,)

    inp, mid, out, ch = adm_blocks(config)
    for i, layers in enumerate(inp):
        for j, L in enumerate(layers):
            layer(f"input_blocks.{i}.{j}", L)
    for j, L in enumerate(mid):
        layer(f"middle_block.{j}", L)
    for i, layers in enumerate(out):
        for j, L in enumerate(layers):
            layer(f"output_blocks.{i}.{j}", L)
    _gn(s, "out.0", ch)
    _conv(s, "out.2", out_channels, ch, 3)
    return s


def adm_state_dict(config, seed=1234):
    """Every tensor random -- including the zero_module'd ones (nn.py:68), else eps == 0 identically."""
    return fill(adm_shapes(config), seed)


# ----------------------------------------------------------------------------------------------
# Noisy classifier (guided_diffusion/unet.py::EncoderUNetModel via script_util.create_classifier)
# ----------------------------------------------------------------------------------------------
def classifier_config(image_size=256, classifier_width=128, classifier_depth=2,
                      classifier_attention_resolutions="32,16,8", classifier_scale=1.0):
    """Keys of configs/imagenet_256_cc.yml:36-45 (classifier_use_fp16 False: the oracle runs fp32)."""
    return types.SimpleNamespace(image_size=image_size, classifier_use_fp16=False, classifier_width=classifier_width,
                                 classifier_depth=classifier_depth,
                                 classifier_attention_resolutions=classifier_attention_resolutions,
                                 classifier_use_scale_shift_norm=True, classifier_resblock_updown=True,
                                 classifier_pool="attention", classifier_scale=classifier_scale)


def classifier_blocks(cc):
    """input_blocks / middle_block of EncoderUNetModel.__init__ (unet.py:737-815) as plain data (cf. adm_blocks)."""
    mult = {512: (0.5, 1, 1, 2, 2, 4, 4), 256: (1, 1, 2, 2, 4, 4), 128: (1, 1, 2, 3, 4), 64: (1, 2, 3, 4),
            32: (1, 2)}[cc.image_size]      # 32: reduced test size only
    mc = cc.classifier_width
    attn_ds = tuple(cc.image_size // int(r) for r in cc.classifier_attention_resolutions.split(","))
    ch = int(mult[0] * mc)
    inp = [[("conv", 3, ch)]]
    ds = 1
    for level, mu in enumerate(mult):
        for _ in range(cc.classifier_depth):
            layers = [("res", ch, int(mu * mc), "")]
            ch = int(mu * mc)
            if ds in attn_ds:
                layers.append(("attn", ch))
            inp.append(layers)
        if level != len(mult) - 1:
            inp.append([("res", ch, ch, "down")])
            ds *= 2
    mid = [("res", ch, ch, ""), ("attn", ch), ("res", ch, ch, "")]
    return inp, mid, ch, cc.image_size // ds


def classifier_shapes(cc):
    mc = cc.classifier_width
    ted = 4 * mc
    s = OrderedDict()
    _lin(s, "time_embed.0", ted, mc)
    _lin(s, "time_embed.2", ted, ted)

    def layer(prefix, L):
        if L[0] == "conv":
            _conv(s, prefix, L[2], L[1], 3)
        elif L[0] == "res":
            _, cin, cout, _mode = L
            _gn(s, prefix + ".in_layers.0", cin)
            _conv(s, prefix + ".in_layers.2", cout, cin, 3)
            _lin(s, prefix + ".emb_layers.1", 2 * cout, ted)
            _gn(s, prefix + ".out_layers.0", cout)
            _conv(s, prefix + ".out_layers.3", cout, cout, 3)
            if cin != cout:
                _conv(s, prefix + ".skip_connection", cout, cin, 1)
        else:
            c = L[1]
            _gn(s, prefix + ".norm", c)
            s[prefix + ".qkv.weight"], s[prefix + ".qkv.bias"] = (3 * c, c, 1), (3 * c,)
            s[prefix + ".proj_out.weight"], s[prefix + ".proj_out.bias"] = (c, c, 1), (c,)

    inp, mid, ch, sp = classifier_blocks(cc)
    for i, layers in enumerate(inp):
        for j, L in enumerate(layers):
            layer(f"input_blocks.{i}.{j}", L)
    for j, L in enumerate(mid):
        layer(f"middle_block.{j}", L)
    _gn(s, "out.0", ch)
    s["out.2.positional_embedding"] = (ch, sp * sp + 1)
    s["out.2.qkv_proj.weight"], s["out.2.qkv_proj.bias"] = (3 * ch, ch, 1), (3 * ch,)
    s["out.2.c_proj.weight"], s["out.2.c_proj.bias"] = (1000, ch, 1), (1000,)
    return s


def classifier_state_dict(cc, seed=4321):
    shapes = classifier_shapes(cc)
    sd = fill(shapes, seed)
    g = torch.Generator().manual_seed(seed + 1)
    pe = shapes["out.2.positional_embedding"]
    sd["out.2.positional_embedding"] = (torch.randn(pe, generator=g) / pe[0] ** 0.5).contiguous()
    return sd
