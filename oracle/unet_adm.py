"""Restated forward of the ImageNet (ADM) noise predictor (TEST INFRASTRUCTURE).

Functional, state-dict driven CPU restatement, in fp32, of
  guided_diffusion/unet.py::UNetModel.forward              (:635-664)
  ... ResBlock._forward (FiLM scale-shift, up/down)          (:236-256)
  ... AttentionBlock._forward + QKVAttentionLegacy           (:299-305, :339-354)
  ... Upsample / Downsample without conv                     (:99-110, :133-140)
  guided_diffusion/nn.py::timestep_embedding ([cos, sin], /half)   (:103-121)
  guided_diffusion/nn.py::GroupNorm32 (fp32 statistics, eps = 1e-5) (:17-19, :93-100)
"""
import math

import torch
import torch.nn.functional as F

from . import weights

GN_EPS = 1e-5


def timestep_embedding(t, dim, max_period=10000):
    half = dim // 2
    freqs = torch.exp(-math.log(max_period) * torch.arange(0, half, dtype=torch.float32) / half)
    args = t[:, None].float() * freqs[None]
    emb = torch.cat([torch.cos(args), torch.sin(args)], dim=-1)
    if dim % 2:
        emb = torch.cat([emb, torch.zeros_like(emb[:, :1])], dim=-1)
    return emb


def _gn(sd, name, x):
    return F.group_norm(x.float(), 32, sd[name + ".weight"], sd[name + ".bias"], eps=GN_EPS).type(x.dtype)


def _res(sd, p, x, emb, mode, film):
    h = F.silu(_gn(sd, p + ".in_layers.0", x))
    if mode == "up":
        h = F.interpolate(h, scale_factor=2, mode="nearest")
        x = F.interpolate(x, scale_factor=2, mode="nearest")
    elif mode == "down":
        h = F.avg_pool2d(h, 2, 2)
        x = F.avg_pool2d(x, 2, 2)
    h = F.conv2d(h, sd[p + ".in_layers.2.weight"], sd[p + ".in_layers.2.bias"], padding=1)
    e = F.linear(F.silu(emb), sd[p + ".emb_layers.1.weight"], sd[p + ".emb_layers.1.bias"])[:, :, None, None]
    if film:
        scale, shift = torch.chunk(e, 2, dim=1)
        h = _gn(sd, p + ".out_layers.0", h) * (1 + scale) + shift
        h = F.silu(h)
    else:
        h = F.silu(_gn(sd, p + ".out_layers.0", h + e))
    h = F.conv2d(h, sd[p + ".out_layers.3.weight"], sd[p + ".out_layers.3.bias"], padding=1)
    if p + ".skip_connection.weight" in sd:
        x = F.conv2d(x, sd[p + ".skip_connection.weight"], sd[p + ".skip_connection.bias"])
    return x + h


def _attn(sd, p, x, head_ch):
    b, c, hh, ww = x.shape
    xf = x.reshape(b, c, -1)
    qkv = F.conv1d(_gn(sd, p + ".norm", xf), sd[p + ".qkv.weight"], sd[p + ".qkv.bias"])
    n_heads = c // head_ch
    length = xf.shape[-1]
    q, k, v = qkv.reshape(b * n_heads, head_ch * 3, length).split(head_ch, dim=1)     # legacy order
    scale = 1 / math.sqrt(math.sqrt(head_ch))
    w = torch.einsum("bct,bcs->bts", q * scale, k * scale)
    w = torch.softmax(w.float(), dim=-1).type(w.dtype)
    a = torch.einsum("bts,bcs->bct", w, v).reshape(b, -1, length)
    h = F.conv1d(a, sd[p + ".proj_out.weight"], sd[p + ".proj_out.bias"])
    return (xf + h).reshape(b, c, hh, ww)


def forward(sd, config, x, t, y=None):
    m = config.model
    film = m.use_scale_shift_norm
    emb = timestep_embedding(t, m.num_channels)
    emb = F.linear(emb, sd["time_embed.0.weight"], sd["time_embed.0.bias"])
    emb = F.linear(F.silu(emb), sd["time_embed.2.weight"], sd["time_embed.2.bias"])
    if m.class_cond:
        assert y is not None and y.shape == (x.shape[0],)
        emb = emb + sd["label_emb.weight"][y]
    inp, mid, out, _ = weights.adm_blocks(config)

    def run(prefix, layers, h):
        for j, L in enumerate(layers):
            p = f"{prefix}.{j}"
            if L[0] == "conv":
                h = F.conv2d(h, sd[p + ".weight"], sd[p + ".bias"], padding=1)
            elif L[0] == "res":
                h = _res(sd, p, h, emb, L[3], film)
            else:
                h = _attn(sd, p, h, m.num_head_channels)
        return h

    hs = []
    h = x.float()
    for i, layers in enumerate(inp):
        h = run(f"input_blocks.{i}", layers, h)
        hs.append(h)
    h = run("middle_block", mid, h)
    for i, layers in enumerate(out):
        h = run(f"output_blocks.{i}", layers, torch.cat([h, hs.pop()], dim=1))
    h = F.silu(_gn(sd, "out.0", h))
    return F.conv2d(h, sd["out.2.weight"], sd["out.2.bias"], padding=1)


class Net:
    """Callable with the reference's `model(xt, t[, classes])` protocol."""

    def __init__(self, sd, config):
        self.sd, self.config = sd, config

    def __call__(self, x, t, y=None):
        with torch.no_grad():
            return forward(self.sd, self.config, x, t, y)
