"""Restated forward of the CelebA-HQ noise predictor (TEST INFRASTRUCTURE).

Functional, state-dict driven CPU restatement of
  guided_diffusion/models.py::Model.forward            (:301-341)
  ... ResnetBlock.forward                               (:115-134)
  ... AttnBlock.forward                                 (:165-189)
  ... Downsample / Upsample                             (:36-75)
  ... get_timestep_embedding ([sin, cos], /(half-1))    (:6-24)
  ... Normalize = GroupNorm(32, C, eps=1e-6)            (:32-33)
torch-CPU fp32 is the oracle arithmetic.
"""
import math

import torch
import torch.nn.functional as F

GN_EPS = 1e-6


def timestep_embedding(t, dim):
    half = dim // 2
    freq = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10000) / (half - 1)))
    arg = t.float()[:, None] * freq[None, :]
    emb = torch.cat([torch.sin(arg), torch.cos(arg)], dim=1)
    if dim % 2 == 1:
        emb = F.pad(emb, (0, 1, 0, 0))
    return emb


def swish(x):
    return x * torch.sigmoid(x)


def _gn(sd, name, x):
    return F.group_norm(x, 32, sd[name + ".weight"], sd[name + ".bias"], eps=GN_EPS)


def _conv(sd, name, x, stride=1, padding=0):
    return F.conv2d(x, sd[name + ".weight"], sd[name + ".bias"], stride=stride, padding=padding)


def _resblock(sd, name, x, temb):
    h = _conv(sd, name + ".conv1", swish(_gn(sd, name + ".norm1", x)), padding=1)
    h = h + F.linear(swish(temb), sd[name + ".temb_proj.weight"], sd[name + ".temb_proj.bias"])[:, :, None, None]
    h = _conv(sd, name + ".conv2", swish(_gn(sd, name + ".norm2", h)), padding=1)
    if name + ".nin_shortcut.weight" in sd:
        x = _conv(sd, name + ".nin_shortcut", x)
    return x + h


def _attn(sd, name, x):
    h = _gn(sd, name + ".norm", x)
    q, k, v = (_conv(sd, f"{name}.{p}", h) for p in ("q", "k", "v"))
    b, c, hh, ww = q.shape
    q = q.reshape(b, c, hh * ww).permute(0, 2, 1)        # b, T, c
    k = k.reshape(b, c, hh * ww)                         # b, c, T
    w = torch.bmm(q, k) * (int(c) ** (-0.5))             # b, Tq, Tk
    w = F.softmax(w, dim=2)
    v = v.reshape(b, c, hh * ww)
    h = torch.bmm(v, w.permute(0, 2, 1)).reshape(b, c, hh, ww)
    return x + _conv(sd, name + ".proj_out", h)


def forward(sd, config, x, t):
    """eps = Model(x, t); x [B,C,R,R] fp32, t [B] float."""
    m = config.model
    ch, mult, nrb = m.ch, tuple(m.ch_mult), m.num_res_blocks
    nres = len(mult)
    temb = timestep_embedding(t, ch)
    temb = F.linear(temb, sd["temb.dense.0.weight"], sd["temb.dense.0.bias"])
    temb = F.linear(swish(temb), sd["temb.dense.1.weight"], sd["temb.dense.1.bias"])

    hs = [_conv(sd, "conv_in", x, padding=1)]
    for lvl in range(nres):
        has_attn = f"down.{lvl}.attn.0.norm.weight" in sd
        for ib in range(nrb):
            h = _resblock(sd, f"down.{lvl}.block.{ib}", hs[-1], temb)
            if has_attn:
                h = _attn(sd, f"down.{lvl}.attn.{ib}", h)
            hs.append(h)
        if lvl != nres - 1:
            h = F.pad(hs[-1], (0, 1, 0, 1))
            hs.append(_conv(sd, f"down.{lvl}.downsample.conv", h, stride=2))
    h = hs[-1]
    h = _resblock(sd, "mid.block_1", h, temb)
    h = _attn(sd, "mid.attn_1", h)
    h = _resblock(sd, "mid.block_2", h, temb)
    for lvl in reversed(range(nres)):
        has_attn = f"up.{lvl}.attn.0.norm.weight" in sd
        for ib in range(nrb + 1):
            h = _resblock(sd, f"up.{lvl}.block.{ib}", torch.cat([h, hs.pop()], dim=1), temb)
            if has_attn:
                h = _attn(sd, f"up.{lvl}.attn.{ib}", h)
        if lvl != 0:
            h = F.interpolate(h, scale_factor=2.0, mode="nearest")
            h = _conv(sd, f"up.{lvl}.upsample.conv", h, padding=1)
    h = swish(_gn(sd, "norm_out", h))
    return _conv(sd, "conv_out", h, padding=1)


class Net:
    """Callable with the reference's `model(xt, t)` protocol."""

    def __init__(self, sd, config):
        self.sd, self.config = sd, config

    def __call__(self, x, t):
        with torch.no_grad():
            return forward(self.sd, self.config, x, t)
