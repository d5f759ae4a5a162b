"""Restated noisy classifier and classifier-guidance gradient (TEST INFRASTRUCTURE).

  * forward   follows guided_diffusion/unet.py::EncoderUNetModel.forward (:872-895) with pool="attention"
              (AttentionPool2d :22-51, QKVAttention (new order) :361-393), reusing the ResBlock /
              AttentionBlock restatements of oracle/unet_adm.py (same classes in the reference);
  * cond_fn   follows guided_diffusion/diffusion.py:183-189:
              classifier_scale * d/dx log_softmax(classifier(x, t))[y]   (torch autograd on CPU).
"""
import math

import torch
import torch.nn.functional as F

from . import unet_adm, weights


def attention_pool(sd, p, x, head_ch):
    b, c = x.shape[:2]
    x = x.reshape(b, c, -1)
    x = torch.cat([x.mean(dim=-1, keepdim=True), x], dim=-1)
    x = x + sd[p + ".positional_embedding"][None]
    qkv = F.conv1d(x, sd[p + ".qkv_proj.weight"], sd[p + ".qkv_proj.bias"])
    n_heads = c // head_ch
    length = qkv.shape[-1]
    q, k, v = qkv.chunk(3, dim=1)                                      # new attention order
    scale = 1 / math.sqrt(math.sqrt(head_ch))
    w = torch.einsum("bct,bcs->bts", (q * scale).reshape(b * n_heads, head_ch, length),
                     (k * scale).reshape(b * n_heads, head_ch, length))
    w = torch.softmax(w.float(), dim=-1)
    a = torch.einsum("bts,bcs->bct", w, v.reshape(b * n_heads, head_ch, length)).reshape(b, -1, length)
    out = F.conv1d(a, sd[p + ".c_proj.weight"], sd[p + ".c_proj.bias"])
    return out[:, :, 0]


def forward(sd, cc, x, t):
    """logits [B, 1000]."""
    emb = unet_adm.timestep_embedding(t, cc.classifier_width)
    emb = F.linear(emb, sd["time_embed.0.weight"], sd["time_embed.0.bias"])
    emb = F.linear(F.silu(emb), sd["time_embed.2.weight"], sd["time_embed.2.bias"])
    inp, mid, ch, sp = weights.classifier_blocks(cc)

    def run(prefix, layers, h):
        for j, L in enumerate(layers):
            p = f"{prefix}.{j}"
            if L[0] == "conv":
                h = F.conv2d(h, sd[p + ".weight"], sd[p + ".bias"], padding=1)
            elif L[0] == "res":
                h = unet_adm._res(sd, p, h, emb, L[3], True)
            else:
                h = unet_adm._attn(sd, p, h, 64)
        return h

    h = x.float()
    for i, layers in enumerate(inp):
        h = run(f"input_blocks.{i}", layers, h)
    h = run("middle_block", mid, h)
    h = F.silu(unet_adm._gn(sd, "out.0", h))
    return attention_pool(sd, "out.2", h, 64)


def cond_fn(sd, cc, x, t, y):
    """classifier_scale * grad_x log p(y | x, t)  (diffusion.py:183-189)."""
    with torch.enable_grad():
        x_in = x.detach().requires_grad_(True)
        logits = forward(sd, cc, x_in, t)
        log_probs = F.log_softmax(logits, dim=-1)
        selected = log_probs[range(len(logits)), y.view(-1)]
        return torch.autograd.grad(selected.sum(), x_in)[0] * cc.classifier_scale
