"""Noisy ImageNet classifier + classifier-guidance gradient on MI355X.

Drop-in for `script_util.create_classifier(...)` -> `guided_diffusion/unet.py::EncoderUNetModel`
(:684-895, pool="attention") and for the `cond_fn` closure of `guided_diffusion/diffusion.py:183-189`:

    cond_fn(x, t, y) = classifier_scale * d/dx log_softmax(classifier(x, t))[y]

The reference gets the gradient from torch autograd.  Here the backward pass is explicit and made of HIP
kernels only (no autograd, no torch ops): data-gradient convolutions are the forward implicit-GEMM kernels
run on flipped / transposed weights, GroupNorm(+FiLM)+SiLU backward is `ddnm_gn_bwd_f32`, attention
backward is four batched MFMA GEMMs + `softmax_bwd_rows`, AttentionPool2d has its own small kernels
(csrc/backward.hip).  Weight gradients are never formed.  The state-dict (249 tensors, reference names,
e.g. the public 256x256_classifier.pt) loads unchanged.

Two engines: fp32 tensors (default; `convert_to_fp16()` under DDNM_CLS_GEN1=1 adds fp16 MFMA operands for the 3x3 / 1x1
convolutions only) and, since round 5, the **fp16-activation path** `convert_to_fp16()` selects by default -- what the
reference's `classifier_use_fp16: true` (configs/imagenet_256_cc.yml:44, unet.py:817-823) means: every activation AND
every activation gradient in HBM is fp16 NHWC, convolutions and their data gradients are `ddnm_conv16`, attention is
the fused `ddnm_attn16_d64` with a fused backward (`ddnm_attn16_d64_bwd`: no [T][T] tensor), GroupNorm backward is
`ddnm_gn_bwd_h16`; statistics, softmax, the attention pool and the embeddings stay fp32.
"""
import math
import os
from collections import OrderedDict

import torch

from .. import _lib, ops
from .._lib import check

GN_EPS = 1e-5
CIN_PAD = 32
GRAD_SCALE_FP16 = 1024.0        # see EncoderUNetModel.convert_to_fp16


def _p(t):
    return None if t is None else t.data_ptr()


def classifier_defaults():
    """script_util.py:27-39."""
    return dict(image_size=64, classifier_use_fp16=False, classifier_width=128, classifier_depth=2,
                classifier_attention_resolutions="32,16,8", classifier_use_scale_shift_norm=True,
                classifier_resblock_updown=True, classifier_pool="attention")


def args_to_dict(args, keys):
    return {k: getattr(args, k) for k in keys}


def create_classifier(image_size, classifier_use_fp16, classifier_width, classifier_depth,
                      classifier_attention_resolutions, classifier_use_scale_shift_norm, classifier_resblock_updown,
                      classifier_pool):
    """script_util.py:229-267."""
    table = {512: (0.5, 1, 1, 2, 2, 4, 4), 256: (1, 1, 2, 2, 4, 4), 128: (1, 1, 2, 3, 4), 64: (1, 2, 3, 4)}
    if image_size not in table:
        raise ValueError(f"unsupported image size: {image_size}")
    attention_ds = tuple(image_size // int(r) for r in classifier_attention_resolutions.split(","))
    return EncoderUNetModel(image_size=image_size, in_channels=3, model_channels=classifier_width, out_channels=1000,
                            num_res_blocks=classifier_depth, attention_resolutions=attention_ds,
                            channel_mult=table[image_size], use_fp16=classifier_use_fp16, num_head_channels=64,
                            use_scale_shift_norm=classifier_use_scale_shift_norm,
                            resblock_updown=classifier_resblock_updown, pool=classifier_pool)


# module attribute (tests): False evaluates the timestep-independent prefix of a grouped pass for every replica
SHARE_PREFIX = True

class EncoderUNetModel:
    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                 channel_mult=(1, 2, 4, 8), use_fp16=False, num_head_channels=-1, use_scale_shift_norm=False,
                 resblock_updown=False, pool="adaptive", device=None, **kwargs):
        if pool != "attention" or not use_scale_shift_norm or not resblock_updown or num_head_channels == -1:
            raise NotImplementedError("only the DDNM classifier configuration (attention pool, FiLM, resblock "
                                      "up/down, 64-channel heads) is built")
        self.image_size, self.in_channels, self.model_channels = image_size, in_channels, model_channels
        self.out_channels, self.head_ch = out_channels, num_head_channels
        self.time_embed_dim = 4 * model_channels
        self.device = torch.device("cuda") if device is None else torch.device(device)
        mc = model_channels
        ch = int(channel_mult[0] * mc)
        self.input_blocks = [[("conv", in_channels, ch)]]
        ds = 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers = [("res", ch, int(mult * mc), "")]
                ch = int(mult * mc)
                if ds in attention_resolutions:
                    layers.append(("attn", ch))
                self.input_blocks.append(layers)
            if level != len(channel_mult) - 1:
                self.input_blocks.append([("res", ch, ch, "down")])
                ds *= 2
        self.middle_block = [("res", ch, ch, ""), ("attn", ch), ("res", ch, ch, "")]
        self.final_ch, self.pool_sp = ch, image_size // ds
        off, self._film_off = 0, {}
        for prefix, layers in self._walk():
            for j, L in enumerate(layers):
                if L[0] == "res":
                    self._film_off[f"{prefix}.{j}"] = off
                    off += 2 * L[2]
        self.film_total = off
        self.w = None
        self.use_fp16 = False          # set by convert_to_fp16(), like the reference's runner (diffusion.py:176-177)
        self.h16 = False               # ... which selects the fp16-activation path unless DDNM_CLS_GEN1=1
        self._ws = None

    def _walk(self):
        for i, layers in enumerate(self.input_blocks):
            yield f"input_blocks.{i}", layers
        yield "middle_block", self.middle_block

    @property
    def max_group_batch(self):
        """Images one pass may hold: the largest activation (model_channels at image_size^2) must stay below the 2 GiB a
        convolution launch can address -- 4 bytes per element for the fp32-tensor engines, 2 for the fp16-activation one."""
        per_image = self.image_size * self.image_size * self.model_channels * (2 if (self.use_fp16 and self.h16) else 4)
        return ops.max_launch_batch(per_image)    # half of the addressable limit: 32 / 64 at 256 x 256 x 128

    # ------------------------------------------------------------------ nn.Module-like surface
    def to(self, device):
        return self

    def eval(self):
        return self

    def convert_to_fp16(self):
        """`classifier.convert_to_fp16()` (diffusion.py:176-177, unet.py:817-823): the torso runs the fp16-ACTIVATION
        path (`_forward_h16` / `_backward_h16`): activations and activation gradients fp16 NHWC in HBM, convolutions and
        data-gradient convolutions on `ddnm_conv16`, fused attention forward / backward, fp32 GroupNorm statistics,
        softmax, pool and embeddings.  The backward pass is linear in d(logits), so it is evaluated on 2^10 * d(logits)
        and rescaled at the end: activation gradients of ~1e-6 would otherwise fall below fp16's normal range.
        DDNM_CLS_GEN1=1 selects the first-generation form (fp32 tensors, fp16 MFMA operands for the convolutions)."""
        self.use_fp16 = True
        self.h16 = os.environ.get("DDNM_CLS_GEN1") != "1"
        if self.w is not None:
            self._pack_h16() if self.h16 else self._pack_f16()
        return self

    def convert_to_fp32(self):
        self.use_fp16 = False
        self.h16 = False
        return self

    def _pack_h16(self):
        """fp16 (O,ky,kx,I) weights of the fp16-activation path, forward and data-gradient (input / output channels
        swapped, taps flipped) forms, Cout padded to 256 rows (ops.pack_conv_weight16); built on first use from the HOST
        copy of the checkpoint."""
        w = self.w
        if "h16.ready" in w:
            return
        for name, raw in self._raw_host.items():
            raw = raw.to(device=self.device, dtype=torch.float32)
            if raw.dim() == 3:
                raw = raw.unsqueeze(-1)
            w[name + ".h16"] = ops.pack_conv_weight16(raw, cin_pad=(64 if raw.shape[1] < 64 else None))
            w[name + ".dgrad.h16"] = ops.pack_conv_weight16(raw.permute(1, 0, 2, 3).flip(2, 3).contiguous())
            if name.endswith(".skip_connection"):
                p16 = w[name + ".h16"]
                w[name + ".h16.flat"] = p16.reshape(p16.shape[0], -1).contiguous()   # the fused shortcut's [Cout][Cin]
        w["h16.ready"] = True

    def _pack_f16(self):
        for key, raw in self._raw.items():
            if key + ".f16" not in self.w:
                self.w[key + ".f16"] = ops.pack_conv_weight_f16(raw)

    def _w16(self, key):
        return self.w.get(key + ".f16") if self.use_fp16 else None

    def parameters(self):
        return iter(())

    def state_dict_shapes(self):
        s = OrderedDict()
        ted, mc = self.time_embed_dim, self.model_channels
        s["time_embed.0.weight"], s["time_embed.0.bias"] = (ted, mc), (ted,)
        s["time_embed.2.weight"], s["time_embed.2.bias"] = (ted, ted), (ted,)
        for prefix, layers in self._walk():
            for j, L in enumerate(layers):
                n = f"{prefix}.{j}"
                if L[0] == "conv":
                    s[n + ".weight"], s[n + ".bias"] = (L[2], L[1], 3, 3), (L[2],)
                elif L[0] == "res":
                    cin, cout = L[1], L[2]
                    s[n + ".in_layers.0.weight"], s[n + ".in_layers.0.bias"] = (cin,), (cin,)
                    s[n + ".in_layers.2.weight"], s[n + ".in_layers.2.bias"] = (cout, cin, 3, 3), (cout,)
                    s[n + ".emb_layers.1.weight"], s[n + ".emb_layers.1.bias"] = (2 * cout, ted), (2 * cout,)
                    s[n + ".out_layers.0.weight"], s[n + ".out_layers.0.bias"] = (cout,), (cout,)
                    s[n + ".out_layers.3.weight"], s[n + ".out_layers.3.bias"] = (cout, cout, 3, 3), (cout,)
                    if cin != cout:
                        s[n + ".skip_connection.weight"], s[n + ".skip_connection.bias"] = (cout, cin, 1, 1), (cout,)
                else:
                    c = L[1]
                    s[n + ".norm.weight"], s[n + ".norm.bias"] = (c,), (c,)
                    s[n + ".qkv.weight"], s[n + ".qkv.bias"] = (3 * c, c, 1), (3 * c,)
                    s[n + ".proj_out.weight"], s[n + ".proj_out.bias"] = (c, c, 1), (c,)
        c = self.final_ch
        s["out.0.weight"], s["out.0.bias"] = (c,), (c,)
        s["out.2.positional_embedding"] = (c, self.pool_sp ** 2 + 1)
        s["out.2.qkv_proj.weight"], s["out.2.qkv_proj.bias"] = (3 * c, c, 1), (3 * c,)
        s["out.2.c_proj.weight"], s["out.2.c_proj.bias"] = (self.out_channels, c, 1), (self.out_channels,)
        return s

    def load_state_dict(self, sd, strict=True):
        dev = self.device
        g = lambda k: sd[k].detach().to(device=dev, dtype=torch.float32).contiguous()   # noqa: E731
        w = {}
        self._raw = {}
        self._raw_host = {}            # conv name -> HOST weight of the checkpoint (packed on demand by _pack_h16)

        def conv(name, raw, cin_pad=None):
            """forward weights and the data-gradient weights (input/output channels swapped, taps flipped)"""
            self._raw_host[name] = sd[name + ".weight"].detach()
            if raw.dim() == 3:
                raw = raw.unsqueeze(-1)
            w[name + ".weight"] = ops.pack_conv_weight(raw, cin_pad=cin_pad)
            rawT = raw.permute(1, 0, 2, 3).flip(2, 3).contiguous()
            w[name + ".dgrad"] = ops.pack_conv_weight(rawT)
            w[name + ".bias"] = g(name + ".bias")
            # candidates of the fp16-operand kernels (Cin % 64 == 0, Cout % 128 == 0): the 3x3 halo kernel and, since
            # round 4, the 1x1 GEMM kernel for skip_connection / qkv / proj_out and their data gradients (the reference's
            # fp16 classifier runs those convolutions in fp16 as well, unet.py:817-823)
            if raw.shape[1] % 64 == 0 and raw.shape[0] % 128 == 0:
                self._raw[name + ".weight"] = raw
            if rawT.shape[1] % 64 == 0 and rawT.shape[0] % 128 == 0:
                self._raw[name + ".dgrad"] = rawT

        for k in ("time_embed.0", "time_embed.2"):
            w[k + ".weight"], w[k + ".bias"] = g(k + ".weight"), g(k + ".bias")
        fw, fb = [], []
        for prefix, layers in self._walk():
            for j, L in enumerate(layers):
                n = f"{prefix}.{j}"
                if L[0] == "conv":
                    conv(n, g(n + ".weight"), cin_pad=CIN_PAD)
                    if 9 * L[1] <= CIN_PAD:       # 3 input channels: the 27 taps fit one 32-wide K chunk (see models.py)
                        w[n + ".weight.im2col"] = ops.pack_conv_in_weight_im2col(g(n + ".weight"), CIN_PAD)
                elif L[0] == "res":
                    for norm in ("in_layers.0", "out_layers.0"):
                        w[f"{n}.{norm}.weight"], w[f"{n}.{norm}.bias"] = g(f"{n}.{norm}.weight"), g(f"{n}.{norm}.bias")
                    conv(n + ".in_layers.2", g(n + ".in_layers.2.weight"))
                    conv(n + ".out_layers.3", g(n + ".out_layers.3.weight"))
                    fw.append(g(n + ".emb_layers.1.weight"))
                    fb.append(g(n + ".emb_layers.1.bias"))
                    if L[1] != L[2]:
                        conv(n + ".skip_connection", g(n + ".skip_connection.weight"))
                        w[n + ".out_plus_skip.bias"] = (g(n + ".out_layers.3.bias") + g(n + ".skip_connection.bias")).contiguous()
                else:
                    w[n + ".norm.weight"], w[n + ".norm.bias"] = g(n + ".norm.weight"), g(n + ".norm.bias")
                    conv(n + ".qkv", g(n + ".qkv.weight"))
                    conv(n + ".proj_out", g(n + ".proj_out.weight"))
        w["film_cat.weight"], w["film_cat.bias"] = torch.cat(fw, 0).contiguous(), torch.cat(fb, 0).contiguous()
        w["out.0.weight"], w["out.0.bias"] = g("out.0.weight"), g("out.0.bias")
        w["pool.pos"] = g("out.2.positional_embedding")
        wq = g("out.2.qkv_proj.weight").squeeze(-1).contiguous()              # [3C, C]
        w["pool.qkv.weight"], w["pool.qkv.bias"] = wq, g("out.2.qkv_proj.bias")
        wc = g("out.2.c_proj.weight").squeeze(-1).contiguous()                # [1000, C]
        w["pool.c.weight"], w["pool.c.weight_t"], w["pool.c.bias"] = wc, wc.t().contiguous(), g("out.2.c_proj.bias")
        half = self.model_channels // 2
        w["time.freq"] = torch.exp(-math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half).to(dev)
        self.w = w
        self._ws = None
        if self.use_fp16:
            self._pack_h16() if self.h16 else self._pack_f16()
        return self

    # ------------------------------------------------------------------ forward (optionally recording a tape)
    def _workspace(self, B):
        if self._ws is None or self._ws_B < B:
            res, mp, mpb = self.image_size, 0, 0
            while res >= 4:
                for c in (self.model_channels, self.final_ch):
                    mp = max(mp, ops.gn_nchunk(res * res, c))
                    mpb = max(mpb, _lib.lib().ddnm_gn_bwd_nchunk(res * res, c))
                res //= 2
            self._ws = ops.GroupNormWorkspace(self.device, B, self.final_ch, B * mp * 32 * 2)
            self._bwd_partial = torch.empty(B * mpb * 32 * 2, dtype=torch.float64, device=self.device)
            self._bwd_coef = torch.empty(B * 32 * 2, dtype=torch.float32, device=self.device)
            self._ws_B = B
        return self._ws

    def _gn(self, x, name, film=None, keep=None):
        f, fs = (None, 0) if film is None else (film, self.film_total)
        return ops.group_norm_affine(x, None, self.w[name + ".weight"], self.w[name + ".bias"], GN_EPS, self._ws,
                                     film=f, film_stride=fs, keep=keep)

    def _res(self, n, L, x, film_all, tape):
        w = self.w
        cin, cout, mode = L[1], L[2], L[3]
        k1 = {} if tape is not None else None
        k2 = {} if tape is not None else None
        gn1 = self._gn(x, n + ".in_layers.0", keep=k1)
        if mode == "down":
            hp = ops.avgpool2_nhwc(x.t, gn=gn1, silu=True)
            xs = ops.avgpool2_nhwc(x.t)
            h = ops.conv2d(hp, w[n + ".in_layers.2.weight"], cout, 3, bias=w[n + ".in_layers.2.bias"], emit_stats=True,
                           weight_f16=self._w16(n + ".in_layers.2.weight"))
        else:
            h = ops.conv2d(x, w[n + ".in_layers.2.weight"], cout, 3, gn=gn1, gn_silu=True,
                           bias=w[n + ".in_layers.2.bias"], emit_stats=True,
                           weight_f16=self._w16(n + ".in_layers.2.weight"))
            xs = x.t if cin == cout else ops.conv2d(x, w[n + ".skip_connection.weight"], cout, 1,
                                                    bias=w[n + ".skip_connection.bias"],
                                                    weight_f16=self._w16(n + ".skip_connection.weight"))
        gn2 = self._gn(h, n + ".out_layers.0", film=film_all[:, self._film_off[n]:], keep=k2)
        out = ops.conv2d(h, w[n + ".out_layers.3.weight"], cout, 3, gn=gn2, gn_silu=True,
                         bias=w[n + ".out_layers.3.bias"], res=xs, emit_stats=True,
                         weight_f16=self._w16(n + ".out_layers.3.weight"))
        if tape is not None:
            tape.append(("res", n, L, x.t, h.t, k1, k2))
        return out

    def _attn(self, n, x, tape):
        w = self.w
        B, H, W, C = x.t.shape
        T, hc = H * W, self.head_ch
        nh = C // hc
        k = {} if tape is not None else None
        gn = self._gn(x, n + ".norm", keep=k)
        qkv = ops.conv2d(x, w[n + ".qkv.weight"], 3 * C, 1, gn=gn, gn_silu=False, bias=w[n + ".qkv.bias"],
                         weight_f16=self._w16(n + ".qkv.weight"))
        flat = qkv.view(-1)
        S = torch.empty(B * nh, T, T, dtype=torch.float32, device=qkv.device)
        ops.bgemm(flat, flat[hc:], S, T, T, hc, lda=3 * C, ldb=3 * C, ldc=T, transb=True, batch=B * nh, inner=nh,
                  sA=(T * 3 * C, 3 * hc), sB=(T * 3 * C, 3 * hc), sC=(nh * T * T, T * T))
        ops.softmax_rows_(S, B * nh * T, T, T, 1.0 / math.sqrt(hc))
        o = torch.empty(B, H, W, C, dtype=torch.float32, device=qkv.device)
        ops.bgemm(S, flat[2 * hc:], o, T, hc, T, lda=T, ldb=3 * C, ldc=C, transb=False, batch=B * nh, inner=nh,
                  sA=(nh * T * T, T * T), sB=(T * 3 * C, 3 * hc), sC=(T * C, hc))
        out = ops.conv2d(o, w[n + ".proj_out.weight"], C, 1, bias=w[n + ".proj_out.bias"], res=x, emit_stats=True,
                         weight_f16=self._w16(n + ".proj_out.weight"))
        if tape is not None:
            tape.append(("attn", n, x.t, qkv, S, k))
        return out

    # ------------------------------------------------------------------ fp16-activation forward (csrc/conv16.hip)
    def _conv3_16(self, key, cout, x, gn, **kw):
        """3x3 convolution (or data-gradient convolution) of an fp16 NHWC tensor on ddnm_conv16, the GroupNorm affine +
        swish of `gn` fused into its loader; images too small for a pixel tile (8 x 8) go through im2col + one GEMM."""
        w16 = self.w[key]
        t = x.t if isinstance(x, ops.Act) else x
        B, H, W, cin = t.shape
        if ops.conv16_supported(B, H, W, cin, cout, 3):
            return ops.conv16(x, w16, cout, 3, gn=gn, gn_silu=True, **kw)
        col = ops.im2col16(x, None, gn, True)
        return ops.conv16(col, w16.reshape(w16.shape[0], 1, -1), cout, 1, **kw)

    def _res_h16(self, n, L, x, film_all, tape, shared=None):
        w = self.w
        cin, cout, mode = L[1], L[2], L[3]
        k1 = {} if tape is not None else None
        k2 = {} if tape is not None else None
        b1, b2 = w[n + ".in_layers.2.bias"], w[n + ".out_layers.3.bias"]
        xs = None
        if shared is not None:      # t-independent half of the block, evaluated once for the replicas of a grouped pass
            h, k1s = shared
            if k1 is not None:
                k1.update(k1s)
            xs = x.t
        elif mode == "down":        # AvgPool2d on both branches (unet.py:237-242)
            gn1 = self._gn(x, n + ".in_layers.0", keep=k1)
            hp = ops.gn_apply16(x, None, gn1, True, pool=True)
            xs = ops.gn_apply16(x, None, None, False, pool=True)
            h = self._conv3_16(n + ".in_layers.2.h16", cout, hp, None, bias=b1)
        else:
            gn1 = self._gn(x, n + ".in_layers.0", keep=k1)
            h = self._conv3_16(n + ".in_layers.2.h16", cout, x, gn1, bias=b1)
            if cin == cout:
                xs = x.t
        gn2 = self._gn(h, n + ".out_layers.0", film=film_all[:, self._film_off[n]:], keep=k2)
        if xs is not None:
            out = self._conv3_16(n + ".out_layers.3.h16", cout, h, gn2, bias=b2, res=xs)
        else:
            B, H, W, _ = h.t.shape
            if ops.conv16_supported(B, H, W, cout, cout, 3):      # 1x1 shortcut fused as extra K chunks over the raw input
                out = ops.conv16(h, w[n + ".out_layers.3.h16"], cout, 3, gn=gn2, gn_silu=True,
                                 bias=w[n + ".out_plus_skip.bias"], skip=(x.t, None),
                                 skip_weight=w[n + ".skip_connection.h16.flat"])
            else:
                xs = ops.conv16(x.t, w[n + ".skip_connection.h16"], cout, 1, bias=w[n + ".skip_connection.bias"],
                                emit_stats=False).t
                out = self._conv3_16(n + ".out_layers.3.h16", cout, h, gn2, bias=b2, res=xs)
        if tape is not None:
            tape.append(("res", n, L, x.t, h.t, k1, k2))
        return out

    def _attn_h16(self, n, x, tape):
        w = self.w
        B, H, W, C = x.t.shape
        if self.head_ch != 64:
            raise NotImplementedError("the fused attention kernels are built for 64-channel heads (the DDNM classifier)")
        k = {} if tape is not None else None
        gn = self._gn(x, n + ".norm", keep=k)
        a = ops.gn_apply16(x, None, gn, False)
        qkv = ops.conv16(a, w[n + ".qkv.h16"], 3 * C, 1, bias=w[n + ".qkv.bias"], emit_stats=False).t
        lse = None if tape is None else torch.empty(B, C // 64, H * W, dtype=torch.float32, device=qkv.device)
        o = ops.attn16(qkv, C, lse=lse)
        out = ops.conv16(o, w[n + ".proj_out.h16"], C, 1, bias=w[n + ".proj_out.bias"], res=x)
        if tape is not None:
            tape.append(("attn", n, x.t, qkv, o, lse, k))
        return out

    def _forward_h16(self, x, film_all, tape, replicas=1):
        """The torso on fp16 NHWC activations; returns the tensor the output head (GroupNorm, SiLU, pool) reads.

        `replicas` = G > 1: the batch is G copies of the same B / G images with different timesteps (the grouped guidance
        pass of svd_ddnm.py::_GuidanceAhead).  The input convolution and the first ResBlock's `in_layers` (GroupNorm, SiLU,
        3x3 convolution) do not see the timestep -- FiLM enters at `out_layers` (unet.py:248-251) -- so they are evaluated
        ONCE on the B / G distinct images and their results (tensor, GroupNorm partials, kept affine) replicated; every
        launch computes a pixel tile from its own image only, so this is bit-identical to evaluating all G copies."""
        w = self.w
        n0 = "input_blocks.0.0"
        B = x.shape[0]
        first = self.input_blocks[1][0] if len(self.input_blocks) > 1 else None
        share = (replicas > 1 and B % replicas == 0 and first is not None and first[0] == "res" and first[3] == ""
                 and first[1] == first[2] and SHARE_PREFIX)
        shared = None
        if share:
            G, nu = replicas, B // replicas
            rep = lambda t_: torch.cat([t_] * G, 0)           # noqa: E731
            hu = ops.nchw_to_nhwc16(x[:nu].float().contiguous(), 64)
            hu = ops.conv16(hu, w[n0 + ".h16"], self.input_blocks[0][0][2], 3, bias=w[n0 + ".bias"])
            n1 = "input_blocks.1.0"
            k1u = {}
            gn1 = self._gn(hu, n1 + ".in_layers.0", keep=k1u)
            au = self._conv3_16(n1 + ".in_layers.2.h16", first[2], hu, gn1, bias=w[n1 + ".in_layers.2.bias"])
            h = ops.Act(rep(hu.t), None if hu.stats is None else rep(hu.stats), hu.tiles)
            a1 = ops.Act(rep(au.t), None if au.stats is None else rep(au.stats), au.tiles)
            k1 = {k: (rep(v) if torch.is_tensor(v) else v) for k, v in k1u.items()}
            shared = (a1, k1)
        else:
            h = ops.nchw_to_nhwc16(x.float().contiguous(), 64)
            h = ops.conv16(h, w[n0 + ".h16"], self.input_blocks[0][0][2], 3, bias=w[n0 + ".bias"])
        for prefix, layers in self._walk():
            for j, Ld in enumerate(layers):
                n = f"{prefix}.{j}"
                if Ld[0] == "res":
                    h = self._res_h16(n, Ld, h, film_all, tape, shared=shared if n == "input_blocks.1.0" else None)
                elif Ld[0] == "attn":
                    h = self._attn_h16(n, h, tape)
        return h

    def forward(self, x, timesteps, tape=None, replicas=1):
        """logits [B, 1000]; with `tape` (a list) the activations needed by the backward pass are recorded."""
        if self.w is None:
            raise RuntimeError("load_state_dict() must be called before forward()")
        w = self.w
        B = x.shape[0]
        mb = self.max_group_batch
        if B > mb:                   # more images than one launch can address: micro-batches (no tape across chunks)
            assert tape is None
            return torch.cat([self.forward(x[i:i + mb], timesteps[i:i + mb]) for i in range(0, B, mb)], 0)
        self._workspace(B)
        L = _lib.lib()
        t = timesteps.to(device=x.device, dtype=torch.float32).contiguous()
        emb = ops.timestep_embedding(t, w["time.freq"], order=1)
        emb = ops.linear(emb, w["time_embed.0.weight"], w["time_embed.0.bias"])
        emb = ops.linear(emb, w["time_embed.2.weight"], w["time_embed.2.bias"], silu_in=True)
        film_all = ops.linear(emb, w["film_cat.weight"], w["film_cat.bias"], silu_in=True)
        if self.use_fp16 and self.h16:
            h = self._forward_h16(x, film_all, tape, replicas)
            return self._head(h, x, tape)
        im2col = "input_blocks.0.0.weight.im2col" in w
        h = (ops.nchw_im2col3x3_pad if im2col else ops.nchw_to_nhwc_pad)(x.float().contiguous(), CIN_PAD)
        for prefix, layers in self._walk():
            for j, Ld in enumerate(layers):
                n = f"{prefix}.{j}"
                if Ld[0] == "conv" and im2col:
                    h = ops.conv2d(h, w[n + ".weight.im2col"], Ld[2], 1, bias=w[n + ".bias"], emit_stats=True)
                elif Ld[0] == "conv":
                    h = ops.conv2d(h, w[n + ".weight"], Ld[2], 3, bias=w[n + ".bias"], emit_stats=True)
                elif Ld[0] == "res":
                    h = self._res(n, Ld, h, film_all, tape)
                else:
                    h = self._attn(n, h, tape)
        return self._head(h, x, tape)

    def _head(self, h, x, tape):
        """out: GroupNorm -> SiLU -> AttentionPool2d (unet.py:833-841, 22-51) over the fp32 or fp16 torso output."""
        w = self.w
        L = _lib.lib()
        B = x.shape[0]
        kp = {} if tape is not None else None
        gn = self._gn(h, "out.0", keep=kp)
        C, HW = self.final_ch, self.pool_sp ** 2
        T, nh = HW + 1, self.final_ch // self.head_ch
        # the B*T = B*65 token rows are padded to a multiple of 64 (zero rows) so that the two projections of the pool run
        # on the MFMA GEMM tiles instead of the one-thread-per-output fallback (2 x 0.94 ms per step at B = 8)
        Mp = (B * T + 63) // 64 * 64
        X = torch.empty(Mp, C, dtype=torch.float32, device=x.device)
        if Mp > B * T:
            ops.fill_(X[B * T:], 0.0)
        tok = L.ddnm_pool_tokens_h16 if h.t.dtype == torch.float16 else L.ddnm_pool_tokens_f32
        check(tok(_p(h.t), _p(gn[0]), _p(gn[1]), _p(w["pool.pos"]), _p(X), B, HW, C, ops._stream()), "ddnm_pool_tokens")
        qkv = torch.empty(Mp, 3 * C, dtype=torch.float32, device=x.device)
        ops.bgemm(X, w["pool.qkv.weight"], qkv, Mp, 3 * C, C, lda=C, ldb=C, ldc=3 * C, transb=True,
                  D=w["pool.qkv.bias"], ldd=0, beta=1.0)
        P = torch.empty(B, nh, T, dtype=torch.float32, device=x.device)
        a0 = torch.empty(B, C, dtype=torch.float32, device=x.device)
        check(L.ddnm_pool_attn_fwd_f32(_p(qkv), _p(P), _p(a0), B, T, C, nh, ops._stream()), "ddnm_pool_attn_fwd_f32")
        logits = ops.linear(a0, w["pool.c.weight"], w["pool.c.bias"])
        if tape is not None:
            tape.append(("pool", h.t, kp, qkv, P))
        return logits

    __call__ = forward

    # ------------------------------------------------------------------ backward: d log p(y|x,t) / dx
    def _gn_bwd(self, x, dA, keep, silu, add=None, dA_ups=False, add_ups=False):
        B, H, W, C = x.shape
        L = _lib.lib()
        nchunk = L.ddnm_gn_bwd_nchunk(H * W, C)
        dx = torch.empty_like(x)
        check(L.ddnm_gn_bwd_f32(_p(x), _p(dA), int(dA_ups), _p(keep["scale"]), _p(keep["shift"]), _p(keep["mean_rstd"]),
                                int(silu), _p(add), int(add_ups), B, H, W, C, keep["groups"], _p(self._bwd_partial),
                                nchunk, _p(self._bwd_coef), _p(dx), ops._stream()), "ddnm_gn_bwd_f32")
        return dx

    def _res_bwd(self, rec, dout):
        _, n, L, x, h1, k1, k2 = rec
        w = self.w
        cin, cout, mode = L[1], L[2], L[3]
        da2 = ops.conv2d(dout, w[n + ".out_layers.3.dgrad"], cout, 3, weight_f16=self._w16(n + ".out_layers.3.dgrad"))
        dh1 = self._gn_bwd(h1, da2, k2, True)
        da1 = ops.conv2d(dh1, w[n + ".in_layers.2.dgrad"], cin, 3, weight_f16=self._w16(n + ".in_layers.2.dgrad"))
        if mode == "down":
            return self._gn_bwd(x, da1, k1, True, add=dout, dA_ups=True, add_ups=True)
        skip = dout if cin == cout else ops.conv2d(dout, w[n + ".skip_connection.dgrad"], cin, 1,
                                                   weight_f16=self._w16(n + ".skip_connection.dgrad"))
        return self._gn_bwd(x, da1, k1, True, add=skip)

    def _attn_bwd(self, rec, dout):
        _, n, x, qkv, P, k = rec
        w = self.w
        B, H, W, C = x.shape
        T, hc = H * W, self.head_ch
        nh = C // hc
        dO = ops.conv2d(dout, w[n + ".proj_out.dgrad"], C, 1, weight_f16=self._w16(n + ".proj_out.dgrad")).view(-1)
        flat = qkv.view(-1)
        dqkv = torch.empty_like(qkv)
        dflat = dqkv.view(-1)
        sq, sp = (T * 3 * C, 3 * hc), (nh * T * T, T * T)
        # dV = P^T dO
        ops.bgemm(P, dO, dflat[2 * hc:], T, hc, T, lda=T, ldb=C, ldc=3 * C, transb=False, transa=True, batch=B * nh,
                  inner=nh, sA=sp, sB=(T * C, hc), sC=sq)
        # dP = dO V^T ;  dS = scale * P .* (dP - rowsum(dP .* P))
        dP = torch.empty_like(P)
        ops.bgemm(dO, flat[2 * hc:], dP, T, T, hc, lda=C, ldb=3 * C, ldc=T, transb=True, batch=B * nh, inner=nh,
                  sA=(T * C, hc), sB=sq, sC=sp)
        check(_lib.lib().ddnm_softmax_bwd_rows_f32(_p(P), _p(dP), B * nh * T, T, T, 1.0 / math.sqrt(hc), ops._stream()),
              "ddnm_softmax_bwd_rows_f32")
        # dQ = dS K ;  dK = dS^T Q
        ops.bgemm(dP, flat[hc:], dflat, T, hc, T, lda=T, ldb=3 * C, ldc=3 * C, transb=False, batch=B * nh, inner=nh,
                  sA=sp, sB=sq, sC=sq)
        ops.bgemm(dP, flat, dflat[hc:], T, hc, T, lda=T, ldb=3 * C, ldc=3 * C, transb=False, transa=True, batch=B * nh,
                  inner=nh, sA=sp, sB=sq, sC=sq)
        dn = ops.conv2d(dqkv, w[n + ".qkv.dgrad"], C, 1, weight_f16=self._w16(n + ".qkv.dgrad"))
        return self._gn_bwd(x, dn, k, False, add=dout)

    # ------------------------------------------------------------------ backward of the fp16-activation path
    def _gn_bwd16(self, x, dA, keep, silu, add=None, dA_ups=False, add_ups=False):
        B, H, W, C = x.shape
        L = _lib.lib()
        nchunk = L.ddnm_gn_bwd_nchunk(H * W, C)
        dx = torch.empty_like(x)
        check(L.ddnm_gn_bwd_h16(_p(x), _p(dA), int(dA_ups), _p(keep["scale"]), _p(keep["shift"]), _p(keep["mean_rstd"]),
                                int(silu), _p(add), int(add_ups), B, H, W, C, keep["groups"], _p(self._bwd_partial),
                                nchunk, _p(self._bwd_coef), _p(dx), ops._stream()), "ddnm_gn_bwd_h16")
        return dx

    def _res_bwd16(self, rec, dout):
        _, n, L, x, h1, k1, k2 = rec
        cin, cout, mode = L[1], L[2], L[3]
        da2 = self._conv3_16(n + ".out_layers.3.dgrad.h16", cout, dout, None, emit_stats=False).t
        dh1 = self._gn_bwd16(h1, da2, k2, True)
        da1 = self._conv3_16(n + ".in_layers.2.dgrad.h16", cin, dh1, None, emit_stats=False).t
        if mode == "down":
            return self._gn_bwd16(x, da1, k1, True, add=dout, dA_ups=True, add_ups=True)
        skip = dout if cin == cout else ops.conv16(dout, self.w[n + ".skip_connection.dgrad.h16"], cin, 1,
                                                   emit_stats=False).t
        return self._gn_bwd16(x, da1, k1, True, add=skip)

    def _attn_bwd16(self, rec, dout):
        _, n, x, qkv, o, lse, k = rec
        w = self.w
        C = x.shape[3]
        dO = ops.conv16(dout, w[n + ".proj_out.dgrad.h16"], C, 1, emit_stats=False).t
        dqkv = ops.attn16_bwd(qkv, o, dO, lse)
        dn = ops.conv16(dqkv, w[n + ".qkv.dgrad.h16"], C, 1, emit_stats=False).t
        return self._gn_bwd16(x, dn, k, False, add=dout)

    def log_prob_grad(self, x, timesteps, y, replicas=1):
        """d/dx log_softmax(classifier(x, t))[y]  as NCHW [B, 3, R, R].  `replicas` = G: the caller asserts that x is G
        copies of the same B / G images (see `_forward_h16`); results do not depend on it."""
        L = _lib.lib()
        w = self.w
        B = x.shape[0]
        mb = self.max_group_batch
        if B > mb:                   # micro-batches: forward + backward per chunk, gradients concatenated
            return torch.cat([self.log_prob_grad(x[i:i + mb], timesteps[i:i + mb], y[i:i + mb]) for i in range(0, B, mb)], 0)
        tape = []
        logits = self.forward(x, timesteps, tape=tape, replicas=replicas)
        yy = y.to(device=x.device, dtype=torch.int64).contiguous().view(-1)
        dlogits = torch.empty_like(logits)
        check(L.ddnm_logsoftmax_grad_f32(_p(logits), _p(yy), _p(dlogits), B, logits.shape[1], ops._stream()),
              "ddnm_logsoftmax_grad_f32")
        gscale = GRAD_SCALE_FP16 if self.use_fp16 else 1.0
        if gscale != 1.0:           # linear backward pass: evaluate on a scaled gradient, rescale at the end
            check(L.ddnm_axpby_f32(_p(dlogits), None, _p(dlogits), dlogits.numel(), gscale, 0.0, ops._stream()),
                  "ddnm_axpby_f32")
        # AttentionPool2d backward
        _, h_pre, kp, qkv, P = tape.pop()
        C, HW = self.final_ch, self.pool_sp ** 2
        T, nh = HW + 1, self.final_ch // self.head_ch
        da0 = ops.linear(dlogits, w["pool.c.weight_t"], None)
        dqkv = torch.empty_like(qkv)                     # [Mp, 3C]: B*T token rows + zero padding rows (see forward)
        Mp = qkv.shape[0]
        if Mp > B * T:
            ops.fill_(dqkv[B * T:], 0.0)
        check(L.ddnm_pool_attn_bwd_f32(_p(qkv), _p(P), _p(da0), _p(dqkv), B, T, C, nh, ops._stream()),
              "ddnm_pool_attn_bwd_f32")
        dX = torch.empty(Mp, C, dtype=torch.float32, device=x.device)
        ops.bgemm(dqkv, w["pool.qkv.weight"], dX, Mp, C, 3 * C, lda=3 * C, ldb=C, ldc=C, transb=False)
        dact = torch.empty_like(h_pre)
        n = "input_blocks.0.0"
        if h_pre.dtype == torch.float16:            # fp16-activation path: gradients travel as fp16 NHWC tensors too
            check(L.ddnm_pool_tokens_bwd_h16(_p(dX), _p(dact), B, HW, C, ops._stream()), "ddnm_pool_tokens_bwd_h16")
            dh = self._gn_bwd16(h_pre, dact, kp, True)
            while tape:
                rec = tape.pop()
                dh = self._res_bwd16(rec, dh) if rec[0] == "res" else self._attn_bwd16(rec, dh)
            # data gradient of the input convolution: 3 output channels, fp32 NCHW (the small-Cout form of conv16)
            grad = ops.conv16_out(dh, w[n + ".dgrad.h16"], self.in_channels)
        else:
            check(L.ddnm_pool_tokens_bwd_f32(_p(dX), _p(dact), B, HW, C, ops._stream()), "ddnm_pool_tokens_bwd_f32")
            dh = self._gn_bwd(h_pre, dact, kp, True)
            while tape:
                rec = tape.pop()
                dh = self._res_bwd(rec, dh) if rec[0] == "res" else self._attn_bwd(rec, dh)
            grad = ops.conv2d(dh, w[n + ".dgrad"], self.in_channels, 3, out_nchw=True)
        if gscale != 1.0:
            check(L.ddnm_axpby_f32(_p(grad), None, _p(grad), grad.numel(), 1.0 / gscale, 0.0, ops._stream()),
                  "ddnm_axpby_f32")
        return grad


def make_cond_fn(classifier, classifier_scale):
    """The `cond_fn` closure of guided_diffusion/diffusion.py:183-189 on the HIP engine."""

    def cond_fn(x, t, y, replicas=1):
        g = classifier.log_prob_grad(x, t, y, replicas=replicas)
        if classifier_scale != 1.0:
            from ..functions.svd_operators import _axpby
            g = _axpby(g, None, float(classifier_scale), 0.0)
        return g
    # marks the engine's own guidance function: ddnm_diffusion may evaluate several reverse steps per pass over a
    # replicated batch (svd_ddnm.py::_GuidanceAhead); `max_group_batch` = images per pass the largest activation allows
    cond_fn.ddnm_engine = classifier
    return cond_fn
