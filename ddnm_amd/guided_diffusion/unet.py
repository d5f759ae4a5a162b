"""MI355X engine for the ImageNet (ADM / guided-diffusion) noise predictor.

Drop-in for the reference's `guided_diffusion/unet.py::UNetModel` (:396-664) as built by
`script_util.create_model` (:130-185): `create_model(**vars(config.model))` here takes the same
keyword arguments, the engine loads the public checkpoints' state-dict unchanged (566 keys;
`label_emb.weight` when class-conditional) and is called as `model(x, t[, y]) -> [B, 6, R, R]`.

Kernel mapping (ddnm_amd/csrc), activations NHWC fp32 in HBM:
  * every 3x3 / 1x1 conv and Conv1d(k=1)  -> the MFMA implicit-GEMM kernels, with GroupNorm(+FiLM)
    + SiLU, nearest x2, skip concat fused into the loader and bias / residual (through a nearest
    x2 upsample for `up=True` blocks) into the epilogue;
  * FiLM  GN(h)*(1+scale)+shift (:248-251)  -> folded into the per-(sample, channel) affine by
    `gn_finalize` -- no extra pass over the tensor;
  * `down=True` halves AvgPool2d(SiLU(GN(x))) / AvgPool2d(x) (:237-242) -> one HBM-bound kernel each;
  * QKVAttentionLegacy (:339-354): heads are strided views of the fused qkv tensor
    (channel = head*192 + {q,k,v}*64 + c), QK^T and PV as batched MFMA GEMMs, fp32 softmax;
  * all `emb_layers` Linears -> ONE launch per step.

Precision: fp32 on `v_mfma_f32_32x32x2_f32` by default.  After `convert_to_fp16()` (what the runner calls for
`use_fp16: true`, diffusion.py:145-146) the torso runs the fp16-ACTIVATION path (`_forward16`): every activation
in HBM is fp16 NHWC like the reference's `h.type(self.dtype)` tensors (unet.py:655-663), convolutions are
`ddnm_conv16` (fp16 MFMA operands, fp32 accumulation, csrc/conv16.hip), attention is the fused `ddnm_attn16_d64`,
GroupNorm statistics / softmax / embeddings stay fp32 (fp16_util.py:15-22, nn.py:17-19) and the output convolution
returns fp32 NCHW.  (`DDNM_ADM_GEN1=1` selects the first-generation path: fp32 activations, fp16 MFMA operands.)
"""
import os
import math
from collections import OrderedDict

import torch

from .. import ops

GN_EPS = 1e-5          # torch.nn.GroupNorm default, guided_diffusion/nn.py:93-100
CIN_PAD = 32
CIN_PAD_F16 = 64        # one K chunk of the fp16-operand kernel
NUM_CLASSES = 1000


def create_model(image_size, num_channels, num_res_blocks, channel_mult="", learn_sigma=False, class_cond=False,
                 use_checkpoint=False, attention_resolutions="16", num_heads=1, num_head_channels=-1,
                 num_heads_upsample=-1, use_scale_shift_norm=False, dropout=0, resblock_updown=False, use_fp16=False,
                 use_new_attention_order=False, **kwargs):
    """script_util.py:130-185 (extra YAML keys are swallowed by **kwargs like in the reference)."""
    if channel_mult == "":
        table = {512: (0.5, 1, 1, 2, 2, 4, 4), 256: (1, 1, 2, 2, 4, 4), 128: (1, 1, 2, 3, 4), 64: (1, 2, 3, 4)}
        if image_size not in table:
            raise ValueError(f"unsupported image size: {image_size}")
        channel_mult = table[image_size]
    else:
        channel_mult = tuple(int(v) for v in channel_mult.split(","))
    attention_ds = tuple(image_size // int(res) for res in attention_resolutions.split(","))
    return UNetModel(image_size=image_size, in_channels=3, model_channels=num_channels,
                     out_channels=(3 if not learn_sigma else 6), num_res_blocks=num_res_blocks,
                     attention_resolutions=attention_ds, channel_mult=channel_mult,
                     num_classes=(NUM_CLASSES if class_cond else None), num_heads=num_heads,
                     num_head_channels=num_head_channels, use_scale_shift_norm=use_scale_shift_norm,
                     resblock_updown=resblock_updown, use_new_attention_order=use_new_attention_order)


class UNetModel:
    def __init__(self, image_size, in_channels, model_channels, out_channels, num_res_blocks, attention_resolutions,
                 channel_mult=(1, 2, 4, 8), num_classes=None, num_heads=1, num_head_channels=-1,
                 use_scale_shift_norm=False, resblock_updown=False, use_new_attention_order=False, device=None,
                 **kwargs):
        if use_new_attention_order:
            raise NotImplementedError("use_new_attention_order=True is not used by the DDNM configs")
        if not resblock_updown:
            raise NotImplementedError("conv_resample Up/Downsample layers are not used by the DDNM configs")
        if not use_scale_shift_norm:
            raise NotImplementedError("use_scale_shift_norm=False is not used by the DDNM configs")
        self.image_size, self.in_channels, self.model_channels = image_size, in_channels, model_channels
        self.out_channels, self.num_classes = out_channels, num_classes
        self.num_heads, self.num_head_channels = num_heads, num_head_channels
        self.time_embed_dim = 4 * model_channels
        self.device = torch.device("cuda") if device is None else torch.device(device)
        mc = model_channels
        ch = int(channel_mult[0] * mc)
        self.input_blocks = [[("conv", in_channels, ch)]]
        chans = [ch]
        ds = 1
        for level, mult in enumerate(channel_mult):
            for _ in range(num_res_blocks):
                layers = [("res", ch, int(mult * mc), "")]
                ch = int(mult * mc)
                if ds in attention_resolutions:
                    layers.append(("attn", ch))
                self.input_blocks.append(layers)
                chans.append(ch)
            if level != len(channel_mult) - 1:
                self.input_blocks.append([("res", ch, ch, "down")])
                chans.append(ch)
                ds *= 2
        self.middle_block = [("res", ch, ch, ""), ("attn", ch), ("res", ch, ch, "")]
        self.output_blocks = []
        for level, mult in list(enumerate(channel_mult))[::-1]:
            for i in range(num_res_blocks + 1):
                ich = chans.pop()
                layers = [("res", ch + ich, int(mc * mult), "", ch)]       # last field: channels of `h` in the concat
                ch = int(mc * mult)
                if ds in attention_resolutions:
                    layers.append(("attn", ch))
                if level and i == num_res_blocks:
                    layers.append(("res", ch, ch, "up"))
                    ds //= 2
                self.output_blocks.append(layers)
        self.final_ch = ch
        # FiLM projection layout: one slice [2*cout] per ResBlock, in execution order
        self._res_names = []
        off = 0
        self._film_off = {}
        for prefix, layers in self._walk():
            for j, L in enumerate(layers):
                if L[0] == "res":
                    n = f"{prefix}.{j}"
                    self._film_off[n] = off
                    off += 2 * L[2]
                    self._res_names.append((n, L))
        self.film_total = off
        self.max_ch = max(L[1] for _, L in self._res_names)
        self.w = None
        self._ws = None
        self._ws_by_stream = {}
        self._graphs = None
        self.use_fp16 = False

    def _walk(self):
        for i, layers in enumerate(self.input_blocks):
            yield f"input_blocks.{i}", layers
        yield "middle_block", self.middle_block
        for i, layers in enumerate(self.output_blocks):
            yield f"output_blocks.{i}", layers

    # ------------------------------------------------------------------ nn.Module-like surface
    def to(self, device):
        return self

    def eval(self):
        return self

    def convert_to_fp16(self):
        """The reference's `model.convert_to_fp16()` (diffusion.py:145-146, unet.py:619-625): the 3x3 convs
        of the torso run on fp16 MFMA operands with fp32 accumulation (csrc/conv_igemm_f16.hip); GroupNorm,
        softmax, embeddings, `out.*` stay fp32 exactly like fp16_util.py:15-22 leaves them."""
        self.use_fp16 = True
        if self.w is not None:
            self._pack_h16()
        return self

    def convert_to_fp32(self):
        self.use_fp16 = False
        return self

    def _dev(self, raw):
        return raw.to(device=self.device, dtype=torch.float32).contiguous()

    def _pack_f16(self):
        """fp16-operand packs of the first-generation path (DDNM_ADM_GEN1=1): built on first use only -- the raw
        weights stay on the HOST (`_raw_all`), so the default fp16-activation path keeps one packed copy per weight in
        HBM (raw fp32 + fp32 packs + both fp16 packs were ~4x the weight footprint)."""
        if "f16.ready" in self.w:
            return
        for key in self._f16_keys:
            self.w[key + ".f16"] = ops.pack_conv_weight_f16(self._dev(self._raw_all[key]))
            if key.endswith(".skip_connection.weight"):
                self.w[key[:-len(".weight")] + ".fused.f16"] = ops.pack_skip_weight(self._dev(self._raw_all[key]), f16=True)
        if self._pad64_key is not None:
            self.w[self._pad64_key + ".pad64.f16"] = ops.pack_conv_weight_f16(self._dev(self._raw_all[self._pad64_key]),
                                                                             cin_pad=CIN_PAD_F16)
        self.w["f16.ready"] = True

    def _pack_h16(self):
        """fp16 (O,ky,kx,I) weights of the fp16-activation path, Cout padded to 256 rows (ops.pack_conv_weight16)."""
        w = self.w
        if "h16.ready" in w:
            return
        for key, raw in self._raw_all.items():
            raw = self._dev(raw)
            if key.endswith("input_blocks.0.0.weight"):
                w[key + ".h16"] = ops.pack_conv_weight16(raw, cin_pad=CIN_PAD_F16)
            elif key.endswith(".skip_connection.weight"):
                p16 = ops.pack_conv_weight16(raw)
                w[key + ".h16"] = p16                                              # as a 1x1 convolution
                w[key + ".h16.flat"] = p16.reshape(p16.shape[0], -1).contiguous()   # as the fused shortcut's [Cout][Cin]
            else:
                w[key + ".h16"] = ops.pack_conv_weight16(raw)
        w["h16.ready"] = True

    def parameters(self):
        return iter(())

    def state_dict_shapes(self):
        s = OrderedDict()
        ted, mc = self.time_embed_dim, self.model_channels
        s["time_embed.0.weight"], s["time_embed.0.bias"] = (ted, mc), (ted,)
        s["time_embed.2.weight"], s["time_embed.2.bias"] = (ted, ted), (ted,)
        if self.num_classes is not None:
            s["label_emb.weight"] = (self.num_classes, ted)
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
        s["out.0.weight"], s["out.0.bias"] = (self.final_ch,), (self.final_ch,)
        s["out.2.weight"], s["out.2.bias"] = (self.out_channels, self.final_ch, 3, 3), (self.out_channels,)
        return s

    def random_state_dict(self, seed=1234):
        g = torch.Generator().manual_seed(seed)
        sd = OrderedDict()
        for name, shape in self.state_dict_shapes().items():
            if name.endswith(".weight") and len(shape) >= 2:
                fan_in = 1
                for d in shape[1:]:
                    fan_in *= d
                sd[name] = torch.randn(shape, generator=g) * fan_in ** -0.5
            elif name.endswith(".weight"):
                sd[name] = 1.0 + 0.1 * torch.randn(shape, generator=g)
            else:
                sd[name] = 0.05 * torch.randn(shape, generator=g)
        return sd

    def load_state_dict(self, sd, strict=True):
        dev = self.device
        g = lambda k: sd[k].detach().to(device=dev, dtype=torch.float32).contiguous()   # noqa: E731 (fp16 ckpts upcast)
        w = {}
        self._f16_keys, self._pad64_key = [], None
        self._raw_all = {}             # HOST references to the checkpoint's conv weights (packed on demand)
        host = lambda k: sd[k].detach()   # noqa: E731
        for k in ("time_embed.0", "time_embed.2"):
            w[k + ".weight"], w[k + ".bias"] = g(k + ".weight"), g(k + ".bias")
        if self.num_classes is not None:
            w["label_emb.weight"] = g("label_emb.weight")
        fw, fb = [], []
        for prefix, layers in self._walk():
            for j, L in enumerate(layers):
                n = f"{prefix}.{j}"
                if L[0] == "conv":
                    w[n + ".weight"] = ops.pack_conv_weight(g(n + ".weight"), cin_pad=CIN_PAD)
                    w[n + ".bias"] = g(n + ".bias")
                    self._raw_all[n + ".weight"] = host(n + ".weight")
                    if L[2] % 128 == 0:          # fp16 mode: the 3 input channels are zero-padded to one 64-channel chunk
                        self._pad64_key = n + ".weight"
                elif L[0] == "res":
                    for norm in ("in_layers.0", "out_layers.0"):
                        w[f"{n}.{norm}.weight"], w[f"{n}.{norm}.bias"] = g(f"{n}.{norm}.weight"), g(f"{n}.{norm}.bias")
                    for conv in ("in_layers.2", "out_layers.3"):
                        raw = g(f"{n}.{conv}.weight")
                        w[f"{n}.{conv}.weight"] = ops.pack_conv_weight(raw)
                        w[f"{n}.{conv}.bias"] = g(f"{n}.{conv}.bias")
                        self._raw_all[f"{n}.{conv}.weight"] = host(f"{n}.{conv}.weight")
                        if raw.shape[1] % 64 == 0 and raw.shape[0] % 128 == 0:
                            self._f16_keys.append(f"{n}.{conv}.weight")
                    fw.append(g(n + ".emb_layers.1.weight"))
                    fb.append(g(n + ".emb_layers.1.bias"))
                    if L[1] != L[2]:
                        raw = g(n + ".skip_connection.weight")
                        self._raw_all[n + ".skip_connection.weight"] = host(n + ".skip_connection.weight")
                        w[n + ".skip_connection.weight"] = ops.pack_conv_weight(raw)
                        if raw.shape[1] % 64 == 0 and raw.shape[0] % 128 == 0:
                            self._f16_keys.append(n + ".skip_connection.weight")
                        w[n + ".skip_connection.bias"] = g(n + ".skip_connection.bias")
                        w[n + ".skip_connection.fused"] = ops.pack_skip_weight(raw)
                        w[n + ".out_plus_skip.bias"] = (g(n + ".out_layers.3.bias") + g(n + ".skip_connection.bias")).contiguous()
                else:
                    w[n + ".norm.weight"], w[n + ".norm.bias"] = g(n + ".norm.weight"), g(n + ".norm.bias")
                    for conv in ("qkv", "proj_out"):                     # Conv1d(k=1) == 1x1 convolution
                        raw = g(f"{n}.{conv}.weight").unsqueeze(-1)
                        self._raw_all[f"{n}.{conv}.weight"] = host(f"{n}.{conv}.weight").unsqueeze(-1)
                        w[f"{n}.{conv}.weight"] = ops.pack_conv_weight(raw)
                        w[f"{n}.{conv}.bias"] = g(f"{n}.{conv}.bias")
                        if raw.shape[1] % 64 == 0 and raw.shape[0] % 128 == 0:
                            self._f16_keys.append(f"{n}.{conv}.weight")
        w["film_cat.weight"] = torch.cat(fw, 0).contiguous()
        w["film_cat.bias"] = torch.cat(fb, 0).contiguous()
        w["out.0.weight"], w["out.0.bias"] = g("out.0.weight"), g("out.0.bias")
        w["out.2.weight"] = ops.pack_conv_weight(g("out.2.weight"))
        w["out.2.bias"] = g("out.2.bias")
        self._raw_all["out.2.weight"] = host("out.2.weight")
        half = self.model_channels // 2
        w["time.freq"] = torch.exp(-math.log(10000) * torch.arange(0, half, dtype=torch.float32) / half).to(dev)
        self.w = w
        self._ws = None
        self._ws_by_stream = {}
        if self._graphs is not None:
            self._graphs.reset()
        self._auto_graphs = None
        if self.use_fp16:
            self._pack_h16()
        return self

    def _w16(self, key):
        return self.w.get(key + ".f16") if self.use_fp16 else None

    # ------------------------------------------------------------------ forward
    def _workspace(self, B):
        """GroupNorm scratch of the current stream (the affine of one GroupNorm is consumed by the next launch on the
        same stream; the two half-batch streams of a captured forward each own one)."""
        key = torch.cuda.current_stream().cuda_stream
        ent = self._ws_by_stream.get(key)
        if ent is None or ent[1] < B:
            res, mp = self.image_size, 0
            while res >= 8:
                mp = max(mp, ops.gn_nchunk(res * res, min(self.max_ch, 4096)))
                res //= 2
            ent = (ops.GroupNormWorkspace(self.device, B, self.max_ch, B * mp * 32 * 2), B)
            self._ws_by_stream[key] = ent
        self._ws = ent[0]
        return self._ws

    def _gn(self, x0, x1, name, film=None):
        if x1 is None and isinstance(x0, ops.Act) and x0.gn is not None and x0.gn[2] == name \
                and x0.gn[3].generation == x0.gn[4]:
            # already finalized by the launch that produced x0 (its split-K reduction pass), and no other GroupNorm has
            # written the shared scale / shift buffers since (generation counter): no launch here
            sc, sh = x0.gn[0], x0.gn[1]
            x0.gn = None
            return sc, sh
        f, fs = (None, 0) if film is None else (film, self.film_total)
        return ops.group_norm_affine(x0, x1, self.w[name + ".weight"], self.w[name + ".bias"], GN_EPS, self._ws,
                                     film=f, film_stride=fs)

    def _res(self, n, L, x0, x1, film_all):
        w = self.w
        cin, cout, mode = L[1], L[2], L[3]
        film = film_all[:, self._film_off[n]:]
        gn1 = self._gn(x0, x1, n + ".in_layers.0")
        if mode == "down":
            hp = ops.avgpool2_nhwc(x0.t, gn=gn1, silu=True)
            xs = ops.avgpool2_nhwc(x0.t)
            h = ops.conv2d(hp, w[n + ".in_layers.2.weight"], cout, 3, bias=w[n + ".in_layers.2.bias"], emit_stats=True,
                           weight_f16=self._w16(n + ".in_layers.2.weight"))
            res_ups = False
        elif mode == "up":
            h = ops.conv2d(x0, w[n + ".in_layers.2.weight"], cout, 3, gn=gn1, gn_silu=True, ups=True,
                           bias=w[n + ".in_layers.2.bias"], emit_stats=True,
                           weight_f16=self._w16(n + ".in_layers.2.weight"))
            xs, res_ups = x0, True
        else:
            h = ops.conv2d(x0, w[n + ".in_layers.2.weight"], cout, 3, src1=x1, gn=gn1, gn_silu=True,
                           bias=w[n + ".in_layers.2.bias"], emit_stats=True,
                           weight_f16=self._w16(n + ".in_layers.2.weight"))
            res_ups = False
            if cin != cout:
                B, H, W, _ = h.t.shape
                lowres_f16 = self.use_fp16 and H * W < 256        # runs as im2col + fp16 GEMM: shortcut stays a GEMM
                if ops.conv_fuses_skip(B, H, W, cout, cout) and not lowres_f16:   # shortcut fused into out_layers.3 as extra K
                    gn2 = self._gn(h, None, n + ".out_layers.0", film=film)
                    return ops.conv2d(h, w[n + ".out_layers.3.weight"], cout, 3, gn=gn2, gn_silu=True,
                                      bias=w[n + ".out_plus_skip.bias"], skip=(x0, x1),
                                      skip_weight=w[n + ".skip_connection.fused"],
                                      skip_weight_f16=w.get(n + ".skip_connection.fused.f16"), emit_stats=True,
                                      weight_f16=self._w16(n + ".out_layers.3.weight"))
                xs = ops.conv2d(x0, w[n + ".skip_connection.weight"], cout, 1, src1=x1,
                                bias=w[n + ".skip_connection.bias"], weight_f16=self._w16(n + ".skip_connection.weight"))
            else:
                assert x1 is None
                xs = x0
        gn2 = self._gn(h, None, n + ".out_layers.0", film=film)
        return ops.conv2d(h, w[n + ".out_layers.3.weight"], cout, 3, gn=gn2, gn_silu=True,
                          bias=w[n + ".out_layers.3.bias"], res=xs, res_ups=res_ups, emit_stats=True,
                          weight_f16=self._w16(n + ".out_layers.3.weight"))

    def _attn(self, n, x):
        w = self.w
        B, H, W, C = x.t.shape
        T = H * W
        hc = self.num_head_channels if self.num_head_channels != -1 else C // self.num_heads
        nh = C // hc
        gn = self._gn(x, None, n + ".norm")
        qkv = ops.conv2d(x, w[n + ".qkv.weight"], 3 * C, 1, gn=gn, gn_silu=False, bias=w[n + ".qkv.bias"],
                         weight_f16=self._w16(n + ".qkv.weight"))
        flat = qkv.view(-1)
        S = torch.empty(B * nh, T, T, dtype=torch.float32, device=qkv.device)
        ops.bgemm(flat, flat[hc:], S, T, T, hc, lda=3 * C, ldb=3 * C, ldc=T, transb=True, batch=B * nh, inner=nh,
                  sA=(T * 3 * C, 3 * hc), sB=(T * 3 * C, 3 * hc), sC=(nh * T * T, T * T))
        ops.softmax_rows_(S, B * nh * T, T, T, 1.0 / math.sqrt(hc))      # (q*s).(k*s), s = hc^-1/4
        o = torch.empty(B, H, W, C, dtype=torch.float32, device=qkv.device)
        ops.bgemm(S, flat[2 * hc:], o, T, hc, T, lda=T, ldb=3 * C, ldc=C, transb=False, batch=B * nh, inner=nh,
                  sA=(nh * T * T, T * T), sB=(T * 3 * C, 3 * hc), sC=(T * C, hc))
        return ops.conv2d(o, w[n + ".proj_out.weight"], C, 1, bias=w[n + ".proj_out.bias"], res=x, emit_stats=True,
                          weight_f16=self._w16(n + ".proj_out.weight"))

    def _run(self, prefix, layers, h, skip, film_all):
        for j, L in enumerate(layers):
            n = f"{prefix}.{j}"
            if L[0] == "conv":
                h = ops.conv2d(h, self.w[n + ".weight"], L[2], 3, bias=self.w[n + ".bias"], emit_stats=True)
            elif L[0] == "res":
                h = self._res(n, L, h, skip if j == 0 else None, film_all)
            else:
                h = self._attn(n, h)
        return h

    # ------------------------------------------------------------------ fp16-activation forward (csrc/conv16.hip)
    def _conv3x3_16(self, key, cout, x0, x1, gn, silu=True, **kw):
        """3x3 convolution of act(concat(x0, x1)) on ddnm_conv16 with the GroupNorm affine + swish and the concat fused
        into its loader; images too small for a
        pixel tile (8x8) go through im2col + one GEMM with K = 9*Cin."""
        w16 = self.w[key + ".h16"]
        B, H, W, _ = x0.t.shape
        cin = x0.t.shape[3] + (0 if x1 is None else x1.t.shape[3])
        ups = kw.get("ups", False)
        Ho, Wo = (2 * H, 2 * W) if ups else (H, W)
        if ops.conv16_supported(B, Ho, Wo, cin, cout, 3, ups=ups):
            return ops.conv16(x0, w16, cout, 3, src1=x1, gn=gn, gn_silu=silu, **kw)
        assert not ups and kw.get("skip") is None
        col = ops.im2col16(x0, x1, gn, silu)
        return ops.conv16(col, w16.reshape(w16.shape[0], 1, -1), cout, 1, **kw)

    def _fin(self, name, film=None):
        """What a producing convolution needs to finalize the GroupNorm `name` over its own output in its split-K
        reduction pass (ops.conv16 `fin=`; no effect on launches that are not split)."""
        f, fs = (None, 0) if film is None else (film, self.film_total)
        return (name, self.w[name + ".weight"], self.w[name + ".bias"], f, fs, GN_EPS, self._ws)

    def _next_fin(self, L, n):
        """`fin` of the GroupNorm a following layer opens with (single-source consumers only)."""
        return self._fin(n + (".in_layers.0" if L[0] == "res" else ".norm"))

    def _res16(self, n, L, x0, x1, film_all, next_fin=None):
        w = self.w
        cin, cout, mode = L[1], L[2], L[3]
        film = film_all[:, self._film_off[n]:]
        gn1 = self._gn(x0, x1, n + ".in_layers.0")
        b1, b2 = w[n + ".in_layers.2.bias"], w[n + ".out_layers.3.bias"]
        k1, k2 = n + ".in_layers.2.weight", n + ".out_layers.3.weight"
        fin2 = self._fin(n + ".out_layers.0", film)
        if mode == "down":          # AvgPool2d on both branches (unet.py:237-242)
            hp = ops.Act(ops.gn_apply16(x0, None, gn1, True, pool=True))
            xs = ops.gn_apply16(x0, None, None, False, pool=True)
            h = self._conv3x3_16(k1, cout, hp, None, None, bias=b1, fin=fin2)
            gn2 = self._gn(h, None, n + ".out_layers.0", film=film)
            return self._conv3x3_16(k2, cout, h, None, gn2, bias=b2, res=xs, fin=next_fin)
        if mode == "up":            # nearest x2 on both branches: operand through `ups`, residual through `res_ups`
            h = self._conv3x3_16(k1, cout, x0, None, gn1, bias=b1, ups=True, fin=fin2)
            gn2 = self._gn(h, None, n + ".out_layers.0", film=film)
            return self._conv3x3_16(k2, cout, h, None, gn2, bias=b2, res=x0, res_ups=True, fin=next_fin)
        h = self._conv3x3_16(k1, cout, x0, x1, gn1, bias=b1, fin=fin2)
        gn2 = self._gn(h, None, n + ".out_layers.0", film=film)
        if cin == cout:
            assert x1 is None
            return self._conv3x3_16(k2, cout, h, None, gn2, bias=b2, res=x0, fin=next_fin)
        B, H, W, _ = h.t.shape
        if ops.conv16_supported(B, H, W, cout, cout, 3):      # 1x1 shortcut fused as extra K chunks over the raw input
            return self._conv3x3_16(k2, cout, h, None, gn2, bias=w[n + ".out_plus_skip.bias"], skip=(x0, x1),
                                    skip_weight=w[n + ".skip_connection.weight.h16.flat"], fin=next_fin)
        raw = x0.t if x1 is None else ops.gn_apply16(x0, x1, None, False)      # materialised concat of the raw tensors
        xs = ops.conv16(raw, w[n + ".skip_connection.weight.h16"], cout, 1, bias=w[n + ".skip_connection.bias"],
                        emit_stats=False)
        return self._conv3x3_16(k2, cout, h, None, gn2, bias=b2, res=xs, fin=next_fin)

    def _attn16(self, n, x):
        w = self.w
        C = x.t.shape[3]
        hc = self.num_head_channels if self.num_head_channels != -1 else C // self.num_heads
        if hc != 64:
            raise NotImplementedError("the fused attention kernel is built for 64-channel heads (all DDNM configs)")
        gn = self._gn(x, None, n + ".norm")
        a = ops.gn_apply16(x, None, gn, False)
        qkv = ops.conv16(a, w[n + ".qkv.weight.h16"], 3 * C, 1, bias=w[n + ".qkv.bias"], emit_stats=False)
        o = ops.attn16(qkv.t, C)
        return ops.conv16(o, w[n + ".proj_out.weight.h16"], C, 1, bias=w[n + ".proj_out.bias"], res=x)

    def _run16(self, prefix, layers, h, skip, film_all, next_fin=None):
        for j, L in enumerate(layers):
            n = f"{prefix}.{j}"
            if L[0] == "res":
                nf = self._next_fin(layers[j + 1], f"{prefix}.{j + 1}") if j + 1 < len(layers) else next_fin
                h = self._res16(n, L, h, skip if j == 0 else None, film_all, nf)
            else:
                h = self._attn16(n, h)
        return h

    def _forward16(self, x, film_all):
        w = self.w
        n0 = "input_blocks.0.0"
        h = ops.nchw_to_nhwc16(x.float().contiguous(), CIN_PAD_F16)
        h = ops.conv16(h, w[n0 + ".weight.h16"], self.input_blocks[0][0][2], 3, bias=w[n0 + ".bias"])
        hs = [h]
        nblk = len(self.input_blocks)
        for i, layers in enumerate(self.input_blocks):
            if i == 0:
                continue
            # the GroupNorm that opens the NEXT block reads this block's output alone (no concat on the way down)
            nxt = (self.input_blocks[i + 1], f"input_blocks.{i + 1}.0") if i + 1 < nblk else (self.middle_block, "middle_block.0")
            h = self._run16(f"input_blocks.{i}", layers, h, None, film_all, self._next_fin(nxt[0][0], nxt[1]))
            hs.append(h)
        h = self._run16("middle_block", self.middle_block, h, None, film_all)
        for i, layers in enumerate(self.output_blocks):
            h = self._run16(f"output_blocks.{i}", layers, h, hs.pop(), film_all)
        gn = self._gn(h, None, "out.0")
        B, H, W, _ = h.t.shape
        if W % 32 == 0 and H % 8 == 0:
            return ops.conv16_out(h, w["out.2.weight.h16"], self.out_channels, bias=w["out.2.bias"], gn=gn, gn_silu=True)
        # images narrower than one 32-pixel output tile (reduced test nets): the exact-fp32 kernel on the fp16 operand
        a = ops.gn_apply16(h, None, gn, True)
        return ops.conv2d(a.float(), w["out.2.weight"], self.out_channels, 3, bias=w["out.2.bias"], out_nchw=True)

    def enable_graphs(self, two_streams=True):
        """Replay the forward from a captured hipGraph (one per batch shape): no per-launch host work, and with
        `two_streams` the two halves of the batch run as concurrent branches of the graph (ddnm_amd/graph.py)."""
        from ..graph import GraphedForward
        self._graphs = GraphedForward(self._forward_eager, two_streams=two_streams)
        return self

    def disable_graphs(self):
        self._graphs = None
        return self

    def auto_graphs(self, max_batch=2):
        """Replay forwards of at most `max_batch` images from a captured hipGraph, decided per call (0: never): the
        reference's shipped configs sample with batch_size 1 (configs/imagenet_256.yml:42), where the ~300 launches of
        a forward cost more host time than GPU time.  The runner (`Diffusion`) switches this on."""
        self.auto_graph_max_batch = int(max_batch)
        self._auto_graphs = None
        return self

    @property
    def max_forward_batch(self):
        """Chunk size of forward(): the largest activation is `model_channels` channels at full resolution (the skip
        concat travels as two tensors), fp32 or -- in the fp16-activation mode -- fp16."""
        elem = 2 if (self.use_fp16 and os.environ.get("DDNM_ADM_GEN1") != "1") else 4
        return ops.max_launch_batch(self.image_size * self.image_size * self.model_channels * elem)

    def forward(self, x, timesteps, y=None):
        if self.w is None:
            raise RuntimeError("load_state_dict() must be called before forward()")
        assert (y is not None) == (self.num_classes is not None), \
            "must specify y if and only if the model is class-conditional"
        mb = self.max_forward_batch
        if x.shape[0] > mb:          # more images than one launch can address: micro-batches, concatenated
            return torch.cat([self.forward(x[i:i + mb], timesteps[i:i + mb], None if y is None else y[i:i + mb])
                              for i in range(0, x.shape[0], mb)], 0)
        if self._graphs is not None:
            return self._graphs(x, timesteps, y)
        if x.shape[0] <= getattr(self, "auto_graph_max_batch", 0):
            if getattr(self, "_auto_graphs", None) is None:
                from ..graph import GraphedForward
                self._auto_graphs = GraphedForward(self._forward_eager, two_streams=False)
            return self._auto_graphs(x, timesteps, y)
        return self._forward_eager(x, timesteps, y)

    def _forward_eager(self, x, timesteps, y=None):
        w = self.w
        B = x.shape[0]
        self._workspace(B)
        t = timesteps.to(device=x.device, dtype=torch.float32).contiguous()
        emb = ops.timestep_embedding(t, w["time.freq"], order=1)
        emb = ops.linear(emb, w["time_embed.0.weight"], w["time_embed.0.bias"])
        emb = ops.linear(emb, w["time_embed.2.weight"], w["time_embed.2.bias"], silu_in=True)
        if self.num_classes is not None:
            assert y.shape == (B,)
            ops.embedding_add_(emb, w["label_emb.weight"], y)
        film_all = ops.linear(emb, w["film_cat.weight"], w["film_cat.bias"], silu_in=True)
        if self.use_fp16 and os.environ.get("DDNM_ADM_GEN1") != "1":
            return self._forward16(x, film_all)
        if self.use_fp16:
            self._pack_f16()            # first-generation path: its fp16-operand packs are built on first use

        n0 = "input_blocks.0.0"
        H, W = x.shape[2], x.shape[3]
        if self.use_fp16 and n0 + ".weight.pad64.f16" in w and ops.conv_runs_f16(B, H, W, CIN_PAD_F16, self.input_blocks[0][0][2]):
            # the input convolution belongs to the fp16 torso too (unet.py:619-625)
            h = ops.nchw_to_nhwc_pad(x.float().contiguous(), CIN_PAD_F16)
            h = ops.conv2d(h, w[n0 + ".weight"], self.input_blocks[0][0][2], 3, bias=w[n0 + ".bias"], emit_stats=True,
                           weight_f16=w[n0 + ".weight.pad64.f16"])
            first = 1
        else:
            h = ops.nchw_to_nhwc_pad(x.float().contiguous(), CIN_PAD)
            first = 0
        hs = []
        for i, layers in enumerate(self.input_blocks):
            if i == 0 and first:
                hs.append(h)
                continue
            h = self._run(f"input_blocks.{i}", layers, h, None, film_all)
            hs.append(h)
        h = self._run("middle_block", self.middle_block, h, None, film_all)
        for i, layers in enumerate(self.output_blocks):
            h = self._run(f"output_blocks.{i}", layers, h, hs.pop(), film_all)
        gn = self._gn(h, None, "out.0")
        return ops.conv2d(h, w["out.2.weight"], self.out_channels, 3, gn=gn, gn_silu=True, bias=w["out.2.bias"],
                          out_nchw=True)

    def __call__(self, x, timesteps, y=None):
        return self.forward(x, timesteps, y)
