"""MI355X engine for the CelebA-HQ noise predictor.

Drop-in for the reference's `guided_diffusion/models.py::Model` (:192-341): same
constructor argument (the YAML config namespace), same `state_dict` key names
(so `load_state_dict(torch.load("celeba_hq.ckpt"))` works unchanged), same call
protocol `model(x[B,3,R,R], t[B]) -> eps[B,3,R,R]` (NCHW fp32).

Inside, nothing is a torch op.  Activations live in HBM as NHWC fp32 and every
layer is a launch of a hand-written HIP kernel (ddnm_amd/csrc):

  * conv3x3 / conv1x1 / strided conv  -> implicit GEMM; its A-tile loader folds in
    GroupNorm-affine + swish, nearest x2 upsampling and the skip-connection concat
    (two source pointers); its epilogue folds in the bias, the timestep-embedding
    projection and the residual add.  The 3x3 / stride-1 layers (97 % of the FLOPs)
    run fp32-GRADE products on the fp16 matrix pipe: every fp32 operand is carried
    as hi + lo fp16 halves and a product is three v_mfma_f32_32x32x16_f16 with fp32
    accumulation (ddnm_conv3x3_s16_f32; measured CLOSER to an fp64 evaluation than
    the fp32 MFMA kernel, at 2.8x its speed); the rest (1x1, strided, 8x8 level)
    and DDNM_CONV_F32=mfma32 use v_mfma_f32_32x32x2_f32.
  * GroupNorm                          -> statistics kernel only (one read of the
    tensor); the normalised tensor is never written.
  * attention (1 head, d=512)          -> fused qkv 1x1 conv, QK^T and PV on MFMA,
    wave-shuffle softmax.
  * all 32 `temb_proj` Linears         -> ONE launch per step (they share swish(temb)).

Layer order restates models.py:301-341 (forward), :115-134 (ResnetBlock),
:165-189 (AttnBlock), :61-71 (Downsample), :47-51 (Upsample).
"""
import math
import os
from collections import OrderedDict

import torch

from .. import ops

GN_EPS = 1e-6          # models.py:33
# DDNM_CONV_F32=mfma32: every convolution on the fp32 MFMA instruction (the pre-split engine; A/B switch)
SPLIT16 = os.environ.get("DDNM_CONV_F32", "split16") != "mfma32"
CIN_PAD = 32           # conv_in reads the image through a 32-channel NHWC staging tensor


class _ResBlock:
    def __init__(self, name, cin, cout):
        self.name, self.cin, self.cout = name, cin, cout
        self.temb_off = None


class _Attn:
    def __init__(self, name, c):
        self.name, self.c = name, c


class Model:
    def __init__(self, config, device=None, split16=None):
        """`split16`: run the 3x3 / stride-1 layers on the split-fp16 kernel (None: the DDNM_CONV_F32 default above)."""
        self.config = config
        self.split16 = SPLIT16 if split16 is None else bool(split16)
        self.fused_attn = self.split16        # AttnBlock in one launch (csrc/attn_d512.hip, split-fp16 products); the strict
        #                                       fp32 engine keeps the three-launch fp32-MFMA route
        self.fused_attn_max_tokens = 64       # measured: 26.0 -> 17.4 us at T = 64, but 42.1 -> 50.7 us at T = 256 (B = 8)
        m = config.model
        self.ch, self.out_ch, self.ch_mult = m.ch, m.out_ch, tuple(m.ch_mult)
        self.num_res_blocks = m.num_res_blocks
        self.attn_resolutions = list(m.attn_resolutions)
        self.in_channels = m.in_channels
        self.resolution = config.data.image_size
        self.temb_ch = 4 * self.ch
        self.num_resolutions = len(self.ch_mult)
        if not m.resamp_with_conv:
            raise NotImplementedError("resamp_with_conv=False is not on the DDNM hot path")
        self.device = torch.device("cuda") if device is None else torch.device(device)
        self._plan()
        self.w = None
        self._ws, self._ws_by_stream, self._graphs = None, {}, None

    # ------------------------------------------------------------------ structure
    def _plan(self):
        ch, mult, nrb = self.ch, self.ch_mult, self.num_res_blocks
        in_mult = (1,) + mult
        res = self.resolution
        self.down = []
        self.res_blocks = []          # execution order (defines the temb_proj concat layout)
        block_in = None
        chans = [ch]
        for lvl in range(self.num_resolutions):
            block_in = ch * in_mult[lvl]
            block_out = ch * mult[lvl]
            blocks, attns = [], []
            for ib in range(nrb):
                rb = _ResBlock(f"down.{lvl}.block.{ib}", block_in, block_out)
                rb.res = res
                blocks.append(rb)
                self.res_blocks.append(rb)
                block_in = block_out
                if res in self.attn_resolutions:
                    attns.append(_Attn(f"down.{lvl}.attn.{ib}", block_in))
                    attns[-1].res = res
                chans.append(block_in)
            has_down = lvl != self.num_resolutions - 1
            if has_down:
                res //= 2
                chans.append(block_in)
            self.down.append((blocks, attns, has_down, block_in))
        self.mid = [_ResBlock("mid.block_1", block_in, block_in), _Attn("mid.attn_1", block_in),
                    _ResBlock("mid.block_2", block_in, block_in)]
        for m in self.mid:
            m.res = res
        self.res_blocks += [self.mid[0], self.mid[2]]
        self.up = {}
        for lvl in reversed(range(self.num_resolutions)):
            block_out = ch * mult[lvl]
            blocks, attns = [], []
            for ib in range(nrb + 1):
                skip = chans.pop()
                rb = _ResBlock(f"up.{lvl}.block.{ib}", block_in + skip, block_out)
                rb.split = (block_in, skip)
                rb.res = res
                blocks.append(rb)
                self.res_blocks.append(rb)
                block_in = block_out
                if res in self.attn_resolutions:
                    attns.append(_Attn(f"up.{lvl}.attn.{ib}", block_in))
                    attns[-1].res = res
            has_up = lvl != 0
            if has_up:
                res *= 2
            self.up[lvl] = (blocks, attns, has_up, block_in)
        self.final_ch = block_in
        off = 0
        for rb in self.res_blocks:
            rb.temb_off = off
            off += rb.cout
        self.temb_total = off
        self.max_ch = max(rb.cin for rb in self.res_blocks)

    # ------------------------------------------------------------------ nn.Module-like surface
    def to(self, device):
        return self

    def eval(self):
        return self

    def parameters(self):
        return iter(())

    def state_dict_shapes(self):
        """name -> shape of every tensor `load_state_dict` consumes; equals the reference
        `Model(config).state_dict()` (pinned in tests/test_host_logic.py against the golden key list)."""
        s = OrderedDict()

        def conv(n, co, ci, k):
            s[n + ".weight"], s[n + ".bias"] = (co, ci, k, k), (co,)

        def vec2(n, c):
            s[n + ".weight"], s[n + ".bias"] = (c,), (c,)

        def resblock(rb):
            n = rb.name
            vec2(n + ".norm1", rb.cin)
            conv(n + ".conv1", rb.cout, rb.cin, 3)
            s[n + ".temb_proj.weight"], s[n + ".temb_proj.bias"] = (rb.cout, self.temb_ch), (rb.cout,)
            vec2(n + ".norm2", rb.cout)
            conv(n + ".conv2", rb.cout, rb.cout, 3)
            if rb.cin != rb.cout:
                conv(n + ".nin_shortcut", rb.cout, rb.cin, 1)

        def attn(a):
            vec2(a.name + ".norm", a.c)
            for p in ("q", "k", "v", "proj_out"):
                conv(f"{a.name}.{p}", a.c, a.c, 1)

        s["temb.dense.0.weight"], s["temb.dense.0.bias"] = (self.temb_ch, self.ch), (self.temb_ch,)
        s["temb.dense.1.weight"], s["temb.dense.1.bias"] = (self.temb_ch, self.temb_ch), (self.temb_ch,)
        conv("conv_in", self.ch, self.in_channels, 3)
        for lvl, (blocks, attns, has_down, c) in enumerate(self.down):
            for rb in blocks:
                resblock(rb)
            for a in attns:
                attn(a)
            if has_down:
                conv(f"down.{lvl}.downsample.conv", c, c, 3)
        resblock(self.mid[0]), attn(self.mid[1]), resblock(self.mid[2])
        for lvl in range(self.num_resolutions):
            blocks, attns, has_up, c = self.up[lvl]
            for rb in blocks:
                resblock(rb)
            for a in attns:
                attn(a)
            if has_up:
                conv(f"up.{lvl}.upsample.conv", c, c, 3)
        vec2("norm_out", self.final_ch)
        conv("conv_out", self.out_ch, self.final_ch, 3)
        return s

    def random_state_dict(self, seed=1234):
        """Seeded random weights (no checkpoints exist offline): N(0, 1/fan_in) kernels, GN gamma near 1."""
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
        g = lambda k: sd[k].detach().to(device=dev, dtype=torch.float32).contiguous()  # noqa: E731
        w = {}

        def s16(t):                 # split-fp16 packing of one weight tensor with its own scale
            sc = ops.s16_weight_scale(t)
            return (ops.pack_conv_weight_s16(t, sc), sc, None)
        for k in ("temb.dense.0", "temb.dense.1"):
            w[k + ".weight"], w[k + ".bias"] = g(k + ".weight"), g(k + ".bias")
        w["conv_in.weight"] = ops.pack_conv_weight(g("conv_in.weight"), cin_pad=CIN_PAD)
        w["conv_in.bias"] = g("conv_in.bias")
        if 9 * self.in_channels <= CIN_PAD:       # 3 input channels: the 27 taps fit one 32-wide K chunk
            w["conv_in.weight.im2col"] = ops.pack_conv_in_weight_im2col(g("conv_in.weight"), CIN_PAD)
            # (stays on the fp32 MFMA tile kernel: the launch is output-store bound -- 268 MB at B = 8 -- and the gather form
            #  of the split arithmetic, with half the workgroups per CU, measured 146 us against 118 us)
        tw, tb = [], []
        for rb in self.res_blocks:
            n = rb.name
            for norm in ("norm1", "norm2"):
                w[f"{n}.{norm}.weight"], w[f"{n}.{norm}.bias"] = g(f"{n}.{norm}.weight"), g(f"{n}.{norm}.bias")
            w[f"{n}.conv1.weight"] = ops.pack_conv_weight(g(f"{n}.conv1.weight"))
            w[f"{n}.conv2.weight"] = ops.pack_conv_weight(g(f"{n}.conv2.weight"))
            w[f"{n}.conv2.bias"] = g(f"{n}.conv2.bias")
            if self.split16:
                # split-fp16 packing (fp32-grade products on the fp16 matrix pipe, ops.pack_conv_weight_s16); conv2 and
                # its fused shortcut share an accumulator, hence one scale
                w1, w2 = g(f"{n}.conv1.weight"), g(f"{n}.conv2.weight")
                wsk = g(f"{n}.nin_shortcut.weight") if rb.cin != rb.cout else None
                s1 = ops.s16_weight_scale(w1)
                s2 = ops.s16_weight_scale(w2) if wsk is None else ops.s16_weight_scale(w2, wsk)
                w[f"{n}.conv1.s16"] = (ops.pack_conv_weight_s16(w1, s1), s1, None)
                w[f"{n}.conv2.s16"] = (ops.pack_conv_weight_s16(w2, s2), s2,
                                       None if wsk is None else ops.pack_conv_weight_s16(wsk, s2))
                if wsk is not None:
                    w[f"{n}.nin_shortcut.s16"] = s16(wsk)          # un-fused form (8 x 8 level: gather kernel)
            # conv1 bias folded into the (concatenated) temb projection: h = conv1(.) + b1 + proj(temb)
            tw.append(g(f"{n}.temb_proj.weight"))
            tb.append(g(f"{n}.temb_proj.bias") + g(f"{n}.conv1.bias"))
            if rb.cin != rb.cout:
                w[f"{n}.nin_shortcut.weight"] = ops.pack_conv_weight(g(f"{n}.nin_shortcut.weight"))
                w[f"{n}.nin_shortcut.bias"] = g(f"{n}.nin_shortcut.bias")
                # fused form: the 1x1 shortcut rides along conv2 as extra K chunks, biases summed
                w[f"{n}.nin_shortcut.fused"] = ops.pack_skip_weight(g(f"{n}.nin_shortcut.weight"))
                w[f"{n}.conv2_plus_shortcut.bias"] = (g(f"{n}.conv2.bias") + g(f"{n}.nin_shortcut.bias")).contiguous()
        w["temb_proj_cat.weight"] = torch.cat(tw, 0).contiguous()
        w["temb_proj_cat.bias"] = torch.cat(tb, 0).contiguous()
        attns = [a for (_, at, _, _) in self.down for a in at] + [self.mid[1]] + \
                [a for lvl in self.up for a in self.up[lvl][1]]
        for a in attns:
            n = a.name
            w[f"{n}.norm.weight"], w[f"{n}.norm.bias"] = g(f"{n}.norm.weight"), g(f"{n}.norm.bias")
            wq = torch.cat([g(f"{n}.{p}.weight") for p in ("q", "k", "v")], 0)
            w[f"{n}.qkv.weight"] = ops.pack_conv_weight(wq)
            if self.split16:
                w[f"{n}.qkv.s16"] = s16(wq)
                w[f"{n}.proj_out.s16"] = s16(g(f"{n}.proj_out.weight"))
            w[f"{n}.qkv.bias"] = torch.cat([g(f"{n}.{p}.bias") for p in ("q", "k", "v")], 0).contiguous()
            # operand scales of the fused attention kernel (split-fp16 products need |operand| < 2^15): a STATIC bound of the
            # 1x1 convolutions' outputs, |W . GN(x) + b| <= max_o sum_c |W[o, c]| * (sqrt(n) max|gamma| + max|beta|) + max|b|
            # over the n = H*W*C/32 elements of a group (the bound _guard_normalised_operands uses), as powers of two
            gmax = math.sqrt(a.res * a.res * max(1, a.c // 32)) * float(sd[f"{n}.norm.weight"].detach().abs().max()) + \
                float(sd[f"{n}.norm.bias"].detach().abs().max())

            def _pow2(*names):
                bnd = max(float(sd[f"{n}.{p}.weight"].detach().float().abs().flatten(1).sum(1).max()) * gmax +
                          float(sd[f"{n}.{p}.bias"].detach().abs().max()) for p in names)
                return 2.0 ** (14 - math.ceil(math.log2(max(bnd, 1e-30))))
            w[f"{n}.attn_scales"] = (_pow2("q", "k"), _pow2("v"))
            w[f"{n}.proj_out.weight"] = ops.pack_conv_weight(g(f"{n}.proj_out.weight"))
            w[f"{n}.proj_out.bias"] = g(f"{n}.proj_out.bias")
        for lvl, (_, _, has_down, c) in enumerate(self.down):
            if has_down:
                w[f"down.{lvl}.downsample.conv.weight"] = ops.pack_conv_weight(g(f"down.{lvl}.downsample.conv.weight"))
                w[f"down.{lvl}.downsample.conv.bias"] = g(f"down.{lvl}.downsample.conv.bias")
                if self.split16:
                    w[f"down.{lvl}.downsample.conv.s16"] = s16(g(f"down.{lvl}.downsample.conv.weight"))
        for lvl, (_, _, has_up, c) in self.up.items():
            if has_up:
                w[f"up.{lvl}.upsample.conv.weight"] = ops.pack_conv_weight(g(f"up.{lvl}.upsample.conv.weight"))
                if self.split16:
                    wu = g(f"up.{lvl}.upsample.conv.weight")
                    su = ops.s16_weight_scale(wu)
                    w[f"up.{lvl}.upsample.conv.s16"] = (ops.pack_conv_weight_s16(wu, su), su, None)
                w[f"up.{lvl}.upsample.conv.bias"] = g(f"up.{lvl}.upsample.conv.bias")
        w["norm_out.weight"], w["norm_out.bias"] = g("norm_out.weight"), g("norm_out.bias")
        w["conv_out.weight"] = ops.pack_conv_weight(g("conv_out.weight"))
        w["conv_out.bias"] = g("conv_out.bias")
        if self.split16:
            self._guard_normalised_operands(sd, w)
        half = self.ch // 2
        freq = torch.exp(torch.arange(half, dtype=torch.float32) * -(math.log(10000) / (half - 1)))   # models.py:16-17
        w["temb.freq"] = freq.to(dev)
        self.w = w
        self._ws = None
        self._ws_by_stream = {}
        if getattr(self, "_graphs", None) is not None:
            self._graphs.reset()
        self._auto_graphs = None
        return self

    def _guard_normalised_operands(self, sd, w):
        """The split-fp16 kernels scale RAW operands per launch (ops.conv2d, ddnm_conv_desc::amax_in); a GroupNorm'd
        operand carries no run-time bound because it has a static one: |GN(x) * gamma + beta| <= sqrt(n) * max|gamma| +
        max|beta| over the n = H*W*C/32 elements of a group (swish does not increase it).  Checked here, once per
        checkpoint: a norm whose bound does not fit fp16 with a factor 2 to spare sends its consumer to the exact-fp32
        kernel (its split weights are dropped).  No shipped checkpoint is expected to trip this (gamma ~ 1: bound ~ 512);
        it makes "no operand-range limit" hold for ANY state dict."""
        groups = 32
        dropped = []

        def bound(norm, res, channels):
            g = sd[norm + ".weight"].detach().abs().max().item()
            b = sd[norm + ".bias"].detach().abs().max().item()
            return math.sqrt(res * res * max(1, channels // groups)) * g + b
        for rb in self.res_blocks:
            for norm, conv, ch in (("norm1", "conv1", rb.cin), ("norm2", "conv2", rb.cout)):
                if bound(f"{rb.name}.{norm}", rb.res, ch) >= 32768.0 and w.pop(f"{rb.name}.{conv}.s16", None) is not None:
                    dropped.append(f"{rb.name}.{conv}")
        attns = [a for (_, at, _, _) in self.down for a in at] + [self.mid[1]] + [a for lvl in self.up for a in self.up[lvl][1]]
        for a in attns:
            if bound(f"{a.name}.norm", a.res, a.c) >= 32768.0 and w.pop(f"{a.name}.qkv.s16", None) is not None:
                dropped.append(f"{a.name}.qkv")
        self.s16_dropped = dropped
        return dropped

    # ------------------------------------------------------------------ forward
    def _workspace(self, B):
        """GroupNorm scratch of the current stream (one per stream: see ddnm_amd/graph.py)."""
        key = torch.cuda.current_stream().cuda_stream
        ent = self._ws_by_stream.get(key)
        if ent is None or ent[1] < B:
            r = self.resolution
            max_partial = 0
            # bound over the (HW, C) pairs that occur: C <= max_ch at every resolution
            res = r
            for _ in range(self.num_resolutions):
                for c in (self.ch, self.max_ch):
                    max_partial = max(max_partial, ops.gn_nchunk(res * res, c))
                res //= 2
            ent = (ops.GroupNormWorkspace(self.device, B, self.max_ch, B * max_partial * 32 * 2), B)
            self._ws_by_stream[key] = ent
        self._ws = ent[0]
        return self._ws

    def _gn(self, x0, x1, name, want_amax=False):
        return ops.group_norm_affine(x0, x1, self.w[name + ".weight"], self.w[name + ".bias"], GN_EPS, self._ws,
                                     want_amax=want_amax)

    def _resblock(self, rb, x0, x1, tproj):
        w, n = self.w, rb.name
        # a block with a 1x1 shortcut reads its input RAW as well: the finalize launch of norm1 also emits the operand
        # bound of (x0, x1) for the split-fp16 kernels' range guard (ops.conv2d(raw_amax=...)), at no extra launch
        x_amax = None
        if rb.cin != rb.cout and self.split16:
            *gn1, x_amax = self._gn(x0, x1, n + ".norm1", want_amax=True)
            gn1 = tuple(gn1)
        else:
            gn1 = self._gn(x0, x1, n + ".norm1")
        h = ops.conv2d(x0, w[n + ".conv1.weight"], rb.cout, 3, src1=x1, gn=gn1, gn_silu=True,
                       badd=tproj[:, rb.temb_off:], badd_stride=self.temb_total, emit_stats=True,
                       weight_s16=w.get(n + ".conv1.s16"))
        gn2 = self._gn(h, None, n + ".norm2")
        if rb.cin != rb.cout:
            B, H, W, _ = h.t.shape
            fuse = ops.conv_fuses_skip(B, H, W, rb.cout, rb.cout)
            if (fuse and self.split16 and not ops.conv_runs_s16(B, H, W, rb.cout, rb.cout)
                    and ops.conv_runs_s16_gather(B, H, W, rb.cout, rb.cout)):
                fuse = False        # 8 x 8 level: two split-fp16 gather launches beat one fp32 MFMA launch with the fused shortcut
            if fuse:
                return ops.conv2d(h, w[n + ".conv2.weight"], rb.cout, 3, gn=gn2, gn_silu=True,
                                  bias=w[n + ".conv2_plus_shortcut.bias"], skip=(x0, x1),
                                  skip_weight=w[n + ".nin_shortcut.fused"], emit_stats=True,
                                  weight_s16=w.get(n + ".conv2.s16"), raw_amax=x_amax)
            xs = ops.conv2d(x0, w[n + ".nin_shortcut.weight"], rb.cout, 1, src1=x1, bias=w[n + ".nin_shortcut.bias"],
                            weight_s16=w.get(n + ".nin_shortcut.s16"), raw_amax=x_amax)
        else:
            assert x1 is None
            xs = x0
        s16 = w.get(n + ".conv2.s16")
        if s16 is not None:
            s16 = (s16[0], s16[1], None)         # un-fused shortcut: the 3x3 weights alone (same packing, same scale)
        return ops.conv2d(h, w[n + ".conv2.weight"], rb.cout, 3, gn=gn2, gn_silu=True, bias=w[n + ".conv2.bias"],
                          res=xs, emit_stats=True, weight_s16=s16)

    def _attn(self, a, x):
        w, n = self.w, a.name
        B, H, W, C = x.t.shape
        T = H * W
        gn = self._gn(x, None, n + ".norm")
        qkv = ops.conv2d(x, w[n + ".qkv.weight"], 3 * C, 1, gn=gn, gn_silu=False, bias=w[n + ".qkv.bias"],
                         weight_s16=w.get(n + ".qkv.s16"))
        sc = w.get(n + ".attn_scales")
        if sc is not None and self.fused_attn and T <= self.fused_attn_max_tokens and ops.attn_fused_supported(T, C):
            # one launch, no [T][T] tensor in HBM (csrc/attn_d512.hip)
            o = ops.attn_fused(qkv, B, T, C, sc[0], sc[1], float(int(C) ** (-0.5))).view(B, H, W, C)
            return ops.conv2d(o, w[n + ".proj_out.weight"], C, 1, bias=w[n + ".proj_out.bias"], res=x, emit_stats=True,
                              weight_s16=w.get(n + ".proj_out.s16"))
        S = torch.empty(B, T, T, dtype=torch.float32, device=qkv.device)
        q, k, v = qkv.view(-1)[0:], qkv.view(-1)[C:], qkv.view(-1)[2 * C:]
        ops.bgemm(q, k, S, T, T, C, lda=3 * C, ldb=3 * C, ldc=T, transb=True, batch=B,
                  sA=(T * 3 * C, 0), sB=(T * 3 * C, 0), sC=(T * T, 0))
        ops.softmax_rows_(S, B * T, T, T, float(int(C) ** (-0.5)))
        o = torch.empty(B, H, W, C, dtype=torch.float32, device=qkv.device)
        ops.bgemm(S, v, o, T, C, T, lda=T, ldb=3 * C, ldc=C, transb=False, batch=B,
                  sA=(T * T, 0), sB=(T * 3 * C, 0), sC=(T * C, 0))
        return ops.conv2d(o, w[n + ".proj_out.weight"], C, 1, bias=w[n + ".proj_out.bias"], res=x, emit_stats=True,
                          weight_s16=w.get(n + ".proj_out.s16"))

    def enable_graphs(self, two_streams=False):
        """Replay the forward from a captured hipGraph (ddnm_amd/graph.py)."""
        from ..graph import GraphedForward
        self._graphs = GraphedForward(lambda x, t, y: self._forward_eager(x, t), two_streams=two_streams)
        return self

    def disable_graphs(self):
        self._graphs = None
        return self

    def auto_graphs(self, max_batch=2):
        """Replay forwards of at most `max_batch` images from a captured hipGraph, decided per call (0: never).  The
        reference's shipped configs sample with batch_size 1 (configs/celeba_hq.yml:34-35); a forward is ~250 launches
        whose host cost (~15 us each through ctypes) then exceeds their GPU time, and `cudnn.benchmark` was the
        reference's own small-batch lever (main.py:145).  The runner (`Diffusion`) switches this on."""
        self.auto_graph_max_batch = int(max_batch)
        self._auto_graphs = None
        return self

    @property
    def max_forward_batch(self):
        """Chunk size of forward(): the largest activation is `ch` channels of fp32 at full resolution."""
        return ops.max_launch_batch(self.resolution * self.resolution * self.ch * 4)

    def forward(self, x, t):
        mb = self.max_forward_batch
        if x.shape[0] > mb:          # more images than one launch can address: micro-batches, concatenated
            return torch.cat([self.forward(x[i:i + mb], t[i:i + mb]) for i in range(0, x.shape[0], mb)], 0)
        if getattr(self, "_graphs", None) is not None:
            return self._graphs(x, t, None)
        if x.shape[0] <= getattr(self, "auto_graph_max_batch", 0):
            if getattr(self, "_auto_graphs", None) is None:
                from ..graph import GraphedForward
                self._auto_graphs = GraphedForward(lambda x_, t_, y_: self._forward_eager(x_, t_), two_streams=False)
            return self._auto_graphs(x, t, None)
        return self._forward_eager(x, t)

    def _forward_eager(self, x, t):
        if self.w is None:
            raise RuntimeError("load_state_dict() must be called before forward()")
        assert x.shape[2] == x.shape[3] == self.resolution
        w = self.w
        B = x.shape[0]
        self._workspace(B)
        t = t.to(device=x.device, dtype=torch.float32).contiguous()
        emb = ops.timestep_embedding(t, w["temb.freq"], order=0)
        temb = ops.linear(emb, w["temb.dense.0.weight"], w["temb.dense.0.bias"])
        temb = ops.linear(temb, w["temb.dense.1.weight"], w["temb.dense.1.bias"], silu_in=True)
        tproj = ops.linear(temb, w["temb_proj_cat.weight"], w["temb_proj_cat.bias"], silu_in=True)

        if "conv_in.weight.im2col" in w:
            # conv_in as a 1x1 convolution over the im2col'ed image: K = 32 instead of 9 x 32 zero-padded channels
            xin = ops.nchw_im2col3x3_pad(x.contiguous(), CIN_PAD)
            hs = [ops.conv2d(xin, w["conv_in.weight.im2col"], self.ch, 1, bias=w["conv_in.bias"], emit_stats=True)]
        else:
            xin = ops.nchw_to_nhwc_pad(x.contiguous(), CIN_PAD)
            hs = [ops.conv2d(xin, w["conv_in.weight"], self.ch, 3, bias=w["conv_in.bias"], emit_stats=True)]
        for lvl, (blocks, attns, has_down, c) in enumerate(self.down):
            for ib, rb in enumerate(blocks):
                h = self._resblock(rb, hs[-1], None, tproj)
                if attns:
                    h = self._attn(attns[ib], h)
                hs.append(h)
            if has_down:
                n = f"down.{lvl}.downsample.conv"
                src = hs[-1]
                # F.pad(x, (0,1,0,1)) + 3x3 stride 2 (models.py:68-71): pad=0 on top/left, the
                # bottom/right zero row/column comes from the loader's bounds check
                hs.append(ops.conv2d(src, w[n + ".weight"], c, 3, bias=w[n + ".bias"], stride=2, pad=0,
                                     out_hw=(src.t.shape[1] // 2, src.t.shape[2] // 2), emit_stats=True,
                                     weight_s16=w.get(n + ".s16")))
        h = hs[-1]
        h = self._resblock(self.mid[0], h, None, tproj)
        h = self._attn(self.mid[1], h)
        h = self._resblock(self.mid[2], h, None, tproj)
        for lvl in reversed(range(self.num_resolutions)):
            blocks, attns, has_up, c = self.up[lvl]
            for ib, rb in enumerate(blocks):
                h = self._resblock(rb, h, hs.pop(), tproj)
                if attns:
                    h = self._attn(attns[ib], h)
            if has_up:
                n = f"up.{lvl}.upsample.conv"
                h = ops.conv2d(h, w[n + ".weight"], c, 3, bias=w[n + ".bias"], ups=True, emit_stats=True,
                               weight_s16=w.get(n + ".s16"))
        gn = self._gn(h, None, "norm_out")
        return ops.conv2d(h, w["conv_out.weight"], self.out_ch, 3, gn=gn, gn_silu=True, bias=w["conv_out.bias"],
                          out_nchw=True)

    def __call__(self, x, t):
        return self.forward(x, t)
