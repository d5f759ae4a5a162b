/*
 * ddnm_hip.h -- C ABI of libddnm_hip.so: hand-written HIP kernels (gfx950 / CDNA4)
 * for the DDNM sampling hot path.
 *
 * The reference (wyhuai/DDNM) has no FFI of its own: it is pure PyTorch and every
 * hot op is an ATen call.  Each entry point below replaces
 * the ATen call sites cited next to it; INTEGRATION.md shows the ctypes stub a
 * maintainer of the reference would add.
 *
 * Conventions (all entry points):
 *   - plain device pointers + sizes; no torch types, no allocation, no host sync,
 *     no global state; work is enqueued on `stream` (hipStream_t passed as void*).
 *   - returns 0 on success, a positive hipError_t on a runtime error, or a negative
 *     DDNM_E_* code on an argument the kernel family does not support.
 *   - activations inside the UNet are NHWC fp32 ("pixel-major": [B][H][W][C]);
 *     the sampler-side images are NCHW fp32 exactly like the reference tensors.
 *   - thread-safe for concurrent calls on different streams / devices.
 */
#ifndef DDNM_HIP_H
#define DDNM_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DDNM_E_BADARG (-1)   /* null pointer / non-positive size / misaligned */
#define DDNM_E_SHAPE (-2)    /* shape not supported by this kernel family */

int ddnm_version(void);                 /* ABI version, currently 6 (bumped on every struct / prototype change) */
const char* ddnm_build_digest(void);    /* sha256 of the sources + flags this binary was built from (build.py) */
int ddnm_sizeof(int which);             /* sizeof of 0: ddnm_conv_desc, 1: ddnm_gemm_desc, 2: ddnm_conv16_desc,
                                           3: ddnm_step_scalars as compiled into the binary (-1: unknown index) */
const char* ddnm_error_string(int code);

/* ------------------------------------------------------------------------- *
 * Implicit-GEMM convolution, fp32 in / fp32 accumulate on v_mfma_f32_32x32x2_f32.
 * Replaces nn.Conv2d 3x3 s1 p1 (guided_diffusion/models.py:87,96,225,295),
 * the asymmetric-pad 3x3 s2 Downsample (models.py:61-71), 1x1 convs
 * (models.py:109,143-162) and, through the fused prologue/epilogue, the
 * GroupNorm+swish in front of each conv (models.py:117-118,123-124,166,338-339),
 * F.interpolate(nearest x2) (models.py:48), torch.cat([h, skip]) (models.py:331),
 * the temb add (models.py:121) and the residual add (models.py:134,189).
 *
 *   out[b,oy,ox,n] = bias[n] + badd[b*badd_stride + n] + res[b,oy,ox,n]
 *                  + sum_{ky,kx,c} W[n,ky,kx,c] * act(in[b, oy*stride-pad+ky, ox*stride-pad+kx, c])
 *   in  = concat_c(src0 (C0 ch), src1 (C1 ch)), optionally nearest-x2 upsampled (ups=1),
 *   act(v) = silu?(v*gn_scale[b,c] + gn_shift[b,c]) when gn_scale != NULL (zero padding
 *            is applied AFTER act, like the reference which pads the activated tensor).
 * ------------------------------------------------------------------------- */
typedef struct ddnm_conv_desc {
    const float* src0;      /* NHWC [B][Hs][Ws][C0] */
    const float* src1;      /* NHWC [B][Hs][Ws][C1] or NULL */
    const float* weight;    /* packed [Cout_pad][KH*KW][Cin] (O,ky,kx,I), Cout_pad = ceil(Cout/tileN)*tileN */
    const float* bias;      /* [Cout] or NULL */
    const float* badd;      /* per-sample addend, row b at badd + b*badd_stride, or NULL */
    const float* res;       /* NHWC [B][Ho][Wo][Cout] or NULL */
    const float* gn_scale;  /* [B][Cin] or NULL */
    const float* gn_shift;  /* [B][Cin] (required iff gn_scale) */
    float* out;             /* NHWC [B][Ho][Wo][Cout], or NCHW [B][Cout][Ho][Wo] if out_nchw */
    int32_t B, Hin, Win;    /* logical input size (after the optional x2 upsample) */
    int32_t C0, C1;         /* C0 % 32 == 0, C1 % 32 == 0 (C1 = 0 without src1) */
    int32_t Cout;
    int32_t ksize;          /* 1 or 3 */
    int32_t stride;         /* 1 or 2 */
    int32_t pad;            /* top/left zero padding; bottom/right is implied by Ho, Wo */
    int32_t Ho, Wo;
    int32_t ups;            /* 1: src is [Hin/2][Win/2], read through nearest x2 */
    int32_t gn_silu;        /* 1: swish after the GroupNorm affine; 0: affine only */
    int32_t out_nchw;
    int32_t badd_stride;
    int32_t tile;           /* 0 auto; 1: 128x128, 2: 64x64, 3: 128x32 (M x N per workgroup) */
    float* workspace;       /* split-K scratch (may be NULL: the kernel then runs unsplit) */
    int64_t workspace_floats;
    int32_t res_ups;        /* 1: `res` is [B][Ho/2][Wo/2][Cout], added through a nearest x2 upsample
                               (x_upd of an `up=True` ResBlock, guided_diffusion/unet.py:237-242) */
    int32_t src_f16;        /* ddnm_conv3x3_f16_f32 only: src0/src1 point to IEEE fp16 NHWC tensors (the output of
                               ddnm_gn_apply_f16); gn_scale must then be NULL */
    float* stats_out;       /* optional [B*tiles][Cout][2]: per-(M tile, channel) sum / sum of squares of `out`,
                               tiles = ddnm_conv2d_f32_stats_tiles(d) per image; feeds ddnm_gn_finalize_tiles_f32 so
                               the consumer's GroupNorm never re-reads the tensor */
    /* fused 1x1 shortcut of a residual block (models.py:109,128-132; unet.py:222,256), 3x3 halo launches only:
     * out += W_skip . concat_c(skip0, skip1) evaluated on the RAW tensors at the output pixel */
    const float* skip0;     /* NHWC [B][Ho][Wo][SC0] or NULL */
    const float* skip1;     /* NHWC [B][Ho][Wo][SC1] or NULL */
    const float* skip_weight; /* packed [Cout_pad][SC0+SC1] (fp16 for ddnm_conv3x3_f16_f32) */
    int32_t SC0, SC1;
    float acc_scale;        /* ddnm_conv3x3_s16_f32 only: power of two that multiplies the accumulator before bias / residual
                               (undoes the operand pre-scaling of the split form); ignored by the other entry points */
    int32_t flags;          /* bit 0 (DDNM_CONV_ONE_TILE): ddnm_conv3x3_s16_f32 runs the one-tile-per-workgroup kernel even where
                               the persistent form (>= 2 tiles per CU) applies -- same results bit for bit; for A/B timing */
    /* ABI 5 -- operand-range guard of the split forms (ddnm_conv3x3_s16_f32, ddnm_conv_gather_s16_f32; ignored by every
     * other entry point).  fp16 carries |v| < 65504 only, fp32 -- the arithmetic the reference runs -- does not care, so
     * operands the kernel reads RAW (no GroupNorm in front: src0 / src1 when gn_scale is NULL, skip0 / skip1 always) are
     * scaled per launch and image by a power of two derived ON THE DEVICE from an upper bound of their magnitude:
     *   amax_in = [B][DDNM_AMAX_N] non-negative floats, max_i amax_in[b][i] >= max |raw operand of image b|
     * (ddnm_gn_finalize_tiles_amax_f32 emits them from the producer's GroupNorm partials for free, ddnm_amax_bound_f32
     * from partials or from the tensor itself).  The kernel multiplies the raw operand by 2^k, k = 14 - exponent(bound)
     * (bound lands in [2^14, 2^15): no overflow for ANY fp32 input magnitude, and uniformly tiny tensors keep fp32
     * grade because hi AND lo become normal fp16 numbers), and the accumulator by 2^-k (exact).  A launch whose main
     * operand is GroupNorm'd and whose fused shortcut is raw shares one accumulator: k is then clamped to <= 0 (scale
     * down only) and applied to both operands.  NULL = no scaling (operands must lie within fp16 range). */
    const float* amax_in;
} ddnm_conv_desc;

#define DDNM_CONV_ONE_TILE 1
#define DDNM_AMAX_N 32   /* bound words per image (= the GroupNorm group count of both networks) */

int ddnm_conv2d_f32(const ddnm_conv_desc* d, void* stream);
/* N-tile (32 or 64 or 128) the kernel will use for this desc: Cout_pad of the packed weight. */
int ddnm_conv2d_f32_tile_n(const ddnm_conv_desc* d);
/* Scratch floats the auto plan wants for this desc (0: no split-K); low-resolution layers whose tile
 * grid cannot fill 256 CUs split their channel chunks over several workgroups per tile. */
int64_t ddnm_conv2d_f32_workspace_floats(const ddnm_conv_desc* d);
/* M tiles per image of the auto plan if this launch can emit `stats_out` (0: split-K or NCHW launch). */
int ddnm_conv2d_f32_stats_tiles(const ddnm_conv_desc* d);
/* 1 if this launch would run the 3x3 halo kernel and can therefore take the fused shortcut fields. */
int ddnm_conv2d_f32_fuses_skip(const ddnm_conv_desc* d);

/* Output convolution with <= 4 output channels (conv_out of the celeba `Model`, guided_diffusion/models.py:295-299): same
 * descriptor with ksize 3, stride 1, pad 1, out_nchw = 1, one source, Cin % 32 == 0, Win % 32 == 0, Hin % 8 == 0, optional
 * fused GroupNorm affine + swish and bias; no residual / shortcut / statistics.  HBM-bound vector-ALU kernel (on the
 * MFMA tile kernel the 3 channels are padded to a 32-wide N tile: 10x the matrix work). */
int ddnm_conv3x3_small_cout_f32(const ddnm_conv_desc* d, void* stream);
int ddnm_conv3x3_small_cout_f32_supported(const ddnm_conv_desc* d);

/* 3x3 / stride 1 / pad 1 convolution with fp16 MFMA operands (v_mfma_f32_32x32x16_f16), fp32 accumulate:
 * the reference's `use_fp16` torso (guided_diffusion/unet.py:619-625, fp16_util.py:15-22).  Same descriptor;
 * `weight` points to the (O,ky,kx,I)-packed weights stored as IEEE fp16, activations / bias / residual /
 * output stay fp32 in HBM (rounded to fp16 while the halo tile is staged).  Needs Cin % 64 == 0,
 * Cout % 128 == 0, Ho*Wo % 256 == 0; `_supported` tells whether a desc qualifies (else use ddnm_conv2d_f32). */
int ddnm_conv3x3_f16_f32(const ddnm_conv_desc* d, void* stream);
int ddnm_conv3x3_f16_supported(const ddnm_conv_desc* d);
int64_t ddnm_conv3x3_f16_workspace_floats(const ddnm_conv_desc* d);
int ddnm_conv3x3_f16_stats_tiles(const ddnm_conv_desc* d);

/* 3x3 / stride 1 / pad 1 convolution of fp32 tensors with fp32-GRADE products on the fp16 matrix pipe (the celeba
 * `Model`, guided_diffusion/models.py:87,96,225 -- the reference runs it in fp32): every operand value v is carried as
 * hi = rn16(v), lo = rn16(v - hi), a product is hi*hi' + hi*lo' + lo*hi' (three v_mfma_f32_32x32x16_f16 whose fp16 x fp16
 * products are exact in the fp32 accumulator; the dropped lo*lo' term is 2^-22 relative), so the operand error is
 * <= 2^-22 -- below the accumulation noise of an fp32 dot product of this length -- at 16/3 of the fp32 MFMA rate.
 * Same descriptor as ddnm_conv2d_f32 (GroupNorm + swish prologue, concat, x2 upsample, fused shortcut, bias / badd /
 * residual, statistics, split-K) except:
 *   weight / skip_weight = the SPLIT packing: per (output row, tap, 32-channel chunk) 32 hi halfs then 32 lo halfs of
 *     2^s * W (the byte size of the fp32 (O,ky,kx,I) array), Cout padded to 128; one power of two 2^s per launch that
 *     brings max|W| into [2^13, 2^14) so that lo is a normal fp16 number;
 *   acc_scale = 2^-s (times the inverse of the kernel's compile-time activation pre-scale, ddnm_conv3x3_s16_act_scale()).
 * Needs Cin % 32 == 0, C0 % 32 == 0, Cout % 128 == 0, Ho*Wo % 256 == 0 (else: ddnm_conv2d_f32). */
int ddnm_conv3x3_s16_f32(const ddnm_conv_desc* d, void* stream);
int ddnm_conv3x3_s16_supported(const ddnm_conv_desc* d);
int ddnm_conv3x3_s16_persistent(const ddnm_conv_desc* d);   /* ABI 7 -- 1: the launch runs the persistent form (>= 2 tiles per CU: one
                                                                workgroup per CU walks its tiles; same results bit for bit) */
int64_t ddnm_conv3x3_s16_workspace_floats(const ddnm_conv_desc* d);
int ddnm_conv3x3_s16_stats_tiles(const ddnm_conv_desc* d);
float ddnm_conv3x3_s16_act_scale(void);   /* compile-time activation pre-scale: 1 since ABI 5 (the scale is per launch and
                                             image, derived on the device from ddnm_conv_desc::amax_in) */

/* The same arithmetic for the layers the 3x3 halo kernel does not take, as a per-tap gather (csrc/conv_gather_s16.hip):
 * Downsample's 3x3 stride 2 with (0,1,0,1) padding (guided_diffusion/models.py:61-71), the 1x1 convolutions of the
 * attention blocks and un-fused nin_shortcut (models.py:109,143-162), and 3x3 / stride 1 on the 8 x 8 level.  Same
 * descriptor, weight packing and acc_scale as ddnm_conv3x3_s16_f32 (ksize 1 or 3, stride 1 or 2, pad as given; GroupNorm
 * prologue, concat, bias / badd / residual, statistics, split-K); no fused shortcut.
 * Needs C0 % 32 == 0, C1 % 32 == 0, Cout % 64 == 0, Ho*Wo % 64 == 0 (else: ddnm_conv2d_f32). */
int ddnm_conv_gather_s16_f32(const ddnm_conv_desc* d, void* stream);
int ddnm_conv_gather_s16_supported(const ddnm_conv_desc* d);
int64_t ddnm_conv_gather_s16_workspace_floats(const ddnm_conv_desc* d);
int ddnm_conv_gather_s16_stats_tiles(const ddnm_conv_desc* d);

/* 1x1 convolution with fp16 MFMA operands, fp32 accumulate / output: the attention blocks' qkv and proj_out
 * Conv1d(k=1) and un-fused 1x1 shortcuts of the `use_fp16` torso (guided_diffusion/unet.py:222,283-289,301-308).
 * Same descriptor with ksize = 1, stride 1, pad 0; `weight` = (O,1,I)-packed fp16; src fp32, or fp16 with src_f16 = 1
 * (output of ddnm_gn_apply_f16 -- there is no fused GroupNorm prologue: gn_scale must be NULL).  Needs Cin % 64 == 0,
 * Cout % 128 == 0, B*Ho*Wo % 256 == 0; statistics only when Ho*Wo % 256 == 0. */
int ddnm_conv1x1_f16_f32(const ddnm_conv_desc* d, void* stream);
int ddnm_conv1x1_f16_supported(const ddnm_conv_desc* d);
int64_t ddnm_conv1x1_f16_workspace_floats(const ddnm_conv_desc* d);
int ddnm_conv1x1_f16_stats_tiles(const ddnm_conv_desc* d);

/* ------------------------------------------------------------------------- *
 * fp16-ACTIVATION path of the `use_fp16` torso (guided_diffusion/unet.py:619-625,655-663; fp16_util.py:15-22):
 * every tensor in HBM is fp16 NHWC, like the reference's `h.type(self.dtype)` tensors; accumulation, GroupNorm
 * statistics, softmax stay fp32.
 *
 * ddnm_conv16: 3x3 (stride 1, pad 1) or 1x1 convolution, v_mfma_f32_32x32x16_f16, 256 x 256 tiles staged by
 * LDS-DMA (csrc/conv16.hip).  Replaces the nn.Conv2d / Conv1d(k=1) of ResBlock / AttentionBlock
 * (unet.py:196-222,283-308) with, fused: nearest x2 upsample of the operand (ups; unet.py:104-109), the 1x1
 * skip_connection as extra K chunks (unet.py:222,256), bias, residual add (through a nearest x2 upsample for
 * `up=True` blocks) and the GroupNorm partials of the output.
 *   out[b,y,x,n] = fp16( bias[n] + res[b,y,x,n] + sum W[n,ky,kx,c] * src[b, y+ky-1, x+kx-1, c]
 *                        + sum W_skip[n,c] * concat_c(skip0, skip1)[b,y,x,c] )
 * act(v) = silu?(v*gn_scale[b,c] + gn_shift[b,c]) when gn_scale != NULL (3x3 only; applied inside LDS after the
 * LDS-DMA, zero padding AFTER act like the reference), identity otherwise (operand already activated by
 * ddnm_gn_apply_h16).  The operand may be the channel concat of `src` (C0 channels) and `src1`.
 * Needs Cin % 64 == 0, Cout % 64 == 0; 3x3: W % 16 == 0 and H*W % 128 == 0 (smaller images: ddnm_im2col3x3_h16 +
 * ksize 1).  ksize 1 treats src as a flat [B*H*W][Cin] matrix (any row count).
 * ------------------------------------------------------------------------- */
typedef struct ddnm_conv16_desc {
    const void* src;          /* fp16 NHWC [B][Hs][Ws][C0] (C0 = Cin without src1); Hs = H/2 when ups */
    const void* weight;       /* fp16 packed [ceil(Cout/256)*256][ksize*ksize][Cin] (O,ky,kx,I) */
    const float* bias;        /* [Cout] or NULL */
    const void* res;          /* fp16 NHWC [B][H][W][Cout] ([B][H/2][W/2][Cout] when res_ups) or NULL */
    const void* skip0;        /* fp16 NHWC [B][H][W][SC0] raw shortcut input or NULL (3x3 only) */
    const void* skip1;        /* fp16 NHWC [B][H][W][SC1] or NULL */
    const void* skip_weight;  /* fp16 [ceil(Cout/256)*256][SC0+SC1] */
    void* out;                /* fp16 NHWC [B][H][W][Cout]; fp32 NCHW [B][Cout][H][W] when out_nchw_f32 */
    float* stats_out;         /* optional [B*tiles][Cout][2] GroupNorm partials of the ROUNDED output,
                                 tiles = ddnm_conv16_stats_tiles(d); feeds ddnm_gn_finalize_tiles_f32 */
    float* workspace;         /* split-K slabs, ddnm_conv16_workspace_floats(d) floats (0: not needed) */
    int64_t workspace_floats;
    int32_t B, H, W;          /* OUTPUT size (= input size; the operand is H/2 x W/2 when ups) */
    int32_t Cin, Cout;
    int32_t ksize;            /* 1 or 3 */
    int32_t ups, res_ups;
    int32_t SC0, SC1;
    const void* src1;         /* fp16 NHWC [B][Hs][Ws][Cin-C0] second tensor of the concat or NULL (torch.cat, unet.py:661) */
    const float* gn_scale;    /* [B][Cin] fp32 or NULL: fused GroupNorm(+FiLM) affine of the operand (3x3 only) */
    const float* gn_shift;    /* [B][Cin] (required iff gn_scale) */
    int32_t C0;               /* channels of `src` when src1 != NULL (C0 % 64 == 0) */
    int32_t gn_silu;          /* 1: swish after the affine */
    int32_t out_nchw_f32;     /* 1: the network's output convolution (unet.py:627-631, `.type(x.dtype)` :664): 3x3,
                                 Cout <= 32 (weight packed to 32 rows or more), fp32 NCHW result, no res / skip / stats */
    int32_t reserved;
    /* Optional: finalize the CONSUMER's GroupNorm inside the split-K reduction pass.  When ddnm_conv16_fuses_fin(d)
       == 1 the launch also writes the affine of GroupNorm(fin_groups, Cout, fin_eps)(+ FiLM rows [s | t]) over ITS OWN
       OUTPUT -- scale / shift [B][Cout], what ddnm_gn_finalize_tiles_f32 would produce from stats_out -- so the
       convolution that consumes the output needs no finalize launch in between (guided_diffusion/nn.py:17-19,
       unet.py:196-205,248-251).  stats_out then holds ONE tile per image (ddnm_conv16_stats_tiles(d) == 1). */
    const float* fin_gamma;   /* [Cout] or NULL (no fused finalize) */
    const float* fin_beta;    /* [Cout] */
    const float* fin_film;    /* rows [s(0..Cout) | t(0..Cout)], row b at fin_film + b*fin_film_stride, or NULL */
    float* fin_scale;         /* [B][Cout] */
    float* fin_shift;         /* [B][Cout] */
    float fin_eps;
    int32_t fin_film_stride;
    int32_t fin_groups;
    int32_t reserved2;
} ddnm_conv16_desc;

int ddnm_conv16(const ddnm_conv16_desc* d, void* stream);
int ddnm_conv16_fuses_fin(const ddnm_conv16_desc* d);     /* 1: this launch will write fin_scale / fin_shift */
int ddnm_conv16_supported(const ddnm_conv16_desc* d);
int64_t ddnm_conv16_workspace_floats(const ddnm_conv16_desc* d);
int ddnm_conv16_stats_tiles(const ddnm_conv16_desc* d);   /* 0: this launch cannot emit stats_out */

/* HBM-bound stages of the fp16-activation path (csrc/act16.hip), all fp16 NHWC:
 *   gn_apply:  out = fp16(act(concat_c(src0, src1) * scale[b,c] + shift[b,c])), act = swish when silu (scale NULL:
 *              plain copy / concat); pool = 1 additionally averages 2x2 pixels (AvgPool2d of `down=True` ResBlocks,
 *              guided_diffusion/unet.py:237-242): out is [B][H/2][W/2][C].  Replaces GroupNorm32 + SiLU
 *              (nn.py:12-19; unet.py:196-205,226-230) and torch.cat([h, hs.pop()], 1) (unet.py:661).
 *   im2col:    col[(b*H*W + p)][tap*C + c] of a 3x3 / pad 1 convolution input with the same prologue (8x8 level).
 *   nchw_to_nhwc: fp32 NCHW [B][C][HW] -> fp16 NHWC [B][HW][cpad], zero padded (`x.type(self.dtype)`, unet.py:655).
 *   gn_stats:  per-(pixel tile, channel) sum / sum of squares of an fp16 tensor, layout of `stats_out`. */
int ddnm_gn_apply_h16(const void* src0, const void* src1, const float* scale, const float* shift, void* out, int32_t B,
                      int32_t H, int32_t W, int32_t C0, int32_t C1, int32_t silu, int32_t pool, void* stream);
int ddnm_im2col3x3_h16(const void* src0, const void* src1, const float* scale, const float* shift, void* out, int32_t B,
                       int32_t H, int32_t W, int32_t C0, int32_t C1, int32_t silu, void* stream);
int ddnm_nchw_to_nhwc_h16(const float* x, void* out, int32_t B, int32_t C, int32_t HW, int32_t cpad, void* stream);
int ddnm_gn_stats_h16(const void* src, float* stats, int32_t B, int32_t HW, int32_t C, int32_t tiles, void* stream);

/* Fused multi-head self-attention, head dim 64 (QKVAttentionLegacy, guided_diffusion/unet.py:328-354 inside
 * AttentionBlock :259-308): out[b,t,h*64+d] = sum_s softmax_s(q_t . k_s / 8) v_s[d] with
 * qkv fp16 [B][T][3C] laid out head-major (channel = h*192 + {q,k,v}*64 + d), out fp16 [B][T][C]; fp32 softmax,
 * probabilities rounded to fp16 like `.type(weight.dtype)`; no [T][T] tensor in HBM.  T % 64 == 0, C % 64 == 0. */
int ddnm_attn16_d64(const void* qkv, void* out, int32_t B, int32_t T, int32_t C, void* stream);
/* ABI 6 -- the same, also writing lse[b][h][t] = log2 sum_s exp2(q_t . k_s * log2(e) / 8) (fp32 [B][C/64][T]) for the
 * backward pass, and that backward pass (input gradient of AttentionBlock's attention inside the classifier-guidance
 * gradient, diffusion.py:183-189): dqkv fp16 [B][T][3C] in the layout of qkv from qkv, the forward output o, its
 * gradient dO (fp16 [B][T][C]) and lse; the probabilities are recomputed tile by tile, dsum [B][C/64][T] is scratch
 * (rowsum(dO . o)).  No [T][T] tensor in HBM. */
int ddnm_attn16_d64_lse(const void* qkv, void* out, float* lse, int32_t B, int32_t T, int32_t C, void* stream);
int ddnm_attn16_d64_bwd(const void* qkv, const void* o, const void* dO, const float* lse, float* dsum, void* dqkv,
                        int32_t B, int32_t T, int32_t C, void* stream);

/* ------------------------------------------------------------------------- *
 * GroupNorm statistics -> per-(sample, channel) affine for the conv prologue.
 * Replaces torch.nn.GroupNorm(32, C, eps) (models.py:32-33; guided_diffusion/nn.py:17-19).
 *   stats:    partial (sum, sumsq) per (b, chunk, group) in double, deterministic order
 *   finalize: scale[b,c] = rstd[b,g(c)]*gamma[c]; shift[b,c] = beta[c] - mean*rstd*gamma[c]
 * src is the concat of two NHWC tensors like the conv input (C1 = 0: single source).
 * ------------------------------------------------------------------------- */
int ddnm_gn_stats_f32(const float* src0, const float* src1, int32_t B, int32_t HW, int32_t C0, int32_t C1,
                      int32_t groups, double* partial /* [B][nchunk][groups][2] */, int32_t nchunk, void* stream);
int ddnm_gn_nchunk(int32_t HW, int32_t C);   /* chunk count `partial` must be sized for */
int ddnm_gn_finalize_f32(const double* partial, int32_t nchunk, const float* gamma, const float* beta,
                         int32_t B, int32_t HW, int32_t C, int32_t groups, float eps,
                         float* scale /* [B][C] */, float* shift /* [B][C] */,
                         const float* film /* optional FiLM rows [s(0..C) | t(0..C)]: GN(x)*(1+s)+t, unet.py:248-251 */,
                         int32_t film_stride, float* mean_rstd /* optional [B][groups][2], kept for backward */,
                         void* stream);

/* GroupNorm affine from the partials a convolution epilogue emitted (`stats_out`); the input may be the
 * channel concat of two tensors (part1 / tpi1 / C1, or NULL / 0 / 0). */
/* GroupNorm affine (+ swish) applied once and written as fp16 for ddnm_conv3x3_f16_f32(src_f16 = 1):
 * out[b,p,c] = fp16(act(concat_c(src0, src1)[b,p,c] * scale[b,c] + shift[b,c])), identical rounding to the fused
 * prologue.  C0 % 8 == 0, C1 % 8 == 0.  Layers with >= 256 output channels use this instead of the fused prologue,
 * which repeats the v_exp / v_rcp work once per 128-output-channel tile. */
int ddnm_gn_apply_f16(const float* src0, const float* src1, const float* scale, const float* shift, void* out_f16,
                      int32_t B, int32_t HW, int32_t C0, int32_t C1, int32_t silu, void* stream);
/* im2col of a 3x3 / stride 1 / pad 1 input with the GroupNorm affine (+ swish) prologue, fp16 output
 * col[B*H*W][9*(C0+C1)] (column = tap*C + c, zero outside the image); scale / shift may both be NULL.
 * The 8x8 level of the fp16 torso then runs as one ddnm_conv1x1_f16_f32 GEMM with K = 9*C (the (O,ky,kx,I)-packed
 * 3x3 weights ARE the [Cout][9*C] matrix).  C0 % 8 == 0, C1 % 8 == 0. */
int ddnm_im2col3x3_f16(const float* src0, const float* src1, const float* scale, const float* shift, void* out_f16,
                       int32_t B, int32_t H, int32_t W, int32_t C0, int32_t C1, int32_t silu, void* stream);
int ddnm_gn_finalize_tiles_f32(const float* part0, int32_t tiles_per_img0, int32_t C0, const float* part1,
                               int32_t tiles_per_img1, int32_t C1, const float* gamma, const float* beta, int32_t B,
                               int32_t HW, int32_t groups, float eps, float* scale, float* shift, const float* film,
                               int32_t film_stride, float* mean_rstd, void* stream);
/* The same launch additionally writes amax_out[b][g] = sqrt(max over the group's (tile, channel) partials of the sum of
 * squares) >= max |x| over the group's channels: the operand bound of ddnm_conv_desc::amax_in for the launches that read
 * the same tensor(s) RAW (a ResnetBlock's fused / un-fused nin_shortcut reads what norm1 normalises, models.py:109,
 * 115-134).  groups must equal DDNM_AMAX_N. */
int ddnm_gn_finalize_tiles_amax_f32(const float* part0, int32_t tiles_per_img0, int32_t C0, const float* part1,
                                    int32_t tiles_per_img1, int32_t C1, const float* gamma, const float* beta, int32_t B,
                                    int32_t HW, int32_t groups, float eps, float* scale, float* shift, const float* film,
                                    int32_t film_stride, float* mean_rstd, float* amax_out, void* stream);
/* Stand-alone form for raw operands that no GroupNorm reads first (Downsample / Upsample convolution inputs,
 * models.py:47-51,61-71; the attention output in front of proj_out, :183-189): out[b][i], i < DDNM_AMAX_N, bounds the
 * i-th 1/32 slice of image b's data of up to two sources.  kind 0: the fp32 tensor itself, per_image elements per image
 * (max |x|); kind 1: GroupNorm partials [tiles][C][2] of the tensor, per_image = tiles*C*2 floats (sqrt(max sum of
 * squares)).  per_image % 4 == 0; src1 may be NULL. */
int ddnm_amax_bound_f32(const float* src0, int64_t per_image0, int32_t kind0, const float* src1, int64_t per_image1,
                        int32_t kind1, float* out, int32_t B, void* stream);

/* ------------------------------------------------------------------------- *
 * Batched GEMM on MFMA f32:  C = alpha * A * op(B) + beta * D
 * A [M][K] (lda), op(B): transb=1 -> B stored [N][K] (ldb), transb=0 -> B stored [K][N];
 * batch index i -> (i / inner, i % inner); offset = outer*stride_o + inner*stride_i per operand.
 * Replaces torch.bmm / einsum in attention (models.py:171-185; unet.py:344-354) and the
 * separable A / A^+ products of SRConv (functions/svd_operators.py:853-859,893-900).
 * MFMA path: M, N multiples of 64, K multiple of 32, 16-byte aligned rows; any other shape runs a
 * scalar fallback kernel (tiny test sizes only).
 * ------------------------------------------------------------------------- */
typedef struct ddnm_gemm_desc {
    const float* A; const float* Bm; const float* D; float* C;
    int32_t M, N, K;
    int32_t lda, ldb, ldc, ldd;
    int32_t transb;
    int32_t batch, inner;                 /* batch = outer*inner */
    int64_t sAo, sAi, sBo, sBi, sCo, sCi, sDo, sDi;   /* element strides */
    float alpha, beta;
    int32_t transa;                       /* 1: A stored [K][M] (lda = pitch of that storage) */
    int32_t reserved;
} ddnm_gemm_desc;
int ddnm_bgemm_f32(const ddnm_gemm_desc* d, void* stream);

/* ------------------------------------------------------------------------- *
 * Input-gradient path of classifier guidance (guided_diffusion/diffusion.py:183-189 through
 * EncoderUNetModel, unet.py:684-895): only activation gradients are needed.
 * ------------------------------------------------------------------------- */
/* GroupNorm(+FiLM)(+SiLU) backward: dx = dL/dx for a = act(x*gn_scale + gn_shift); dA may sit behind a 2x2
 * average pool (dA_ups: [B][H/2][W/2][C], x0.25); `add` (same optional mapping) is summed in (skip branch). */
int ddnm_gn_bwd_f32(const float* x, const float* dA, int32_t dA_ups, const float* gn_scale, const float* gn_shift,
                    const float* mean_rstd, int32_t silu, const float* add, int32_t add_ups, int32_t B, int32_t H,
                    int32_t W, int32_t C, int32_t groups, double* partial /* [B][nchunk][groups][2] */, int32_t nchunk,
                    float* coef /* [B][groups][2] */, float* dx, void* stream);
int ddnm_gn_bwd_nchunk(int32_t HW, int32_t C);
/* ABI 6 -- the same for the fp16-activation classifier path (`classifier_use_fp16`, unet.py:817-823: the reference's
 * autograd then carries activations AND their gradients in fp16): x, dA, add, dx fp16 NHWC; arithmetic fp32, the two
 * group reductions fp64 in a fixed order. */
int ddnm_gn_bwd_h16(const void* x, const void* dA, int32_t dA_ups, const float* gn_scale, const float* gn_shift,
                    const float* mean_rstd, int32_t silu, const void* add, int32_t add_ups, int32_t B, int32_t H,
                    int32_t W, int32_t C, int32_t groups, double* partial /* [B][nchunk][groups][2] */, int32_t nchunk,
                    float* coef /* [B][groups][2] */, void* dx, void* stream);
/* in place on dP: dS = scale * P .* (dP - rowsum(dP .* P)) */
int ddnm_softmax_bwd_rows_f32(const float* P, float* dP, int64_t rows, int32_t n, int32_t ld, float scale, void* stream);
/* AttentionPool2d (unet.py:22-51): tokens, class-token attention (new qkv order), their backward, and the
 * gradient of log_softmax(logits)[y]. */
int ddnm_pool_tokens_f32(const float* h, const float* gn_scale, const float* gn_shift, const float* pos /* [C][HW+1] */,
                         float* X /* [B][HW+1][C] */, int32_t B, int32_t HW, int32_t C, void* stream);
int ddnm_pool_attn_fwd_f32(const float* qkv /* [B][T][3C] */, float* P /* [B][heads][T] */, float* a0 /* [B][C] */,
                           int32_t B, int32_t T, int32_t C, int32_t heads, void* stream);
int ddnm_pool_attn_bwd_f32(const float* qkv, const float* P, const float* da0, float* dqkv, int32_t B, int32_t T,
                           int32_t C, int32_t heads, void* stream);
int ddnm_pool_tokens_bwd_f32(const float* dX, float* dact /* [B][HW][C] */, int32_t B, int32_t HW, int32_t C, void* stream);
int ddnm_logsoftmax_grad_f32(const float* logits, const int64_t* y, float* dlogits, int32_t B, int32_t N, void* stream);
/* ABI 6 -- token construction / its backward over an fp16 NHWC activation (tokens and their gradient stay fp32) */
int ddnm_pool_tokens_h16(const void* h, const float* gn_scale, const float* gn_shift, const float* pos /* [C][HW+1] */,
                         float* X /* [B][HW+1][C] */, int32_t B, int32_t HW, int32_t C, void* stream);
int ddnm_pool_tokens_bwd_h16(const float* dX, void* dact /* fp16 [B][HW][C] */, int32_t B, int32_t HW, int32_t C, void* stream);

/* ABI 7 -- fused single-head self-attention of the celeba `Model`'s AttnBlock (guided_diffusion/models.py:171-185: the two
 * torch.bmm and the softmax between them): qkv = [B][T][3C] fp32 (q | k | v per token, the output of the fused
 * q / k / v 1x1 convolution), out = [B][T][C] fp32 = softmax_j(q_i . k_j * softmax_scale) v_j.  No [T][T] tensor in HBM.
 * Split-fp16 products (hi*hi' + hi*lo' + lo*hi', fp32 accumulate) like ddnm_conv3x3_s16_f32; s_qk / s_v = powers of two with
 * max|q|, max|k| <= 2^15 / s_qk and max|v| <= 2^15 / s_v (the host derives them once per checkpoint from a static bound of the
 * convolution's output; scores / outputs are multiplied by the inverse powers: exact).  T % 32 == 0, T <= 256, C % 128 == 0,
 * C <= 512. */
int ddnm_attn_fused_f32(const float* qkv, float* out, int32_t B, int32_t T, int32_t C, float s_qk, float s_v,
                        float softmax_scale, void* stream);
int ddnm_attn_fused_supported(int32_t T, int32_t C);

/* Row softmax in place: x[r][0..n) <- softmax(scale * x[r][:]); rows contiguous with ld. */
int ddnm_softmax_rows_f32(float* x, int64_t rows, int32_t n, int32_t ld, float scale, void* stream);

/* ------------------------------------------------------------------------- *
 * Small dense layers on the embedding path (models.py:216-222,306-308,92,121).
 *   y[b][n] = bias[n] + sum_k W[n][k] * act(x[b][k]),  act = swish if silu_in else id
 * ------------------------------------------------------------------------- */
int ddnm_linear_f32(const float* x, const float* W, const float* bias, float* y, int32_t B, int32_t K,
                    int32_t N, int32_t silu_in, void* stream);
/* Sinusoidal embedding: emb[b][i] = f(t[b]*freq[i]); order 0: [sin, cos] (models.py:6-24),
 * order 1: [cos, sin] (guided_diffusion/nn.py:103-121).  freq [half] is computed on the host. */
int ddnm_timestep_embedding_f32(const float* t, const float* freq, float* emb, int32_t B, int32_t half,
                                int32_t order, void* stream);

/* out = mean_2x2(act(in)) on NHWC, act = optional per-(b,c) affine (+swish): the two halves of a
 * `down=True` ResBlock (guided_diffusion/unet.py:133-140,237-242).  in is [B][2Ho][2Wo][C]. */
int ddnm_avgpool2_nhwc_f32(const float* in, const float* gn_scale, const float* gn_shift, int32_t silu, float* out,
                           int32_t B, int32_t Ho, int32_t Wo, int32_t C, void* stream);
/* emb[b][:] += table[idx[b]][:]  (class-label embedding, guided_diffusion/unet.py:651-653); idx is int64; `rows` =
 * rows of `table`: a label outside [0, rows) poisons its row with NaN instead of reading out of bounds. */
int ddnm_embedding_add_f32(float* emb, const float* table, const int64_t* idx, int32_t B, int32_t D, int32_t rows,
                           void* stream);

/* NCHW [B][C][H][W] -> NHWC [B][H][W][Cpad], channels >= C zero filled. */
int ddnm_nchw_to_nhwc_pad_f32(const float* src, float* dst, int32_t B, int32_t C, int32_t HW, int32_t Cpad,
                              void* stream);
/* im2col of the 3-channel network input for the 3x3 / pad 1 input convolution (models.py:225, unet.py:472-476):
 * dst[b][y][x][(ky*3+kx)*C + c] = src[b][c][y+ky-1][x+kx-1] (0 outside the image and for entries >= 9*C), so that
 * conv_in is ONE 32-wide K chunk (a 1x1 convolution over 27 real entries) instead of 9 taps x 32 padded channels. */
int ddnm_nchw_im2col3x3_pad_f32(const float* src, float* dst, int32_t B, int32_t C, int32_t H, int32_t W, int32_t Cpad,
                                void* stream);

/* ------------------------------------------------------------------------- *
 * DDNM sampler step (functions/svd_ddnm.py:57-65,74; guided_diffusion/diffusion.py:365-384).
 * Images are NCHW fp32 [B][3][H][W].  Scalars are the host-computed alpha-bar terms:
 *   x0   = (xt - et*sqrt(1-at)) / sqrt(at)                       (svd_ddnm.py:57)
 *   x0h  = x0 - lambda * A^+(A x0 - y)                           (svd_ddnm.py:59-61)
 *   xt'  = sqrt(at') * x0h + gamma*(c1*noise + c2*et)            (svd_ddnm.py:63-65)
 * `et` may have row stride et_bstride (6-channel learn_sigma heads: first 3 used, :54-55).
 * ------------------------------------------------------------------------- */
typedef struct ddnm_step_scalars {
    float sqrt_1m_at;     /* sqrt(1 - alpha_bar_t) */
    float sqrt_at;        /* sqrt(alpha_bar_t), applied as a true division */
    float sqrt_at_next;   /* sqrt(alpha_bar_{t'}) */
    float c1, c2;         /* noise / eps mixing coefficients (already multiplied by gamma) */
    float lambda;         /* 1 for DDNM (sigma_y = 0) */
    /* ABI 6 -- in-kernel noise: when a step entry point gets noise == NULL and rng_on != 0, the N(0, I) draw of
     * `torch.randn_like(x)` (svd_ddnm.py:65) happens inside the kernel: Philox4x32-10 + Box-Muller with the counter
     * (element / 4, rng_iter, rng_image_base + b, 0) and the key (rng_seed_lo, rng_seed_hi) -- no noise tensor is written
     * or read, and an image's noise depends on its GLOBAL index only (rank-count independent sharding).
     * ddnm_randn_philox_f32 produces the same values as a tensor. */
    uint32_t rng_on;
    uint32_t rng_seed_lo, rng_seed_hi;
    uint32_t rng_iter;        /* loop iteration k */
    uint32_t rng_image_base;  /* global index of image 0 of this launch */
    uint32_t reserved_rng;
} ddnm_step_scalars;


/* out [B][chw] fp32 = the Philox draw described above (chw % 4 == 0) */
int ddnm_randn_philox_f32(float* out, int32_t B, int64_t chw, uint32_t seed_lo, uint32_t seed_hi, uint32_t iter,
                          uint32_t image_base, void* stream);

/* x0 only (first half of every step; also feeds time travel). */
int ddnm_step_x0_f32(const float* xt, const float* et, int64_t et_bstride, float* x0, int32_t B, int64_t chw,
                     const ddnm_step_scalars* s, void* stream);
/* xt' = sqrt_at_next*(x0 - lambda*(proj - apy)) + c1*noise + c2*et
 * proj = A^+ A x0 and apy = A^+ y (constant over the run), or proj = A^+(A x0 - y) with apy = NULL. */
int ddnm_step_combine_f32(const float* x0, const float* proj, const float* apy, const float* noise,
                          const float* et, int64_t et_bstride, float* xt_next, int32_t B, int64_t chw,
                          const ddnm_step_scalars* s, void* stream);
/* Fused single-pass steps (x0 written too, for time travel): */
int ddnm_step_sr_avgpool_f32(const float* xt, const float* et, int64_t et_bstride, const float* noise,
                             const float* y /* [B][3][H/r][W/r] */, float* x0, float* xt_next, int32_t B,
                             int32_t H, int32_t W, int32_t r, const ddnm_step_scalars* s, void* stream);
/* w3_host: HOST pointer to the 3 per-pixel weights, NULL = (0.3333, 0.3334, 0.3333) of
 * svd_operators.py:632; the simplified path uses (1/3, 1/3, 1/3) (diffusion.py:33-42). */
int ddnm_step_color_f32(const float* xt, const float* et, int64_t et_bstride, const float* noise,
                        const float* y /* [B][H*W] */, float* x0, float* xt_next, int32_t B, int32_t HW,
                        const float* w3_host, const ddnm_step_scalars* s, void* stream);
int ddnm_step_inpaint_f32(const float* xt, const float* et, int64_t et_bstride, const float* noise,
                          const float* y /* [B][3*n_kept], HWC order of kept pixels */,
                          const int32_t* rank /* [HW]: index of pixel among kept ones, -1 if missing */,
                          int32_t n_kept, float* x0, float* xt_next, int32_t B, int32_t HW,
                          const ddnm_step_scalars* s, void* stream);
int ddnm_step_denoise_f32(const float* xt, const float* et, int64_t et_bstride, const float* noise,
                          const float* y, float* x0, float* xt_next, int32_t B, int64_t chw,
                          const ddnm_step_scalars* s, void* stream);
/* time-travel re-noise: xt' = a*x0 + b*noise  (svd_ddnm.py:74) */
int ddnm_renoise_f32(const float* x0, const float* noise, float* xt_next, int64_t n, float a, float b, void* stream);

/* ---- DDNM+ (sigma_y > 0, functions/svd_ddnm.py:80-164) building blocks --------------------------------- */
/* out = a*x + b*y (y may be NULL) */
int ddnm_axpby_f32(const float* x, const float* y, float* out, int64_t n, float a, float b, void* stream);
/* out[b][i] = a*x[b*x_bstride + i] + b*y[b*chw + i]  (guided eps: eps[:, :3] - sqrt(1-abar)*grad, svd_ddnm.py:51-52) */
int ddnm_axpby_strided_f32(const float* x, int64_t x_bstride, const float* y, float* out, int32_t B, int64_t chw,
                           float a, float b, void* stream);
/* out[i] = value for i < n (padding rows of token matrices; replaces tensor.zero_() / fill_() on the path). */
int ddnm_fill_f32(float* out, int64_t n, float value, void* stream);
/* out = (m ? cx_m : cx_n)*x + (m ? cy_m : cy_n)*y with m = mask[plane % planes_mask][p] != 0 (mask NULL: all measured):
 * Lambda / Lambda_noise of Inpainting (svd_operators.py:361-439), spectral weights of WalshHadamardCS (:253-320). */
int ddnm_mask_mix_f32(const float* x, const float* y, const float* mask, int32_t planes_mask, int64_t plane_elems,
                      float* out, int64_t total, float cx_m, float cx_n, float cy_m, float cy_n, void* stream);
/* Matrix-free SVD surface of the operators (A_functions.V / Vt / U / Ut / add_zeros / At / A_pinv_eta,
 * functions/svd_operators.py:9-97; not on the sampling hot path):
 *   gather_scale: out[b][i] = (idx[i] >= 0 ? in[b][idx[i]] : 0) * (scale ? scale[i] : 1); idx NULL = identity padded with
 *                 zeros up to n_out (`add_zeros`, and with `scale` the `singulars * temp[:, :n]` products);
 *   site_matmul:  out[b][s][i] = sum_j Mop[i][j] in[b][s][j] with Mop = M (trans 0) or M^T, n <= 16, element (b,s,j) at
 *                 b*sb + s*ss + j*sj in both tensors (the V_small / Vt_small products, :490-517,636-656); in != out. */
int ddnm_gather_scale_f32(const float* in, const int32_t* idx, const float* scale, float* out, int32_t B, int64_t n_in,
                          int64_t n_out, void* stream);
int ddnm_site_matmul_f32(const float* in, const float* M, float* out, int32_t B, int64_t sites, int32_t n, int64_t sb,
                         int64_t ss, int64_t sj, int32_t trans, void* stream);
/* Per-site spectral ops with the n x n orthogonal V (device, row-major) of a 1 x n measurement row:
 * mode 0 = r x r patches of [B*C][H][W] planes (SuperResolution, :535-623), mode 1 = RGB needles (Colorization, :669-736);
 * op 0: out = x + (c0-1) V[:,0] (V[:,0].x)  (Lambda, c0 = lambda of the measured direction);
 * op 1: out = V (d1 .* x_raw + d2 .* y_raw), d1 = (c0, c1, c1, ...), d2 = (c2, c3, c3, ...)  (Lambda_noise). */
int ddnm_site_spectral_f32(const float* x, const float* y, const float* V, int32_t n, int32_t mode, int32_t r,
                           int32_t B, int32_t C, int32_t H, int32_t W, float* out, int32_t op, float c0, float c1,
                           float c2, float c3, void* stream);

/* out = x .* table[plane % planes_table][p]: spectral gains of Deblurring / Deblurring2D (svd_operators.py:934-1165) */
int ddnm_mul_planes_f32(const float* x, const float* table, int32_t planes_table, int64_t plane_elems, float* out,
                        int64_t total, void* stream);

/* DDNM+ spectral weights of Deblurring (svd_operators.py:1016-1091) applied in the (V1^T . V1) spectral plane:
 * mode 0: out = x .* lambda(s[p]); mode 1: out = x .* d1(s[p]) + y .* d2(s[p]); s = un-thresholded s1_i*s1_j table
 * [plane_elems], p = i % plane_elems; lambda/d1/d2 as in svd_ddnm.py:121-131 (see ddnm_step.hip). */
int ddnm_spectral_mix_f32(const float* x, const float* y, const float* singulars, int64_t plane_elems, float* out,
                          int64_t total, float a, float sigma_y, float sigma_t, float eta, int32_t mode, void* stream);

/* Stand-alone operator kernels (A and A^+ of functions/svd_operators.py, direct form). */
int ddnm_op_avgpool_f32(const float* x, float* y, int32_t BC, int32_t H, int32_t W, int32_t r, void* stream);
int ddnm_op_upsample_f32(const float* y, float* x, int32_t BC, int32_t H, int32_t W, int32_t r, void* stream);
int ddnm_op_color_A_f32(const float* x, float* y, int32_t B, int32_t HW, const float* w3_host, void* stream);
int ddnm_op_color_pinv_f32(const float* y, float* x, int32_t B, int32_t HW, const float* w3_host, void* stream);
int ddnm_op_inpaint_A_f32(const float* x, const int32_t* rank, int32_t n_kept, float* y, int32_t B, int32_t HW,
                          void* stream);
int ddnm_op_inpaint_pinv_f32(const float* y, const int32_t* rank, int32_t n_kept, float* x, int32_t B,
                             int32_t HW, void* stream);
/* Orthonormal 2-D separable Walsh-Hadamard transform H_n (x) H_n of each [n][n] plane
 * (== the 1-D natural-order FWHT over n*n points of svd_operators.py:212-222), n in {32,...,256}.
 * mask (optional, [planes_mask][n*n], plane p uses mask[(p % planes_mask)]) multiplies the
 * spectrum between a forward and an inverse transform: out = H (mask .* (H in)). */
int ddnm_fwht2d_f32(const float* in, float* out, int32_t planes, int32_t n, void* stream);
int ddnm_fwht2d_masked_f32(const float* in, const float* mask, int32_t planes_mask, float* out,
                           int32_t planes, int32_t n, float* scratch /* planes*n*n */, void* stream);
/* gather/scatter between the permuted (k, c)-interleaved measurement vector and WH planes */
int ddnm_wh_gather_f32(const float* planes, const int32_t* perm, float* y, int32_t B, int32_t C, int32_t N,
                       int32_t n_keep, void* stream);
int ddnm_wh_scatter_f32(const float* y, const int32_t* perm, float* planes, int32_t B, int32_t C, int32_t N,
                        int32_t n_keep, void* stream);

/* Block-based CS, svd_operators.py:101-159: every ps x ps patch of every plane [planes][D][D] becomes one row of
 * patches[planes*(D/ps)^2][ps*ps] (row = plane, py, px; column = i*ps + j), inverse = 1 scatters the rows back.
 * Vt_small / V_small (:108-109) then act as one ddnm_bgemm_f32 over all patches.  D % ps == 0, ps % 4 == 0. */
int ddnm_patchify_f32(const float* src, float* dst, int32_t planes, int32_t D, int32_t ps, int32_t inverse, void* stream);

/* ------------------------------------------------------------------------- *
 * hq_demo sampler: DDPM posterior step with the DDNM core and the mask-shift tiles
 * (hq_demo/guided_diffusion/gaussian_diffusion.py:246-404,430-487,664-746).
 * ------------------------------------------------------------------------- */
/* x0 = clamp?(c_recip*x_t - c_recipm1*eps, -1, 1); eps row b at eps + b*eps_bstride (first 3 of the 6 output channels) */
int ddnm_hq_x0_f32(const float* xt, const float* eps, int64_t eps_bstride, float* x0, int32_t B, int64_t chw,
                   float c_recip, float c_recipm1, int32_t clip, void* stream);
/* x0_hat = lambda*A^+y + x0 - lambda*A^+A x0 (Eq. 17, :339) */
int ddnm_hq_project_f32(const float* x0, const float* apy, const float* apax0, float* x0_hat, int64_t n, float lam,
                        void* stream);
/* dst[p][dy+i][dx+j] = src[p][sy+i][sx+j], i < h, j < w, for `planes` planes of [Hs][Ws] / [Hd][Wd] */
int ddnm_copy_rect_f32(const float* src, int32_t Hs, int32_t Ws, int32_t sy, int32_t sx, float* dst, int32_t Hd, int32_t Wd,
                       int32_t dy, int32_t dx, int32_t planes, int32_t h, int32_t w, void* stream);
/* out = (coef1*x0_hat + coef2*x_t [+ gamma*grad]) + noise_scale*noise; grad may be NULL */
int ddnm_hq_sample_f32(const float* x0_hat, const float* xt, const float* grad, const float* noise, float* out, int64_t n,
                       float coef1, float coef2, float gamma, float noise_scale, void* stream);

/* inverse_data_transform + per-image MSE (datasets/__init__.py:218-227; diffusion.py:599-602):
 * img = clamp((x+1)/2, 0, 1); mse[b] = mean((img - clamp((x_orig+1)/2,0,1))^2) */
int ddnm_finalize_psnr_f32(const float* x, const float* x_orig, float* img /* may be NULL */, double* sse /* [B], zeroed by callee */,
                           int32_t B, int64_t chw, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DDNM_HIP_H */
