// HBM-bound helpers of the fp16-activation path (gfx950): everything between two convolutions of the ADM `use_fp16`
// torso that is not a convolution.  All tensors fp16 NHWC; arithmetic fp32 (the reference computes GroupNorm32 in
// fp32 and casts back, guided_diffusion/nn.py:17-19; SiLU on the cast tensor, nn.py:12-14).
//   * ddnm_gn_apply_h16     GroupNorm affine (+ FiLM, folded into scale/shift by gn_finalize) + swish, applied ONCE per
//                           element and written as the next convolution's operand; channel concat of two sources
//                           (torch.cat([h, hs.pop()], dim=1), unet.py:661) materialised on the way; optional fused
//                           2x2 average pooling (`down=True` ResBlocks: h_upd / x_upd = AvgPool2d, unet.py:237-242,
//                           113-140 with use_conv=False).
//   * ddnm_im2col3x3_h16    the 8x8 level's 3x3 convolutions as one GEMM (K = 9*Cin) -- tiles of 128+ pixels would
//                           span several images there.
//   * ddnm_nchw_to_nhwc_h16 the sampler's fp32 NCHW x_t -> zero-padded fp16 NHWC operand of the input convolution
//                           (`h = x.type(self.dtype)`, unet.py:655).
//   * ddnm_gn_stats_h16     stand-alone GroupNorm partials of an fp16 tensor (only where no producer epilogue could
//                           emit them).
// Thread = 8 consecutive channels (16 bytes) of one output pixel; grid-stride loops; no LDS.
#include "conv_common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ void act8(float (&v)[8], const float* __restrict__ sc, const float* __restrict__ sh, int silu) {
    const f32x4 s0 = *reinterpret_cast<const f32x4*>(sc), s1 = *reinterpret_cast<const f32x4*>(sc + 4);
    const f32x4 h0 = *reinterpret_cast<const f32x4*>(sh), h1 = *reinterpret_cast<const f32x4*>(sh + 4);
    const float s[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w};
    const float h[8] = {h0.x, h0.y, h0.z, h0.w, h1.x, h1.y, h1.z, h1.w};
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float t = v[e] * s[e] + h[e];
        v[e] = silu ? silu_f(t) : t;
    }
}

__device__ __forceinline__ void load8(const _Float16* p, float (&v)[8]) {
    const half8 h = *reinterpret_cast<const half8*>(p);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = (float)h[e];
}

__device__ __forceinline__ void store8(_Float16* p, const float (&v)[8]) {
    half8 h;
#pragma unroll
    for (int e = 0; e < 8; ++e) h[e] = (_Float16)v[e];
    *reinterpret_cast<half8*>(p) = h;
}

__global__ __launch_bounds__(256) void gn_apply_h16_kernel(const _Float16* __restrict__ src0, const _Float16* __restrict__ src1,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           _Float16* __restrict__ out, int H, int W, int C0, int C1, int silu,
                                                           int pool, size_t total8) {
    const int C = C0 + C1, C8 = C >> 3;
    const int Ho = pool ? H >> 1 : H, Wo = pool ? W >> 1 : W;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total8; i += (size_t)gridDim.x * 256) {
        const size_t opix = i / C8;
        const int c = (int)(i - opix * C8) * 8;
        const int b = (int)(opix / ((size_t)Ho * Wo));
        const _Float16* s;
        int cs;
        if (c < C0) { s = src0 + c; cs = C0; } else { s = src1 + (c - C0); cs = C1; }
        float v[8];
        if (!pool) {
            load8(s + opix * cs, v);
            if (scale) act8(v, scale + (size_t)b * C + c, shift + (size_t)b * C + c, silu);
        } else {
            const int r = (int)(opix - (size_t)b * Ho * Wo);
            const int oy = r / Wo, ox = r - oy * Wo;
            const size_t p00 = ((size_t)b * H + 2 * oy) * W + 2 * ox;
            float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                load8(s + (p00 + (q >> 1) * W + (q & 1)) * cs, v);
                if (scale) act8(v, scale + (size_t)b * C + c, shift + (size_t)b * C + c, silu);
#pragma unroll
                for (int e = 0; e < 8; ++e) a[e] += v[e];
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = 0.25f * a[e];
        }
        store8(out + i * 8, v);
    }
}

extern "C" int ddnm_gn_apply_h16(const void* src0, const void* src1, const float* scale, const float* shift, void* out,
                                 int32_t B, int32_t H, int32_t W, int32_t C0, int32_t C1, int32_t silu, int32_t pool,
                                 void* stream) {
    if (!src0 || !out || B <= 0 || H <= 0 || W <= 0 || C0 <= 0 || C1 < 0) return DDNM_E_BADARG;
    if ((scale == nullptr) != (shift == nullptr)) return DDNM_E_BADARG;
    if ((C0 | C1) & 7) return DDNM_E_SHAPE;
    if (C1 > 0 && !src1) return DDNM_E_BADARG;
    if (pool && ((H | W) & 1)) return DDNM_E_SHAPE;
    const size_t opix = pool ? (size_t)B * (H / 2) * (W / 2) : (size_t)B * H * W;
    const size_t total8 = opix * (C0 + C1) / 8;
    const size_t blocks = (total8 + 255) / 256;
    const unsigned g = (unsigned)(blocks < 32768 ? blocks : 32768);
    DDNM_LAUNCH(gn_apply_h16_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const _Float16*>(src0),
                reinterpret_cast<const _Float16*>(src1), scale, shift, reinterpret_cast<_Float16*>(out), H, W, C0, C1, silu,
                pool, total8);
    return 0;
}

// col[(b*HW + p)][tap*C + c] = act(concat_c(src0, src1))[b, y+ky-1, x+kx-1, c]   (0 outside the image)
__global__ __launch_bounds__(256) void im2col3x3_h16_kernel(const _Float16* __restrict__ src0, const _Float16* __restrict__ src1,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            _Float16* __restrict__ out, int H, int W, int C0, int C1,
                                                            int silu, size_t total8) {
    const int C = C0 + C1, C8 = C >> 3, HW = H * W;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total8; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C8) * 8;
        size_t t = i / C8;
        const int tap = (int)(t % 9);
        t /= 9;
        const int p = (int)(t % HW), b = (int)(t / HW);
        const int y = p / W + tap / 3 - 1, x = p % W + tap % 3 - 1;
        float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
        if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) {
            const size_t pix = (size_t)b * HW + (size_t)y * W + x;
            load8(c < C0 ? src0 + pix * C0 + c : src1 + pix * C1 + (c - C0), v);
            if (scale) act8(v, scale + (size_t)b * C + c, shift + (size_t)b * C + c, silu);
        }
        store8(out + i * 8, v);
    }
}

extern "C" int ddnm_im2col3x3_h16(const void* src0, const void* src1, const float* scale, const float* shift, void* out,
                                  int32_t B, int32_t H, int32_t W, int32_t C0, int32_t C1, int32_t silu, void* stream) {
    if (!src0 || !out || B <= 0 || H <= 0 || W <= 0 || C0 <= 0 || C1 < 0) return DDNM_E_BADARG;
    if ((scale == nullptr) != (shift == nullptr)) return DDNM_E_BADARG;
    if ((C0 | C1) & 7) return DDNM_E_SHAPE;
    if (C1 > 0 && !src1) return DDNM_E_BADARG;
    const size_t total8 = (size_t)B * H * W * 9 * (C0 + C1) / 8;
    const size_t blocks = (total8 + 255) / 256;
    const unsigned g = (unsigned)(blocks < 32768 ? blocks : 32768);
    DDNM_LAUNCH(im2col3x3_h16_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const _Float16*>(src0),
                reinterpret_cast<const _Float16*>(src1), scale, shift, reinterpret_cast<_Float16*>(out), H, W, C0, C1, silu,
                total8);
    return 0;
}

// x fp32 NCHW [B][C][HW] -> fp16 NHWC [B][HW][cpad], channels >= C zero.  Thread = one pixel, 8 output channels.
__global__ __launch_bounds__(256) void nchw_to_nhwc_h16_kernel(const float* __restrict__ x, _Float16* __restrict__ out, int C,
                                                               int HW, int cpad, size_t total8) {
    const int P8 = cpad >> 3;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total8; i += (size_t)gridDim.x * 256) {
        // consecutive threads walk consecutive PIXELS of one 8-channel group: coalesced NCHW reads
        const size_t bp = i % ((size_t)HW), rest = i / (size_t)HW;
        const int g8 = (int)(rest % P8);
        const size_t b = rest / P8;
        float v[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int c = g8 * 8 + e;
            v[e] = c < C ? x[(b * C + c) * HW + bp] : 0.f;
        }
        store8(out + (b * HW + bp) * cpad + g8 * 8, v);
    }
}

extern "C" int ddnm_nchw_to_nhwc_h16(const float* x, void* out, int32_t B, int32_t C, int32_t HW, int32_t cpad, void* stream) {
    if (!x || !out || B <= 0 || C <= 0 || HW <= 0 || cpad < C || (cpad & 7)) return DDNM_E_BADARG;
    const size_t total8 = (size_t)B * HW * (cpad / 8);
    const size_t blocks = (total8 + 255) / 256;
    const unsigned g = (unsigned)(blocks < 32768 ? blocks : 32768);
    DDNM_LAUNCH(nchw_to_nhwc_h16_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, x, reinterpret_cast<_Float16*>(out), C, HW,
                cpad, total8);
    return 0;
}

// Stand-alone GroupNorm partials of an fp16 NHWC tensor in the layout of the convolution epilogues:
// stats[((b*tiles + t)*C + c)*2 + {0,1}] = sum / sum of squares over the pixels of tile t (HW/tiles pixels each).
// grid (B*tiles, ceil(C/2048)): a thread owns 8 channels and walks its tile's pixels in a fixed order.
__global__ __launch_bounds__(256) void gn_stats_h16_kernel(const _Float16* __restrict__ src, float* __restrict__ stats, int HW,
                                                           int C, int tiles) {
    const int b = blockIdx.x / tiles, t = blockIdx.x - b * tiles;
    const int c = (blockIdx.y * 256 + threadIdx.x) * 8;
    if (c >= C) return;
    const int P = HW / tiles;
    const _Float16* s = src + ((size_t)b * HW + (size_t)t * P) * C + c;
    float a[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, q[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    for (int p = 0; p < P; ++p) {
        float v[8];
        load8(s + (size_t)p * C, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) { a[e] += v[e]; q[e] += v[e] * v[e]; }
    }
    float* o = stats + ((size_t)blockIdx.x * C + c) * 2;
#pragma unroll
    for (int e = 0; e < 8; ++e) { o[2 * e] = a[e]; o[2 * e + 1] = q[e]; }
}

extern "C" int ddnm_gn_stats_h16(const void* src, float* stats, int32_t B, int32_t HW, int32_t C, int32_t tiles, void* stream) {
    if (!src || !stats || B <= 0 || HW <= 0 || C <= 0 || tiles <= 0) return DDNM_E_BADARG;
    if ((C & 7) || HW % tiles) return DDNM_E_SHAPE;
    DDNM_LAUNCH(gn_stats_h16_kernel, dim3(B * tiles, (C / 8 + 255) / 256), dim3(256), 0, (hipStream_t)stream,
                reinterpret_cast<const _Float16*>(src), stats, HW, C, tiles);
    return 0;
}
