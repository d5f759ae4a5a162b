// Small kernels of the embedding path and layout plumbing, gfx950.
#include "common.h"

// y[b][n] = bias[n] + sum_k W[n][k] * act(x[b][k]); one wave per output, coalesced over k.
// Replaces nn.Linear on the timestep embedding (guided_diffusion/models.py:216-222,306-308)
// and every ResnetBlock.temb_proj (models.py:92,121) -- the latter concatenated into ONE
// launch per step since they all read the same swish(temb).
__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                     const float* __restrict__ bias, float* __restrict__ y, int K,
                                                     int N, int silu_in) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;
    if (n >= N) return;
    const float* xr = x + (size_t)b * K;
    const float* wr = W + (size_t)n * K;
    float acc = 0.f;
    for (int k = lane; k < K; k += 64) {
        float v = xr[k];
        if (silu_in) v = silu_f(v);
        acc += wr[k] * v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) y[(size_t)b * N + n] = acc + (bias ? bias[n] : 0.f);
}

// Row-streaming form for the wide FiLM projection (all ResBlock emb_layers of the ADM UNet in one launch: N = 51712
// rows of K = 1024 fp32 = 212 MB per forward, the x rows are a few KB).  A wave owns whole W rows and reads each ONCE
// with 16-byte loads (next row in flight while the current one is multiplied); BT (4 or 8) x rows sit in registers in the
// same lane layout (k = 256 i + 4 lane ...), so the only other traffic is one xor-reduction per (row, batch row).
// Each (b, n) dot product is summed in an order that does not depend on B or on the launch geometry.
template <int KV, int BT>           // K = 256 * KV
__global__ __launch_bounds__(256) void linear_rows_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                          const float* __restrict__ bias, float* __restrict__ y, int B,
                                                          int N, int silu_in) {
    constexpr int K = 256 * KV;
    const int lane = threadIdx.x & 63;
    const int wave = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    for (int b0 = 0; b0 < B; b0 += BT) {
        f32x4 xv[BT][KV];
#pragma unroll
        for (int b = 0; b < BT; ++b)
#pragma unroll
            for (int i = 0; i < KV; ++i) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (b0 + b < B) v = *reinterpret_cast<const f32x4*>(x + (size_t)(b0 + b) * K + i * 256 + lane * 4);
                if (silu_in) v = f32x4{silu_f(v.x), silu_f(v.y), silu_f(v.z), silu_f(v.w)};
                xv[b][i] = v;
            }
        f32x4 wv[KV], wn[KV];
        int n = wave;
        if (n < N) {
#pragma unroll
            for (int i = 0; i < KV; ++i) wv[i] = *reinterpret_cast<const f32x4*>(W + (size_t)n * K + i * 256 + lane * 4);
        }
        for (; n < N; n += nwaves) {
            const int nn = n + nwaves;
            if (nn < N) {
#pragma unroll
                for (int i = 0; i < KV; ++i) wn[i] = *reinterpret_cast<const f32x4*>(W + (size_t)nn * K + i * 256 + lane * 4);
            }
            float acc[BT];
#pragma unroll
            for (int b = 0; b < BT; ++b) {
                float a = 0.f;
#pragma unroll
                for (int i = 0; i < KV; ++i) {
                    a += wv[i].x * xv[b][i].x;
                    a += wv[i].y * xv[b][i].y;
                    a += wv[i].z * xv[b][i].z;
                    a += wv[i].w * xv[b][i].w;
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) a += __shfl_xor(a, o);
                acc[b] = a;
            }
            if (lane < BT && b0 + lane < B) {
                float v = acc[0];
#pragma unroll
                for (int b = 1; b < BT; ++b) v = lane == b ? acc[b] : v;
                y[(size_t)(b0 + lane) * N + n] = v + (bias ? bias[n] : 0.f);
            }
#pragma unroll
            for (int i = 0; i < KV; ++i) wv[i] = wn[i];
        }
    }
}

extern "C" int ddnm_linear_f32(const float* x, const float* W, const float* bias, float* y, int32_t B, int32_t K,
                               int32_t N, int32_t silu_in, void* stream) {
    if (!x || !W || !y || B <= 0 || K <= 0 || N <= 0) return DDNM_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    if (N >= 4096 && (K == 512 || K == 1024) && (((uintptr_t)x | (uintptr_t)W) & 15) == 0) {
        const int grid = 2048;                  // 8 workgroups (32 waves) per CU: >= 128 KB of W rows in flight per CU
        if (K == 1024 && B <= 4) { DDNM_LAUNCH((linear_rows_kernel<4, 4>), dim3(grid), dim3(256), 0, s, x, W, bias, y, B, N, silu_in); }
        else if (K == 1024) { DDNM_LAUNCH((linear_rows_kernel<4, 8>), dim3(grid), dim3(256), 0, s, x, W, bias, y, B, N, silu_in); }
        else if (B <= 4) { DDNM_LAUNCH((linear_rows_kernel<2, 4>), dim3(grid), dim3(256), 0, s, x, W, bias, y, B, N, silu_in); }
        else { DDNM_LAUNCH((linear_rows_kernel<2, 8>), dim3(grid), dim3(256), 0, s, x, W, bias, y, B, N, silu_in); }
        return 0;
    }
    DDNM_LAUNCH(linear_kernel, dim3((N + 3) / 4, B), dim3(256), 0, s, x, W, bias, y, K, N, silu_in);
    return 0;
}

__global__ void temb_kernel(const float* __restrict__ t, const float* __restrict__ freq, float* __restrict__ emb,
                            int half, int order) {
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < half; i += blockDim.x) {
        const float a = t[b] * freq[i];
        const float s = sinf(a), c = cosf(a);
        float* e = emb + (size_t)b * 2 * half;
        if (order == 0) { e[i] = s; e[half + i] = c; }
        else { e[i] = c; e[half + i] = s; }
    }
}

extern "C" int ddnm_timestep_embedding_f32(const float* t, const float* freq, float* emb, int32_t B, int32_t half,
                                           int32_t order, void* stream) {
    if (!t || !freq || !emb || B <= 0 || half <= 0) return DDNM_E_BADARG;
    DDNM_LAUNCH(temb_kernel, dim3(B), dim3(128), 0, (hipStream_t)stream, t, freq, emb, half, order);
    return 0;
}

// NCHW [B][C][HW] -> NHWC [B][HW][Cpad]; float4 stores, channels >= C are zero.
__global__ __launch_bounds__(256) void nchw_to_nhwc_pad_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                               int C, int HW, int Cpad, size_t total4) {
    const int q = Cpad >> 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        const int c4 = (int)(i % q);
        const size_t pix = i / q;                // b*HW + p
        const size_t b = pix / HW, p = pix - b * HW;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const int c = c4 * 4;
        const float* s = src + (b * C) * HW + p;
        if (c + 0 < C) v.x = s[(size_t)(c + 0) * HW];
        if (c + 1 < C) v.y = s[(size_t)(c + 1) * HW];
        if (c + 2 < C) v.z = s[(size_t)(c + 2) * HW];
        if (c + 3 < C) v.w = s[(size_t)(c + 3) * HW];
        reinterpret_cast<f32x4*>(dst)[i] = v;
    }
}

extern "C" int ddnm_nchw_to_nhwc_pad_f32(const float* src, float* dst, int32_t B, int32_t C, int32_t HW, int32_t Cpad,
                                         void* stream) {
    if (!src || !dst || B <= 0 || C <= 0 || HW <= 0 || Cpad < C || (Cpad & 3)) return DDNM_E_BADARG;
    const size_t total4 = (size_t)B * HW * (Cpad / 4);
    const unsigned grid = (unsigned)((total4 + 255) / 256 < 4096 ? (total4 + 255) / 256 : 4096);
    DDNM_LAUNCH(nchw_to_nhwc_pad_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, dst, C, HW, Cpad,
                       total4);
    return 0;
}

// im2col of the network's 3-channel input for the 3x3 / pad 1 input convolution (models.py:225, unet.py:472-476):
// NCHW [B][C][H][W] -> NHWC [B][H][W][Cpad] with entry k = (ky*3 + kx)*C + c = x[b][c][y+ky-1][x+kx-1] (0 outside,
// 0 for k >= 9*C).  With C = 3 the 27 taps fit ONE 32-channel K chunk, so conv_in runs as a 1x1 convolution with
// K = 32 instead of 9 taps x 32 zero-padded channels (K = 288, 10.7x the MFMA work: 350 us -> HBM-bound).
__global__ __launch_bounds__(256) void nchw_im2col3x3_pad_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                                 int C, int H, int W, int Cpad, size_t total4) {
    const int q = Cpad >> 2, HW = H * W;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        const int k4 = (int)(i % q);
        const size_t pix = i / q;
        const size_t b = pix / HW;
        const int p = (int)(pix - b * HW);
        const int y = p / W, x = p - y * W;
        float v[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const int k = k4 * 4 + e;
            const int tap = k / C, c = k - tap * C;
            const int yy = y + tap / 3 - 1, xx = x + tap % 3 - 1;
            v[e] = (tap < 9 && (unsigned)yy < (unsigned)H && (unsigned)xx < (unsigned)W)
                       ? src[(b * C + c) * HW + (size_t)yy * W + xx] : 0.f;
        }
        reinterpret_cast<f32x4*>(dst)[i] = f32x4{v[0], v[1], v[2], v[3]};
    }
}

extern "C" int ddnm_nchw_im2col3x3_pad_f32(const float* src, float* dst, int32_t B, int32_t C, int32_t H, int32_t W,
                                           int32_t Cpad, void* stream) {
    if (!src || !dst || B <= 0 || C <= 0 || H <= 0 || W <= 0 || Cpad < 9 * C || (Cpad & 3)) return DDNM_E_BADARG;
    const size_t total4 = (size_t)B * H * W * (Cpad / 4);
    const unsigned grid = (unsigned)((total4 + 255) / 256 < 8192 ? (total4 + 255) / 256 : 8192);
    DDNM_LAUNCH(nchw_im2col3x3_pad_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, dst, C, H, W, Cpad, total4);
    return 0;
}

// 2x2 average pooling of an NHWC tensor with an optional per-(sample, channel) affine + swish applied
// to every input element first: out = mean_2x2(act(in)).  Replaces the `down=True` ResBlock halves
// h = AvgPool2d(SiLU(GroupNorm(x))) and x = AvgPool2d(x)  (guided_diffusion/unet.py:133-140,237-242).
__global__ __launch_bounds__(256) void avgpool2_nhwc_kernel(const float* __restrict__ in, const float* __restrict__ sc,
                                                            const float* __restrict__ sh, int silu,
                                                            float* __restrict__ out, int Ho, int Wo, int C4,
                                                            size_t total4) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        const int c4 = (int)(i % C4);
        size_t r = i / C4;
        const int ox = (int)(r % Wo);
        r /= Wo;
        const int oy = (int)(r % Ho);
        const size_t b = r / Ho;
        const f32x4* src = reinterpret_cast<const f32x4*>(in) + ((b * 2 * Ho + 2 * oy) * (size_t)(2 * Wo) + 2 * ox) * C4 + c4;
        f32x4 v[4] = {src[0], src[C4], src[(size_t)2 * Wo * C4], src[(size_t)2 * Wo * C4 + C4]};
        if (sc) {
            const f32x4 a = reinterpret_cast<const f32x4*>(sc)[b * C4 + c4], t = reinterpret_cast<const f32x4*>(sh)[b * C4 + c4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                v[k] = v[k] * a + t;
                if (silu) { v[k].x = silu_f(v[k].x); v[k].y = silu_f(v[k].y); v[k].z = silu_f(v[k].z); v[k].w = silu_f(v[k].w); }
            }
        }
        reinterpret_cast<f32x4*>(out)[i] = ((v[0] + v[1]) + (v[2] + v[3])) * 0.25f;
    }
}

extern "C" int ddnm_avgpool2_nhwc_f32(const float* in, const float* gn_scale, const float* gn_shift, int32_t silu,
                                      float* out, int32_t B, int32_t Ho, int32_t Wo, int32_t C, void* stream) {
    if (!in || !out || B <= 0 || Ho <= 0 || Wo <= 0 || C <= 0 || (gn_scale && !gn_shift)) return DDNM_E_BADARG;
    if (C & 3) return DDNM_E_SHAPE;
    const size_t total4 = (size_t)B * Ho * Wo * (C / 4);
    const unsigned grid = (unsigned)((total4 + 255) / 256 < 8192 ? (total4 + 255) / 256 : 8192);
    DDNM_LAUNCH(avgpool2_nhwc_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, in, gn_scale, gn_shift, silu, out,
                Ho, Wo, C / 4, total4);
    return 0;
}

// emb[b][:] += table[idx[b]][:]   (nn.Embedding lookup of the class label, guided_diffusion/unet.py:651-653)
// A label outside [0, rows) never reads the table: the row is poisoned with NaN instead (torch.nn.Embedding would
// raise; a kernel cannot, and a host-side check would cost a device->host sync per forward) -- loud, not silent.
__global__ void embedding_add_kernel(float* __restrict__ emb, const float* __restrict__ table,
                                     const int64_t* __restrict__ idx, int D, int rows) {
    const int b = blockIdx.x;
    const int64_t r = idx[b];
    const bool ok = r >= 0 && r < rows;
    const float* row = table + (size_t)(ok ? r : 0) * D;
    for (int i = threadIdx.x; i < D; i += blockDim.x)
        emb[(size_t)b * D + i] = ok ? emb[(size_t)b * D + i] + row[i] : __builtin_nanf("");
}

extern "C" int ddnm_embedding_add_f32(float* emb, const float* table, const int64_t* idx, int32_t B, int32_t D,
                                      int32_t rows, void* stream) {
    if (!emb || !table || !idx || B <= 0 || D <= 0 || rows <= 0) return DDNM_E_BADARG;
    DDNM_LAUNCH(embedding_add_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, emb, table, idx, D, rows);
    return 0;
}

// ABI version: bumped whenever a descriptor struct or a prototype of include/ddnm_hip.h changes
// (2: ddnm_conv16_desc and the fp16-activation entry points, ddnm_build_digest, ddnm_sizeof; 3: ddnm_conv16_desc::fin_*;
// 4: ddnm_conv_desc::acc_scale and the split-fp16 entry points ddnm_conv3x3_s16_*; 5: ddnm_conv_desc::amax_in;
// 6: the fp16-activation classifier entry points ddnm_gn_bwd_h16, ddnm_pool_tokens*_h16, ddnm_attn16_d64_lse / _bwd;
// 7: ddnm_conv_desc::flags (was reserved0), ddnm_step_srconv_* removed, no environment variable is read by the library).
extern "C" int ddnm_version(void) { return 7; }

// sha256 of the sources + flags this binary was compiled from (ddnm_amd/build.py passes it with -D); the loader
// (ddnm_amd/_lib.py) compares it with the digest of the sources next to it and refuses a stale binary.
#ifndef DDNM_BUILD_DIGEST
#define DDNM_BUILD_DIGEST "unstamped"
#endif
extern "C" const char* ddnm_build_digest(void) { return DDNM_BUILD_DIGEST; }

// sizeof() of the descriptor structs as this binary sees them: 0 conv_desc, 1 gemm_desc, 2 conv16_desc, 3 step_scalars
extern "C" int ddnm_sizeof(int which) {
    switch (which) {
        case 0: return (int)sizeof(ddnm_conv_desc);
        case 1: return (int)sizeof(ddnm_gemm_desc);
        case 2: return (int)sizeof(ddnm_conv16_desc);
        case 3: return (int)sizeof(ddnm_step_scalars);
        default: return -1;
    }
}

extern "C" const char* ddnm_error_string(int code) {
    if (code == 0) return "success";
    if (code == DDNM_E_BADARG) return "ddnm: bad argument (null pointer, non-positive size or misalignment)";
    if (code == DDNM_E_SHAPE) return "ddnm: shape not supported by this kernel family";
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "ddnm: unknown error";
}
