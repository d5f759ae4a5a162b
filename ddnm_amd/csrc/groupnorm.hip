// GroupNorm statistics for NHWC activations, gfx950.  HBM-bound: one coalesced read of the
// tensor (float4 per lane along C), fp32 partial sums over <= 64 pixels per thread, then
// fp64 combination in a fixed order (deterministic, no atomics).  The normalisation itself
// is never materialised: `finalize` emits a per-(sample, channel) affine that the consumer
// convolution applies while it stages its A tile (conv_igemm_f32.hip).
// Replaces torch.nn.GroupNorm(32, C, eps) -- guided_diffusion/models.py:32-33,
// guided_diffusion/nn.py:17-19,93-100.
#include "conv_common.h"

// max that PROPAGATES a non-finite operand as +inf (fmaxf drops NaN): the operand bounds of the split-fp16 kernels must
// not underestimate when a partial is NaN
__device__ __forceinline__ float nan_max(float q, float v) { return (v == v) ? fmaxf(q, v) : INFINITY; }

constexpr int GN_PIX_PER_THREAD = 64;

static inline int gn_block_dim(int C4) { return C4 <= 256 ? 256 : (C4 <= 512 ? 512 : 1024); }

extern "C" int ddnm_gn_nchunk(int32_t HW, int32_t C) {
    const int C4 = C / 4;
    if (C4 <= 0 || C4 > 1024) return DDNM_E_SHAPE;
    const int rows = gn_block_dim(C4) / C4;
    const int pix = rows * GN_PIX_PER_THREAD;
    return (HW + pix - 1) / pix;
}

__global__ void gn_stats_kernel(const float* __restrict__ src0, const float* __restrict__ src1, int HW, int C0,
                                int C1, int groups, double* __restrict__ partial, int nchunk, int pix_per_chunk) {
    // per-thread, per-channel fp32 sums over <= 64 pixels -> LDS [2][blockDim][4]; then one thread per group adds
    // its channels in fp64 (any group size: ADM nets have 1 .. 32 channels per group)
    extern __shared__ __attribute__((aligned(16))) float redf[];
    const int C = C0 + C1, C4 = C >> 2;
    const int rows = blockDim.x / C4, active = rows * C4;
    const int tid = threadIdx.x, chunk = blockIdx.x, b = blockIdx.y;
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, ss = {0.f, 0.f, 0.f, 0.f};
    if (tid < active) {
        const int c4 = tid % C4, prow = tid / C4;
        const int c = c4 * 4;
        const float* base;
        int cs;
        if (c < C0) { base = src0 + (size_t)b * HW * C0 + c; cs = C0; }
        else { base = src1 + (size_t)b * HW * C1 + (c - C0); cs = C1; }
        const int p_end = min(HW, (chunk + 1) * pix_per_chunk);
        for (int p = chunk * pix_per_chunk + prow; p < p_end; p += rows) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(base + (size_t)p * cs);
            s += v;
            ss += v * v;
        }
    }
    f32x4* r4 = reinterpret_cast<f32x4*>(redf);
    r4[tid] = s;
    r4[blockDim.x + tid] = ss;
    __syncthreads();
    if (tid < groups) {
        const int cpg = C / groups;
        const float* sum = redf;
        const float* sq = redf + 4 * blockDim.x;
        double a = 0.0, a2 = 0.0;
        for (int r = 0; r < rows; ++r)
            for (int j = 0; j < cpg; ++j) {
                const int t = r * C + tid * cpg + j;        // row r holds channels [0, C) as C4 float4s
                a += (double)sum[t];
                a2 += (double)sq[t];
            }
        double* o = partial + (((size_t)b * nchunk + chunk) * groups + tid) * 2;
        o[0] = a;
        o[1] = a2;
    }
}

extern "C" int ddnm_gn_stats_f32(const float* src0, const float* src1, int32_t B, int32_t HW, int32_t C0, int32_t C1,
                                 int32_t groups, double* partial, int32_t nchunk, void* stream) {
    if (!src0 || !partial || B <= 0 || HW <= 0 || C0 <= 0 || (C1 > 0 && !src1)) return DDNM_E_BADARG;
    const int C = C0 + C1;
    if (groups <= 0 || groups > 64 || C % groups || C % 4 || C0 % 4 || C / 4 > 1024) return DDNM_E_SHAPE;
    const int C4 = C / 4, bd = gn_block_dim(C4);
    const int rows = bd / C4, pix = rows * GN_PIX_PER_THREAD;
    if (nchunk != (HW + pix - 1) / pix) return DDNM_E_BADARG;
    DDNM_LAUNCH(gn_stats_kernel, dim3(nchunk, B), dim3(bd), 2 * bd * 4 * sizeof(float), (hipStream_t)stream, src0,
                       src1, HW, C0, C1, groups, partial, nchunk, pix);
    return 0;
}

__global__ __launch_bounds__(256) void gn_finalize_kernel(const double* __restrict__ partial, int nchunk,
                                                          const float* __restrict__ gamma,
                                                          const float* __restrict__ beta, int HW, int C, int groups,
                                                          float eps, float* __restrict__ scale,
                                                          float* __restrict__ shift,
                                                          const float* __restrict__ film, int film_stride,
                                                          float* __restrict__ mean_rstd) {
    __shared__ double acc[64][4][2];
    __shared__ float mean_s[64], rstd_s[64];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int g = tid >> 2, sub = tid & 3;            // 64 groups x 4 partial lanes
    if (g < groups) {
        double a = 0.0, a2 = 0.0;
        for (int ch = sub; ch < nchunk; ch += 4) {
            const double* p = partial + (((size_t)b * nchunk + ch) * groups + g) * 2;
            a += p[0];
            a2 += p[1];
        }
        acc[g][sub][0] = a;
        acc[g][sub][1] = a2;
    }
    __syncthreads();
    if (g < groups && sub == 0) {
        const double a = (acc[g][0][0] + acc[g][1][0]) + (acc[g][2][0] + acc[g][3][0]);
        const double a2 = (acc[g][0][1] + acc[g][1][1]) + (acc[g][2][1] + acc[g][3][1]);
        const double cnt = (double)HW * (double)(C / groups);
        const double mean = a / cnt;
        double var = a2 / cnt - mean * mean;
        var = var > 0.0 ? var : 0.0;
        mean_s[g] = (float)mean;
        rstd_s[g] = (float)(1.0 / sqrt(var + (double)eps));
        if (mean_rstd) {                      // kept for the backward pass (classifier input gradient)
            mean_rstd[((size_t)b * groups + g) * 2 + 0] = mean_s[g];
            mean_rstd[((size_t)b * groups + g) * 2 + 1] = rstd_s[g];
        }
    }
    __syncthreads();
    const int cpg = C / groups;
    for (int c = tid; c < C; c += 256) {
        const int gg = c / cpg;
        float sc = rstd_s[gg] * gamma[c];
        float sh = beta[c] - mean_s[gg] * sc;
        if (film) {   // FiLM: GN(x)*(1+s)+t  (guided_diffusion/unet.py:248-251), row = [s(0..C) | t(0..C)]
            const float s1 = 1.0f + film[(size_t)b * film_stride + c];
            sc = sc * s1;
            sh = sh * s1 + film[(size_t)b * film_stride + C + c];
        }
        scale[(size_t)b * C + c] = sc;
        shift[(size_t)b * C + c] = sh;
    }
}

extern "C" int ddnm_gn_finalize_f32(const double* partial, int32_t nchunk, const float* gamma, const float* beta,
                                    int32_t B, int32_t HW, int32_t C, int32_t groups, float eps, float* scale,
                                    float* shift, const float* film, int32_t film_stride, float* mean_rstd,
                                    void* stream) {
    if (!partial || !gamma || !beta || !scale || !shift || B <= 0 || nchunk <= 0) return DDNM_E_BADARG;
    if (groups <= 0 || groups > 64 || C % groups) return DDNM_E_SHAPE;
    DDNM_LAUNCH(gn_finalize_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, partial, nchunk, gamma, beta,
                       HW, C, groups, eps, scale, shift, film, film_stride, mean_rstd);
    return 0;
}

// ---- finalize from the per-(M tile, channel) partials emitted by the producing convolution's epilogue
// (conv_igemm_f32.hip).  The input may be the channel concat of two tensors with different tilings.
// grid (B, groups): one workgroup per (sample, group); 256 threads stride over the group's (tile, channel)
// partials, fp64 sums combined in a fixed order.
__global__ __launch_bounds__(256) void gn_finalize_tiles_kernel(const float* __restrict__ part0, int tpi0, int C0,
                                                                const float* __restrict__ part1, int tpi1, int C1,
                                                                const float* __restrict__ gamma,
                                                                const float* __restrict__ beta, int HW, int groups,
                                                                float eps, float* __restrict__ scale,
                                                                float* __restrict__ shift,
                                                                const float* __restrict__ film, int film_stride,
                                                                float* __restrict__ mean_rstd,
                                                                float* __restrict__ amax_out) {
    __shared__ double red[4][2];
    __shared__ float mean_s, rstd_s;
    __shared__ float qmax_s[4];
    float qmax = 0.f;      // largest per-(tile, channel) sum of squares of the group: sqrt() bounds max |x| (amax_out)
    const int b = blockIdx.x, g = blockIdx.y, tid = threadIdx.x;
    const int C = C0 + C1, cpg = C / groups;
    const int c_lo = g * cpg;
    double a = 0.0, q = 0.0;
    // the affine's own operands do not depend on the statistics: request them first (cpg <= 256 channels per group)
    float gam = 0.f, bet = 0.f, f_s = 0.f, f_t = 0.f;
    if (tid < cpg) {
        gam = gamma[c_lo + tid];
        bet = beta[c_lo + tid];
        if (film) {
            f_s = film[(size_t)b * film_stride + c_lo + tid];
            f_t = film[(size_t)b * film_stride + C + c_lo + tid];
        }
    }
    // channels of the group below C0 come from part0, the rest from part1
    const int n0 = max(0, min(C0, c_lo + cpg) - c_lo);
    const int n1 = cpg - n0;
    const int items0 = n0 * tpi0, items1 = n1 * tpi1;
    for (int it = tid; it < items0; it += 256) {
        const int t = it / n0, c = c_lo + (it - t * n0);
        const float2 v = *reinterpret_cast<const float2*>(part0 + (((size_t)b * tpi0 + t) * C0 + c) * 2);
        a += (double)v.x;
        q += (double)v.y;
        qmax = nan_max(qmax, v.y);
    }
    const int c1_lo = c_lo + n0 - C0;
    for (int it = tid; it < items1; it += 256) {
        const int t = it / n1, c = c1_lo + (it - t * n1);
        const float2 v = *reinterpret_cast<const float2*>(part1 + (((size_t)b * tpi1 + t) * C1 + c) * 2);
        a += (double)v.x;
        q += (double)v.y;
        qmax = nan_max(qmax, v.y);
    }
    // fixed-shape reduction (deterministic): xor-butterfly inside each wave, then the four wave sums in order --
    // one barrier instead of the eight of an LDS tree (this kernel is pure latency: 101 launches per ADM forward)
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a += __shfl_xor(a, o);
        q += __shfl_xor(q, o);
    }
    if (amax_out) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) qmax = fmaxf(qmax, __shfl_xor(qmax, o));
        if ((tid & 63) == 0) qmax_s[tid >> 6] = qmax;
    }
    if ((tid & 63) == 0) { red[tid >> 6][0] = a; red[tid >> 6][1] = q; }
    __syncthreads();
    if (amax_out && tid == 64) {
        // a NaN / inf partial propagates as inf (nan_max: fmaxf alone would DROP a NaN operand): the consumer's result
        // is non-finite either way
        const float m = fmaxf(fmaxf(qmax_s[0], qmax_s[1]), fmaxf(qmax_s[2], qmax_s[3]));
        amax_out[(size_t)b * groups + g] = sqrtf(m) * 1.0009765625f;        // (1 + 2^-10): rounding of the fp32 partial sums
    }
    if (tid == 0) {
        red[0][0] = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
        red[0][1] = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
    }
    if (tid == 0) {
        const double cnt = (double)HW * (double)cpg;
        const double mean = red[0][0] / cnt;
        double var = red[0][1] / cnt - mean * mean;
        var = var > 0.0 ? var : 0.0;
        mean_s = (float)mean;
        rstd_s = (float)(1.0 / sqrt(var + (double)eps));
        if (mean_rstd) {
            mean_rstd[((size_t)b * groups + g) * 2 + 0] = mean_s;
            mean_rstd[((size_t)b * groups + g) * 2 + 1] = rstd_s;
        }
    }
    __syncthreads();
    if (tid < cpg) {
        const int c = c_lo + tid;
        float sc = rstd_s * gam;
        float sh = bet - mean_s * sc;
        if (film) {
            const float s1 = 1.0f + f_s;
            sc = sc * s1;
            sh = sh * s1 + f_t;
        }
        scale[(size_t)b * C + c] = sc;
        shift[(size_t)b * C + c] = sh;
    }
}

extern "C" int ddnm_gn_finalize_tiles_amax_f32(const float* part0, int32_t tpi0, int32_t C0, const float* part1,
                                               int32_t tpi1, int32_t C1, const float* gamma, const float* beta, int32_t B,
                                               int32_t HW, int32_t groups, float eps, float* scale, float* shift,
                                               const float* film, int32_t film_stride, float* mean_rstd, float* amax_out,
                                               void* stream) {
    if (!part0 || !gamma || !beta || !scale || !shift || B <= 0 || tpi0 <= 0 || C0 <= 0) return DDNM_E_BADARG;
    if (C1 > 0 && (!part1 || tpi1 <= 0)) return DDNM_E_BADARG;
    const int C = C0 + C1;
    if (groups <= 0 || C % groups || C / groups > 256) return DDNM_E_SHAPE;
    if (amax_out && groups != DDNM_AMAX_N) return DDNM_E_SHAPE;
    DDNM_LAUNCH(gn_finalize_tiles_kernel, dim3(B, groups), dim3(256), 0, (hipStream_t)stream, part0, tpi0, C0, part1, tpi1,
                C1, gamma, beta, HW, groups, eps, scale, shift, film, film_stride, mean_rstd, amax_out);
    return 0;
}

extern "C" int ddnm_gn_finalize_tiles_f32(const float* part0, int32_t tpi0, int32_t C0, const float* part1,
                                          int32_t tpi1, int32_t C1, const float* gamma, const float* beta, int32_t B,
                                          int32_t HW, int32_t groups, float eps, float* scale, float* shift,
                                          const float* film, int32_t film_stride, float* mean_rstd, void* stream) {
    return ddnm_gn_finalize_tiles_amax_f32(part0, tpi0, C0, part1, tpi1, C1, gamma, beta, B, HW, groups, eps, scale, shift,
                                           film, film_stride, mean_rstd, nullptr, stream);
}

// ---------------------------------------------------------------------------------------------
// Operand bound of the split-fp16 convolutions (ddnm_conv_desc::amax_in) for raw operands that no GroupNorm reads first:
// grid (B, DDNM_AMAX_N), block (b, i) scans the i-th 1/32 slice of image b's data of up to two sources with 16-byte loads.
// kind 0: the tensor itself (max |x|); kind 1: its GroupNorm partials, pairs (sum, sum of squares): sqrt(max sumsq) >= max |x|.
// No atomics: 32 words per image, the consumer takes their maximum with scalar instructions.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void amax_bound_kernel(const float* __restrict__ src0, int64_t n40, int kind0,
                                                         const float* __restrict__ src1, int64_t n41, int kind1,
                                                         float* __restrict__ out) {
    __shared__ float red[4];
    const int b = blockIdx.x, i = blockIdx.y, tid = threadIdx.x;
    float m = 0.f;
    auto scan = [&](const float* __restrict__ src, int64_t n4, int kind) {
        const int64_t chunk = (n4 + DDNM_AMAX_N - 1) / DDNM_AMAX_N;
        const int64_t lo = (int64_t)i * chunk, hi = lo + chunk < n4 ? lo + chunk : n4;
        const f32x4* __restrict__ p = reinterpret_cast<const f32x4*>(src) + (int64_t)b * n4;
        float q = 0.f;
        for (int64_t j = lo + tid; j < hi; j += 256) {
            const f32x4 v = p[j];
            if (kind) q = nan_max(nan_max(q, v.y), v.w);
            else q = nan_max(nan_max(nan_max(nan_max(q, fabsf(v.x)), fabsf(v.y)), fabsf(v.z)), fabsf(v.w));
        }
        m = fmaxf(m, kind ? sqrtf(q) * 1.0009765625f : q);
    };
    scan(src0, n40, kind0);
    if (src1) scan(src1, n41, kind1);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    if ((tid & 63) == 0) red[tid >> 6] = m;
    __syncthreads();
    if (tid == 0) out[(size_t)b * DDNM_AMAX_N + i] = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
}

extern "C" int ddnm_amax_bound_f32(const float* src0, int64_t per_image0, int32_t kind0, const float* src1,
                                   int64_t per_image1, int32_t kind1, float* out, int32_t B, void* stream) {
    if (!src0 || !out || B <= 0 || per_image0 <= 0 || (per_image0 & 3)) return DDNM_E_BADARG;
    if (src1 && (per_image1 <= 0 || (per_image1 & 3))) return DDNM_E_BADARG;
    if ((kind0 | kind1) & ~1) return DDNM_E_BADARG;
    DDNM_LAUNCH(amax_bound_kernel, dim3(B, DDNM_AMAX_N), dim3(256), 0, (hipStream_t)stream, src0, per_image0 / 4, kind0, src1,
                src1 ? per_image1 / 4 : 0, kind1, out);
    return 0;
}


// ---------------------------------------------------------------------------------------------
// GroupNorm affine + swish applied once, written as fp16 (operand of conv3x3_halo_f16_kernel<SRC16>).
// HBM-bound: 4 B read + 2 B written per element; thread = 8 consecutive channels of one pixel.
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gn_apply_f16_kernel(const float* __restrict__ src0, const float* __restrict__ src1,
                                                           const float* __restrict__ scale, const float* __restrict__ shift,
                                                           _Float16* __restrict__ out, int HW, int C0, int C1, int silu,
                                                           size_t total8) {
    typedef _Float16 half8 __attribute__((ext_vector_type(8)));
    const int C = C0 + C1, C8 = C >> 3;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total8; i += (size_t)gridDim.x * 256) {
        const size_t pix = i / C8;
        const int c = (int)(i - pix * C8) * 8;
        const int b = (int)(pix / HW);
        const float* s = c < C0 ? src0 + pix * C0 + c : src1 + pix * C1 + (c - C0);
        const f32x4 v0 = *reinterpret_cast<const f32x4*>(s), v1 = *reinterpret_cast<const f32x4*>(s + 4);
        const float* sc = scale + (size_t)b * C + c;
        const float* sh = shift + (size_t)b * C + c;
        const f32x4 r0 = gn_act(v0, *reinterpret_cast<const f32x4*>(sc), *reinterpret_cast<const f32x4*>(sh), silu);
        const f32x4 r1 = gn_act(v1, *reinterpret_cast<const f32x4*>(sc + 4), *reinterpret_cast<const f32x4*>(sh + 4), silu);
        half8 h = {(_Float16)r0.x, (_Float16)r0.y, (_Float16)r0.z, (_Float16)r0.w,
                   (_Float16)r1.x, (_Float16)r1.y, (_Float16)r1.z, (_Float16)r1.w};
        *reinterpret_cast<half8*>(out + i * 8) = h;
    }
}

extern "C" int ddnm_gn_apply_f16(const float* src0, const float* src1, const float* scale, const float* shift, void* out_f16,
                                 int32_t B, int32_t HW, int32_t C0, int32_t C1, int32_t silu, void* stream) {
    if (!src0 || !scale || !shift || !out_f16 || B <= 0 || HW <= 0 || C0 <= 0 || C1 < 0) return DDNM_E_BADARG;
    if ((C0 | C1) & 7) return DDNM_E_SHAPE;
    if (C1 > 0 && !src1) return DDNM_E_BADARG;
    const size_t total8 = (size_t)B * HW * (C0 + C1) / 8;
    const size_t blocks = (total8 + 255) / 256;
    const unsigned g = (unsigned)(blocks < 16384 ? blocks : 16384);
    DDNM_LAUNCH(gn_apply_f16_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, src0, src1, scale, shift,
                reinterpret_cast<_Float16*>(out_f16), HW, C0, C1, silu, total8);
    return 0;
}


// ---------------------------------------------------------------------------------------------
// im2col of a 3x3 / stride 1 / pad 1 convolution input with the GroupNorm affine (+ swish) prologue, fp16 output:
// col[(b*HW + p)][tap*C + c] = act(concat_c(src0, src1)[b, y+ky-1, x+kx-1, c]) (0 outside the image).
// Only for the lowest-resolution level of the fp16 torso (8x8: 64 pixels per image): a 256-pixel MFMA tile would
// span four images, so the layer runs as ONE GEMM over [B*64][9*C] on conv1x1_f16_kernel instead (the column matrix
// is a few MB).  Thread = 8 consecutive channels of one (pixel, tap).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void im2col3x3_f16_kernel(const float* __restrict__ src0, const float* __restrict__ src1,
                                                            const float* __restrict__ scale, const float* __restrict__ shift,
                                                            _Float16* __restrict__ out, int H, int W, int C0, int C1,
                                                            int silu, size_t total8) {
    typedef _Float16 half8 __attribute__((ext_vector_type(8)));
    const int C = C0 + C1, C8 = C >> 3, HW = H * W;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total8; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C8) * 8;
        size_t t = i / C8;
        const int tap = (int)(t % 9);
        t /= 9;
        const int p = (int)(t % HW), b = (int)(t / HW);
        const int y = p / W + tap / 3 - 1, x = p % W + tap % 3 - 1;
        half8 h = {0, 0, 0, 0, 0, 0, 0, 0};
        if ((unsigned)y < (unsigned)H && (unsigned)x < (unsigned)W) {
            const size_t pix = (size_t)b * HW + (size_t)y * W + x;
            const float* s = c < C0 ? src0 + pix * C0 + c : src1 + pix * C1 + (c - C0);
            f32x4 v0 = *reinterpret_cast<const f32x4*>(s), v1 = *reinterpret_cast<const f32x4*>(s + 4);
            if (scale) {
                const float* sc = scale + (size_t)b * C + c;
                const float* sh = shift + (size_t)b * C + c;
                v0 = gn_act(v0, *reinterpret_cast<const f32x4*>(sc), *reinterpret_cast<const f32x4*>(sh), silu);
                v1 = gn_act(v1, *reinterpret_cast<const f32x4*>(sc + 4), *reinterpret_cast<const f32x4*>(sh + 4), silu);
            }
            h = half8{(_Float16)v0.x, (_Float16)v0.y, (_Float16)v0.z, (_Float16)v0.w,
                      (_Float16)v1.x, (_Float16)v1.y, (_Float16)v1.z, (_Float16)v1.w};
        }
        *reinterpret_cast<half8*>(out + i * 8) = h;
    }
}

extern "C" int ddnm_im2col3x3_f16(const float* src0, const float* src1, const float* scale, const float* shift,
                                  void* out_f16, int32_t B, int32_t H, int32_t W, int32_t C0, int32_t C1, int32_t silu,
                                  void* stream) {
    if (!src0 || !out_f16 || B <= 0 || H <= 0 || W <= 0 || C0 <= 0 || C1 < 0) return DDNM_E_BADARG;
    if ((scale == nullptr) != (shift == nullptr)) return DDNM_E_BADARG;
    if ((C0 | C1) & 7) return DDNM_E_SHAPE;
    if (C1 > 0 && !src1) return DDNM_E_BADARG;
    const size_t total8 = (size_t)B * H * W * 9 * (C0 + C1) / 8;
    const size_t blocks = (total8 + 255) / 256;
    const unsigned g = (unsigned)(blocks < 16384 ? blocks : 16384);
    DDNM_LAUNCH(im2col3x3_f16_kernel, dim3(g), dim3(256), 0, (hipStream_t)stream, src0, src1, scale, shift,
                reinterpret_cast<_Float16*>(out_f16), H, W, C0, C1, silu, total8);
    return 0;
}
