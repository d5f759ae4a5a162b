// Input-gradient kernels for classifier guidance (gfx950).
//
// The reference evaluates  classifier_scale * d/dx log softmax(classifier(x, t))[y]  with torch autograd
// (guided_diffusion/diffusion.py:183-189) through EncoderUNetModel (guided_diffusion/unet.py:684-895).
// Only the gradient with respect to activations is needed (no weight gradients), so the backward pass is
//   * data-gradient convolutions       = the forward implicit-GEMM kernels on flipped / transposed weights,
//   * GroupNorm(+FiLM)(+SiLU) backward  = gn_bwd_{reduce,finalize,apply} below (HBM-bound, 2 passes),
//   * attention backward                = batched MFMA GEMMs (gemm_f32.hip, with transposed-A support)
//                                         + softmax_bwd_rows,
//   * AttentionPool2d (unet.py:22-51)   = the pool_* kernels (only the class token's query matters).
#include "common.h"

#define GRID_1D(n) dim3((unsigned)(((n) + 255) / 256 < 8192 ? ((n) + 255) / 256 : 8192))

__device__ __forceinline__ float silu_grad(float u) {      // d/du [u * sigmoid(u)]
    const float s = __builtin_amdgcn_rcpf(1.0f + __expf(-u));
    return s * (1.0f + u * (1.0f - s));
}

// ------------------------------------------------------------------------------------------------
// GroupNorm (+FiLM) (+SiLU) backward.  Forward: u = x*sc + sh (sc = rstd*gamma', sh = beta' - mean*sc),
// a = silu(u) or u.  With t = dA * act'(u) * sc:   dx = t - P1/cnt - (x - mean) * rstd^2 * P2/cnt,
// P1 = sum_group t, P2 = sum_group t*(x - mean).   dA may live at half resolution behind the 2x2 average
// pool of a `down=True` ResBlock (dA_ups: read (y/2, x/2), times 0.25).
// ------------------------------------------------------------------------------------------------
constexpr int GNB_PIX_PER_THREAD = 32;

typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
// four consecutive channels of an fp32 or fp16 NHWC tensor (element index `i4` counts groups of four)
template <typename T> __device__ __forceinline__ f32x4 ld4(const T* __restrict__ p, size_t i4);
template <> __device__ __forceinline__ f32x4 ld4<float>(const float* __restrict__ p, size_t i4) {
    return reinterpret_cast<const f32x4*>(p)[i4];
}
template <> __device__ __forceinline__ f32x4 ld4<_Float16>(const _Float16* __restrict__ p, size_t i4) {
    const half4_t h = reinterpret_cast<const half4_t*>(p)[i4];
    return f32x4{(float)h.x, (float)h.y, (float)h.z, (float)h.w};
}
__device__ __forceinline__ void st4(float* __restrict__ p, size_t i4, f32x4 v) { reinterpret_cast<f32x4*>(p)[i4] = v; }
__device__ __forceinline__ void st4(_Float16* __restrict__ p, size_t i4, f32x4 v) {
    reinterpret_cast<half4_t*>(p)[i4] = half4_t{(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
}

// T = float: the fp32-tensor classifier path; T = _Float16: the fp16-activation path (activations AND their gradients
// fp16 NHWC in HBM like the reference's `classifier_use_fp16` autograd, arithmetic and every reduction fp32 / fp64)
template <typename T>
__global__ void gn_bwd_reduce_kernel(const T* __restrict__ x, const T* __restrict__ dA, int dA_ups,
                                     const float* __restrict__ sc, const float* __restrict__ sh,
                                     const float* __restrict__ mean_rstd, int silu, int H, int W, int C, int groups,
                                     double* __restrict__ partial, int nchunk, int pix_per_chunk) {
    extern __shared__ __attribute__((aligned(16))) double red[];   // [2][blockDim]
    const int C4 = C >> 2, HW = H * W;
    const int rows = blockDim.x / C4, active = rows * C4;
    const int tid = threadIdx.x, chunk = blockIdx.x, b = blockIdx.y;
    float p1 = 0.f, p2 = 0.f;
    if (tid < active) {
        const int c4 = tid % C4, prow = tid / C4, c = c4 * 4;
        const int g = c / (C / groups);
        const float mean = mean_rstd[((size_t)b * groups + g) * 2];
        const f32x4 a = *reinterpret_cast<const f32x4*>(sc + (size_t)b * C + c);
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(sh + (size_t)b * C + c);
        const int p_end = min(HW, (chunk + 1) * pix_per_chunk);
        for (int p = chunk * pix_per_chunk + prow; p < p_end; p += rows) {
            const f32x4 xv = ld4<T>(x, ((size_t)b * HW + p) * C4 + c4);
            f32x4 d;
            if (dA_ups) {
                const int yy = p / W, xx = p - yy * W;
                d = ld4<T>(dA, (((size_t)b * (H >> 1) + (yy >> 1)) * (W >> 1) + (xx >> 1)) * C4 + c4) * 0.25f;
            } else {
                d = ld4<T>(dA, ((size_t)b * HW + p) * C4 + c4);
            }
            f32x4 u = xv * a + t0;
            if (silu) { d.x *= silu_grad(u.x); d.y *= silu_grad(u.y); d.z *= silu_grad(u.z); d.w *= silu_grad(u.w); }
            const f32x4 t = d * a;
            const f32x4 xc = xv - mean;
            p1 += (t.x + t.y) + (t.z + t.w);
            p2 += (t.x * xc.x + t.y * xc.y) + (t.z * xc.z + t.w * xc.w);
        }
    }
    red[tid] = (double)p1;
    red[blockDim.x + tid] = (double)p2;
    __syncthreads();
    if (tid < groups) {
        const int q = (C / groups) >> 2;
        double a1 = 0.0, a2 = 0.0;
        for (int r = 0; r < rows; ++r)
            for (int j = 0; j < q; ++j) {
                const int t = r * C4 + tid * q + j;
                a1 += red[t];
                a2 += red[blockDim.x + t];
            }
        double* o = partial + (((size_t)b * nchunk + chunk) * groups + tid) * 2;
        o[0] = a1;
        o[1] = a2;
    }
}

// coef[b][g] = (P1/cnt, rstd^2 * P2/cnt).  One workgroup per (sample, group): 256 threads stride over the chunk partials
// (up to 256 of them at 256^2), fp64 xor-butterfly inside each wave, the four wave sums added in order -- fixed
// shape, deterministic.  (One workgroup per SAMPLE with 8 threads per group took 13 us per launch: 46 launches per
// guidance evaluation.)
__global__ __launch_bounds__(256) void gn_bwd_finalize_kernel(const double* __restrict__ partial, int nchunk,
                                                              const float* __restrict__ mean_rstd, int HW, int C, int groups,
                                                              float* __restrict__ coef) {
    __shared__ double red[4][2];
    const int b = blockIdx.x, g = blockIdx.y, tid = threadIdx.x;
    const float rstd_f = mean_rstd[((size_t)b * groups + g) * 2 + 1];
    double a1 = 0.0, a2 = 0.0;
    for (int ch = tid; ch < nchunk; ch += 256) {
        const double* p = partial + (((size_t)b * nchunk + ch) * groups + g) * 2;
        a1 += p[0];
        a2 += p[1];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        a1 += __shfl_xor(a1, o);
        a2 += __shfl_xor(a2, o);
    }
    if ((tid & 63) == 0) { red[tid >> 6][0] = a1; red[tid >> 6][1] = a2; }
    __syncthreads();
    if (tid != 0) return;
    a1 = (red[0][0] + red[1][0]) + (red[2][0] + red[3][0]);
    a2 = (red[0][1] + red[1][1]) + (red[2][1] + red[3][1]);
    const double cnt = (double)HW * (double)(C / groups);
    const double rstd = (double)rstd_f;
    coef[((size_t)b * groups + g) * 2 + 0] = (float)(a1 / cnt);
    coef[((size_t)b * groups + g) * 2 + 1] = (float)(rstd * rstd * a2 / cnt);
}

// dx = t - c1 - (x - mean)*c2  [+ add (optionally through the same half-resolution x0.25 mapping)]
template <typename T>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ dA,
                                                           int dA_ups, const float* __restrict__ sc,
                                                           const float* __restrict__ sh,
                                                           const float* __restrict__ mean_rstd,
                                                           const float* __restrict__ coef, int silu,
                                                           const T* __restrict__ add, int add_ups, int H, int W,
                                                           int C, int groups, T* __restrict__ dx, size_t total4) {
    const int C4 = C >> 2, HW = H * W, cpg = C / groups;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C4) * 4;
        const size_t pix = i / C4;
        const size_t b = pix / HW;
        const int p = (int)(pix - b * HW);
        const int g = c / cpg;
        const float mean = mean_rstd[(b * groups + g) * 2];
        const float c1 = coef[(b * groups + g) * 2], c2 = coef[(b * groups + g) * 2 + 1];
        const f32x4 a = *reinterpret_cast<const f32x4*>(sc + b * C + c);
        const f32x4 t0 = *reinterpret_cast<const f32x4*>(sh + b * C + c);
        const f32x4 xv = ld4<T>(x, i);
        const int yy = p / W, xx = p - yy * W;
        const size_t half_idx = ((b * (H >> 1) + (yy >> 1)) * (size_t)(W >> 1) + (xx >> 1)) * C4 + (c >> 2);
        f32x4 d = dA_ups ? ld4<T>(dA, half_idx) * 0.25f : ld4<T>(dA, i);
        f32x4 u = xv * a + t0;
        if (silu) { d.x *= silu_grad(u.x); d.y *= silu_grad(u.y); d.z *= silu_grad(u.z); d.w *= silu_grad(u.w); }
        f32x4 r = d * a - c1 - (xv - mean) * c2;
        if (add) r = r + (add_ups ? ld4<T>(add, half_idx) * 0.25f : ld4<T>(add, i));
        st4(dx, i, r);
    }
}

static inline int gnb_block_dim(int C4) { return C4 <= 256 ? 256 : (C4 <= 512 ? 512 : 1024); }

extern "C" int ddnm_gn_bwd_nchunk(int32_t HW, int32_t C) {
    const int C4 = C / 4;
    if (C4 <= 0 || C4 > 1024) return DDNM_E_SHAPE;
    const int rows = gnb_block_dim(C4) / C4;
    const int pix = rows * GNB_PIX_PER_THREAD;
    return (HW + pix - 1) / pix;
}

template <typename T>
static int gn_bwd_launch(const T* x, const T* dA, int32_t dA_ups, const float* gn_scale,
                         const float* gn_shift, const float* mean_rstd, int32_t silu, const T* add,
                         int32_t add_ups, int32_t B, int32_t H, int32_t W, int32_t C, int32_t groups,
                         double* partial, int32_t nchunk, float* coef, T* dx, void* stream) {
    if (!x || !dA || !gn_scale || !gn_shift || !mean_rstd || !partial || !coef || !dx) return DDNM_E_BADARG;
    if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || groups <= 0 || groups > 64) return DDNM_E_BADARG;
    if (C % (groups * 4) || C / 4 > 1024) return DDNM_E_SHAPE;
    if ((dA_ups || add_ups) && ((H | W) & 1)) return DDNM_E_SHAPE;
    const int C4 = C / 4, bd = gnb_block_dim(C4);
    const int rows = bd / C4, pix = rows * GNB_PIX_PER_THREAD, HW = H * W;
    if (nchunk != (HW + pix - 1) / pix) return DDNM_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    DDNM_LAUNCH(gn_bwd_reduce_kernel<T>, dim3(nchunk, B), dim3(bd), 2 * bd * sizeof(double), s, x, dA, dA_ups, gn_scale,
                gn_shift, mean_rstd, silu, H, W, C, groups, partial, nchunk, pix);
    DDNM_LAUNCH(gn_bwd_finalize_kernel, dim3(B, groups), dim3(256), 0, s, partial, nchunk, mean_rstd, HW, C, groups, coef);
    const size_t total4 = (size_t)B * HW * C4;
    DDNM_LAUNCH(gn_bwd_apply_kernel<T>, GRID_1D(total4), dim3(256), 0, s, x, dA, dA_ups, gn_scale, gn_shift, mean_rstd,
                coef, silu, add, add_ups, H, W, C, groups, dx, total4);
    return 0;
}

extern "C" int ddnm_gn_bwd_f32(const float* x, const float* dA, int32_t dA_ups, const float* gn_scale,
                               const float* gn_shift, const float* mean_rstd, int32_t silu, const float* add,
                               int32_t add_ups, int32_t B, int32_t H, int32_t W, int32_t C, int32_t groups,
                               double* partial, int32_t nchunk, float* coef, float* dx, void* stream) {
    return gn_bwd_launch<float>(x, dA, dA_ups, gn_scale, gn_shift, mean_rstd, silu, add, add_ups, B, H, W, C, groups, partial,
                                nchunk, coef, dx, stream);
}

extern "C" int ddnm_gn_bwd_h16(const void* x, const void* dA, int32_t dA_ups, const float* gn_scale,
                               const float* gn_shift, const float* mean_rstd, int32_t silu, const void* add,
                               int32_t add_ups, int32_t B, int32_t H, int32_t W, int32_t C, int32_t groups,
                               double* partial, int32_t nchunk, float* coef, void* dx, void* stream) {
    return gn_bwd_launch<_Float16>(reinterpret_cast<const _Float16*>(x), reinterpret_cast<const _Float16*>(dA), dA_ups,
                                   gn_scale, gn_shift, mean_rstd, silu, reinterpret_cast<const _Float16*>(add), add_ups, B, H,
                                   W, C, groups, partial, nchunk, coef, reinterpret_cast<_Float16*>(dx), stream);
}

// ------------------------------------------------------------------------------------------------
// softmax backward, in place on dP:  dS = scale * P .* (dP - rowsum(dP .* P))   (one wave per row)
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void softmax_bwd_rows_kernel(const float* __restrict__ P, float* __restrict__ dP,
                                                               int64_t rows, int n, int ld, float scale) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* p = P + row * ld;
    float* d = dP + row * ld;
    float dot = 0.f;
    for (int c = lane; c < n; c += 64) dot += d[c] * p[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
    for (int c = lane; c < n; c += 64) d[c] = scale * p[c] * (d[c] - dot);
}

extern "C" int ddnm_softmax_bwd_rows_f32(const float* P, float* dP, int64_t rows, int32_t n, int32_t ld, float scale,
                                         void* stream) {
    if (!P || !dP || rows <= 0 || n <= 0) return DDNM_E_BADARG;
    DDNM_LAUNCH(softmax_bwd_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, P, dP, rows,
                n, ld, scale);
    return 0;
}

// ------------------------------------------------------------------------------------------------
// AttentionPool2d (guided_diffusion/unet.py:22-51).  Tokens X[b][0] = mean_p act(h[b][p]) + pos[:,0],
// X[b][1+p] = act(h[b][p]) + pos[:,1+p], act = silu(GroupNorm(h)) folded as (sc, sh).  Only token 0 is read
// from the attention output, so only its query matters.
// ------------------------------------------------------------------------------------------------
template <typename TH>
__global__ __launch_bounds__(256) void pool_tokens_kernel(const TH* __restrict__ h, const float* __restrict__ sc,
                                                          const float* __restrict__ sh, const float* __restrict__ pos,
                                                          float* __restrict__ X, int HW, int C) {
    const int b = blockIdx.x, T = HW + 1;
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        const float a = sc[(size_t)b * C + c], t0 = sh[(size_t)b * C + c];
        float sum = 0.f;
        for (int p = 0; p < HW; ++p) {
            const float v = silu_f((float)h[((size_t)b * HW + p) * C + c] * a + t0);
            sum += v;
            X[((size_t)b * T + 1 + p) * C + c] = v + pos[(size_t)c * T + 1 + p];
        }
        X[((size_t)b * T) * C + c] = sum / (float)HW + pos[(size_t)c * T];
    }
}

extern "C" int ddnm_pool_tokens_f32(const float* h, const float* gn_scale, const float* gn_shift, const float* pos,
                                    float* X, int32_t B, int32_t HW, int32_t C, void* stream) {
    if (!h || !gn_scale || !gn_shift || !pos || !X || B <= 0 || HW <= 0 || C <= 0) return DDNM_E_BADARG;
    DDNM_LAUNCH(pool_tokens_kernel<float>, dim3(B), dim3(256), 0, (hipStream_t)stream, h, gn_scale, gn_shift, pos, X, HW, C);
    return 0;
}

// the same over the fp16 NHWC activation of the fp16-activation classifier path (tokens stay fp32)
extern "C" int ddnm_pool_tokens_h16(const void* h, const float* gn_scale, const float* gn_shift, const float* pos,
                                    float* X, int32_t B, int32_t HW, int32_t C, void* stream) {
    if (!h || !gn_scale || !gn_shift || !pos || !X || B <= 0 || HW <= 0 || C <= 0) return DDNM_E_BADARG;
    DDNM_LAUNCH(pool_tokens_kernel<_Float16>, dim3(B), dim3(256), 0, (hipStream_t)stream, reinterpret_cast<const _Float16*>(h),
                gn_scale, gn_shift, pos, X, HW, C);
    return 0;
}

// one workgroup (64 threads) per (b, head); qkv [B][T][3C] in the NEW order (q | k | v blocks of C, head-major inside)
__global__ __launch_bounds__(64) void pool_attn_fwd_kernel(const float* __restrict__ qkv, float* __restrict__ P,
                                                           float* __restrict__ a0, int T, int C, int heads) {
    __shared__ float prob[1024];
    const int b = blockIdx.x / heads, hd = blockIdx.x % heads, lane = threadIdx.x;
    const int ch = C / heads;                               // 64
    const float* base = qkv + (size_t)b * T * 3 * C;
    const float* q0 = base + hd * ch;
    float mx = -INFINITY;
    for (int s = lane; s < T; s += 64) {
        const float* k = base + (size_t)s * 3 * C + C + hd * ch;
        float d = 0.f;
        for (int c = 0; c < ch; ++c) d += q0[c] * k[c];
        d *= rsqrtf((float)ch);                             // (q*s).(k*s), s = ch^-1/4
        prob[s] = d;
        mx = fmaxf(mx, d);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
    for (int s = lane; s < T; s += 64) { const float e = expf(prob[s] - mx); prob[s] = e; sum += e; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    __syncthreads();
    const float inv = 1.0f / sum;
    for (int s = lane; s < T; s += 64) { prob[s] *= inv; P[((size_t)b * heads + hd) * T + s] = prob[s]; }
    __syncthreads();
    for (int c = lane; c < ch; c += 64) {
        float acc = 0.f;
        for (int s = 0; s < T; ++s) acc += prob[s] * base[(size_t)s * 3 * C + 2 * C + hd * ch + c];
        a0[(size_t)b * C + hd * ch + c] = acc;
    }
}

extern "C" int ddnm_pool_attn_fwd_f32(const float* qkv, float* P, float* a0, int32_t B, int32_t T, int32_t C,
                                      int32_t heads, void* stream) {
    if (!qkv || !P || !a0 || B <= 0 || T <= 0 || T > 1024 || heads <= 0 || C % heads) return DDNM_E_BADARG;
    DDNM_LAUNCH(pool_attn_fwd_kernel, dim3(B * heads), dim3(64), 0, (hipStream_t)stream, qkv, P, a0, T, C, heads);
    return 0;
}

// backward of the above: dqkv [B][T][3C] (fully written: dq rows of tokens >= 1 are zero)
__global__ __launch_bounds__(64) void pool_attn_bwd_kernel(const float* __restrict__ qkv, const float* __restrict__ P,
                                                           const float* __restrict__ da0, float* __restrict__ dqkv,
                                                           int T, int C, int heads) {
    __shared__ float ds[1024];
    const int b = blockIdx.x / heads, hd = blockIdx.x % heads, lane = threadIdx.x;
    const int ch = C / heads;
    const float* base = qkv + (size_t)b * T * 3 * C;
    float* dbase = dqkv + (size_t)b * T * 3 * C;
    const float* prob = P + ((size_t)b * heads + hd) * T;
    const float* g = da0 + (size_t)b * C + hd * ch;
    float dot = 0.f;
    for (int s = lane; s < T; s += 64) {
        const float* v = base + (size_t)s * 3 * C + 2 * C + hd * ch;
        float dp = 0.f;
        for (int c = 0; c < ch; ++c) dp += g[c] * v[c];
        ds[s] = dp;
        dot += dp * prob[s];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) dot += __shfl_xor(dot, o);
    __syncthreads();
    const float sc2 = rsqrtf((float)ch);
    for (int s = lane; s < T; s += 64) ds[s] = sc2 * prob[s] * (ds[s] - dot);
    __syncthreads();
    const float* q0 = base + hd * ch;
    for (int c = lane; c < ch; c += 64) {
        float dq = 0.f;
        for (int s = 0; s < T; ++s) {
            dq += ds[s] * base[(size_t)s * 3 * C + C + hd * ch + c];
            dbase[(size_t)s * 3 * C + C + hd * ch + c] = ds[s] * q0[c];          // dk
            dbase[(size_t)s * 3 * C + 2 * C + hd * ch + c] = prob[s] * g[c];     // dv
            if (s > 0) dbase[(size_t)s * 3 * C + hd * ch + c] = 0.f;             // dq of non-class tokens
        }
        dbase[hd * ch + c] = dq;
    }
}

extern "C" int ddnm_pool_attn_bwd_f32(const float* qkv, const float* P, const float* da0, float* dqkv, int32_t B,
                                      int32_t T, int32_t C, int32_t heads, void* stream) {
    if (!qkv || !P || !da0 || !dqkv || B <= 0 || T <= 0 || T > 1024 || heads <= 0 || C % heads) return DDNM_E_BADARG;
    DDNM_LAUNCH(pool_attn_bwd_kernel, dim3(B * heads), dim3(64), 0, (hipStream_t)stream, qkv, P, da0, dqkv, T, C, heads);
    return 0;
}

// d act[b][p][c] = dX[b][1+p][c] + dX[b][0][c] / HW
template <typename T>
__global__ __launch_bounds__(256) void pool_tokens_bwd_kernel(const float* __restrict__ dX, T* __restrict__ dact,
                                                              int HW, int C, size_t total) {
    const int Tk = HW + 1;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % C);
        const size_t bp = i / C;
        const size_t b = bp / HW, p = bp - b * HW;
        dact[i] = (T)(dX[(b * Tk + 1 + p) * C + c] + dX[(b * Tk) * C + c] / (float)HW);
    }
}

extern "C" int ddnm_pool_tokens_bwd_f32(const float* dX, float* dact, int32_t B, int32_t HW, int32_t C, void* stream) {
    if (!dX || !dact || B <= 0 || HW <= 0 || C <= 0) return DDNM_E_BADARG;
    const size_t total = (size_t)B * HW * C;
    DDNM_LAUNCH(pool_tokens_bwd_kernel<float>, GRID_1D(total), dim3(256), 0, (hipStream_t)stream, dX, dact, HW, C, total);
    return 0;
}

extern "C" int ddnm_pool_tokens_bwd_h16(const float* dX, void* dact, int32_t B, int32_t HW, int32_t C, void* stream) {
    if (!dX || !dact || B <= 0 || HW <= 0 || C <= 0) return DDNM_E_BADARG;
    const size_t total = (size_t)B * HW * C;
    DDNM_LAUNCH(pool_tokens_bwd_kernel<_Float16>, GRID_1D(total), dim3(256), 0, (hipStream_t)stream, dX,
                reinterpret_cast<_Float16*>(dact), HW, C, total);
    return 0;
}

// dlogits[b][j] = [j == y[b]] - softmax(logits[b])[j]    (gradient of log_softmax(logits)[y], diffusion.py:186-188)
__global__ __launch_bounds__(256) void logsoftmax_grad_kernel(const float* __restrict__ logits, const int64_t* __restrict__ y,
                                                              float* __restrict__ dl, int N) {
    __shared__ float red[256];
    const int b = blockIdx.x, tid = threadIdx.x;
    const float* l = logits + (size_t)b * N;
    float mx = -INFINITY;
    for (int j = tid; j < N; j += 256) mx = fmaxf(mx, l[j]);
    red[tid] = mx;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] = fmaxf(red[tid], red[tid + s]); __syncthreads(); }
    mx = red[0];
    __syncthreads();
    float sum = 0.f;
    for (int j = tid; j < N; j += 256) sum += expf(l[j] - mx);
    red[tid] = sum;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
    sum = red[0];
    const int yy = (int)y[b];
    for (int j = tid; j < N; j += 256) dl[(size_t)b * N + j] = (j == yy ? 1.0f : 0.0f) - expf(l[j] - mx) / sum;
}

extern "C" int ddnm_logsoftmax_grad_f32(const float* logits, const int64_t* y, float* dlogits, int32_t B, int32_t N,
                                        void* stream) {
    if (!logits || !y || !dlogits || B <= 0 || N <= 0) return DDNM_E_BADARG;
    DDNM_LAUNCH(logsoftmax_grad_kernel, dim3(B), dim3(256), 0, (hipStream_t)stream, logits, y, dlogits, N);
    return 0;
}
