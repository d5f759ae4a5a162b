// Pieces shared by the fp32 and fp16-operand implicit-GEMM convolution kernels (gfx950).
#pragma once
#include "common.h"

struct ConvArgs {
    ddnm_conv_desc d;
    int Cin, ntaps, Hs, Ws;
    int m_tiles, n_tiles;
    int tiles_x;        // 2-D tiling of the output image (0: flat strips, gather kernel only)
    int TW, TW_log2;    // tile width in pixels (power of two)
    int ksplit;         // workgroups per output tile along K (channel chunks)
    float* ws;          // split-K slabs [ksplit][B*Ho*Wo][Cout]
};

// The kernels address the output / residual / split-K slabs with 32-bit ELEMENT offsets and the operands (sources, packed
// weights) through raw buffer descriptors with 32-bit BYTE sizes: every tensor of a launch must stay below 2^31 elements
// (output side) resp. 2 GiB (operand side; the out-of-range sentinel 0x80000000 fetches the zero padding).  Larger
// launches are refused (split the batch on the host) instead of wrapping around silently.
static inline bool conv_sizes_addressable(const ddnm_conv_desc* d) {
    const int64_t lim = (int64_t)1 << 31;
    const int64_t out_el = (int64_t)d->B * d->Ho * d->Wo * d->Cout;
    const int64_t hs = d->ups ? d->Hin / 2 : d->Hin, ws = d->ups ? d->Win / 2 : d->Win;
    const int64_t cmax = d->C0 > d->C1 ? d->C0 : d->C1;
    const int64_t src_b = (int64_t)d->B * hs * ws * cmax * (d->src_f16 ? 2 : 4);
    const int64_t cout_pad = ((int64_t)d->Cout + 127) / 128 * 128;
    const int64_t w_b = cout_pad * d->ksize * d->ksize * ((int64_t)d->C0 + d->C1) * 4;      // fp32 / split packing; fp16 is half
    const int64_t sk_el = (int64_t)d->B * d->Ho * d->Wo * (d->SC0 > d->SC1 ? d->SC0 : d->SC1);
    return out_el < lim && src_b < lim && w_b < lim && sk_el < lim;
}

// persistent form of the split-fp16 3x3 kernel (conv_s16_persist.hip), dispatched from conv_igemm_f16.hip::run_f16
bool conv3x3_s16_persist_eligible(const ConvArgs& p);
int conv3x3_s16_persist_launch(const ConvArgs& p, hipStream_t s);

// ---- tile geometry helper: local row r of M-tile -> output pixel
struct TileMap {
    int img, ty0, tx0, th_unused, TW, TW_log2, flat_base, Wo;
    bool two_d;
    __device__ __forceinline__ void pixel(int r, int& oy, int& ox) const {
        if (two_d) {
            oy = ty0 + (r >> TW_log2);
            ox = tx0 + (r & (TW - 1));
        } else {
            const int pix = flat_base + r;
            oy = pix / Wo;
            ox = pix - oy * Wo;
        }
    }
};

template <int BM>
__device__ __forceinline__ TileMap make_tilemap(const ConvArgs& p, int m_tile) {
    TileMap t;
    const int HWo = p.d.Ho * p.d.Wo;
    const int per_img = HWo / BM;
    t.img = m_tile / per_img;
    const int t_in_img = m_tile - t.img * per_img;
    t.TW = p.TW;
    t.TW_log2 = p.TW_log2;
    t.Wo = p.d.Wo;
    t.two_d = p.tiles_x != 0;
    if (t.two_d) {
        const int ty = t_in_img / p.tiles_x, tx = t_in_img - ty * p.tiles_x;
        t.ty0 = ty * (BM >> p.TW_log2);
        t.tx0 = tx << p.TW_log2;
        t.flat_base = 0;
    } else {
        t.ty0 = t.tx0 = 0;
        t.flat_base = t_in_img * BM;
    }
    return t;
}

// ---- shared epilogue.  C/D map of 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5)
template <int WM, int WN, int MT, int NT, bool RES_ALL_UPFRONT = false>
__device__ __forceinline__ void conv_epilogue(const ConvArgs& p, const TileMap& tm, int n_tile, int m_tile, int slice,
                                              f32x16 (&acc)[MT][NT], float* lds, const float acc_scale = 1.f) {
    constexpr int BN = WN * NT * 32;
    const ddnm_conv_desc& d = p.d;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int ncol = lane & 31, rsel = 4 * (lane >> 5);
    const bool partial = p.ksplit > 1;
    float* __restrict__ ws = partial ? p.ws + (size_t)slice * d.B * d.Ho * d.Wo * d.Cout : nullptr;
    const float* __restrict__ res = d.res;
    float* __restrict__ out = d.out;
    const bool want_stats = d.stats_out != nullptr && !partial;
    float cs[NT], cq[NT];              // per-lane column (= output channel) partial sum / sum of squares
#pragma unroll
    for (int j = 0; j < NT; ++j) cs[j] = cq[j] = 0.f;
    if (RES_ALL_UPFRONT && d.Cout % BN == 0 && !d.out_nchw) {
        // One workgroup per CU (fp16 kernel): nothing else hides the residual round trip, so ALL residual
        // loads of the wave's MT x NT tiles are issued before the first add/store (one latency, not MT*NT).
        unsigned o[MT][NT][16];
        float rv[MT][NT][16];
        const bool use_res = res && !partial;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (wm * MT + i) * 32 + (r & 3) + 8 * (r >> 2) + rsel;
                int oy, ox;
                tm.pixel(row, oy, ox);
                const unsigned pix = (unsigned)((tm.img * d.Ho + oy) * d.Wo + ox);
                const unsigned rpix = d.res_ups ? (unsigned)((tm.img * (d.Ho >> 1) + (oy >> 1)) * (d.Wo >> 1) + (ox >> 1)) : pix;
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    const unsigned n = (unsigned)(n_tile * BN + (wn * NT + j) * 32 + ncol);
                    o[i][j][r] = pix * (unsigned)d.Cout + n;
                    rv[i][j][r] = use_res ? res[(size_t)rpix * d.Cout + n] : 0.f;
                }
            }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = n_tile * BN + (wn * NT + j) * 32 + ncol;
            float add = 0.f;
            if (!partial) {
                if (d.bias) add = d.bias[n];
                if (d.badd) add += d.badd[(size_t)tm.img * d.badd_stride + n];
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = acc[i][j][r] * acc_scale + add + rv[i][j][r];
                    if (partial) ws[o[i][j][r]] = v;
                    else out[o[i][j][r]] = v;
                    cs[j] += v;
                    cq[j] = __builtin_fmaf(v, v, cq[j]);      // explicit: every kernel that shares this epilogue (and conv_s16_persist.hip) rounds alike
                }
        }
    } else {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = n_tile * BN + (wn * NT + j) * 32 + ncol;
        if (n >= d.Cout) continue;
        float add = 0.f;
        if (!partial) {
            if (d.bias) add = d.bias[n];
            if (d.badd) add += d.badd[(size_t)tm.img * d.badd_stride + n];
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            // 32-bit element offsets (tensors here are < 2^31 elements): 16 registers instead of 32
            unsigned o[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (wm * MT + i) * 32 + (r & 3) + 8 * (r >> 2) + rsel;
                int oy, ox;
                tm.pixel(row, oy, ox);
                const unsigned pix = (unsigned)((tm.img * d.Ho + oy) * d.Wo + ox);
                if (d.res_ups && res && !partial) {
                    o[r] = pix;                 // resolved below: two different address forms are needed
                } else {
                    o[r] = d.out_nchw && !partial ? (unsigned)(((tm.img * d.Cout + n) * d.Ho + oy) * d.Wo + ox)
                                                  : pix * (unsigned)d.Cout + (unsigned)n;
                }
            }
            // all 16 residual loads of this 32x32 tile are issued before the first store (a
            // load -> add -> store chain per element would serialise 16 HBM round trips)
            float rv[16];
            if (res && !partial && d.res_ups) {      // residual read through a nearest x2 upsample
                const unsigned hw = (unsigned)(d.Ho * d.Wo);
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const unsigned pin = o[r] - (unsigned)tm.img * hw;
                    const unsigned oy = pin / (unsigned)d.Wo, ox = pin - oy * (unsigned)d.Wo;
                    rv[r] = res[(size_t)((tm.img * (d.Ho >> 1) + (int)(oy >> 1)) * (d.Wo >> 1) + (int)(ox >> 1)) * d.Cout + n];
                    o[r] = o[r] * (unsigned)d.Cout + (unsigned)n;
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) rv[r] = (res && !partial) ? res[o[r]] : 0.f;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[i][j][r] * acc_scale + add + rv[r];
                if (partial) ws[o[r]] = v;
                else out[o[r]] = v;
                cs[j] += v;
                cq[j] = __builtin_fmaf(v, v, cq[j]);      // explicit: every kernel that shares this epilogue (and conv_s16_persist.hip) rounds alike
            }
            // 128 x 64 wave tiles (8 accumulator tiles live): keep the scheduler from interleaving the loads / stores of
            // several tiles, which pushed the 4-wave split kernel over its 256-register budget (8 spilled registers)
            if constexpr (MT * NT > 4) __builtin_amdgcn_sched_barrier(0);
        }
    }
    }
    // GroupNorm statistics of the tensor just produced, emitted here so the consumer's GroupNorm needs no
    // extra pass over HBM: per (M tile, channel) fp32 partials over the tile's rows, fixed order.
    if (want_stats) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            cs[j] += __shfl_xor(cs[j], 32);
            cq[j] += __shfl_xor(cq[j], 32);
        }
        __syncthreads();                       // every wave is done with the operand tiles in LDS
        if (lane < 32) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int c = (wn * NT + j) * 32 + lane;
                lds[(wm * BN + c) * 2 + 0] = cs[j];
                lds[(wm * BN + c) * 2 + 1] = cq[j];
            }
        }
        __syncthreads();
        for (int c = threadIdx.x; c < BN; c += 256) {
            const int n = n_tile * BN + c;
            if (n < d.Cout) {
                float a = 0.f, q = 0.f;
#pragma unroll
                for (int w = 0; w < WM; ++w) { a += lds[(w * BN + c) * 2]; q += lds[(w * BN + c) * 2 + 1]; }
                d.stats_out[((size_t)m_tile * d.Cout + n) * 2 + 0] = a;
                d.stats_out[((size_t)m_tile * d.Cout + n) * 2 + 1] = q;
            }
        }
    }
}

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

// fp32 x4 -> hi | lo fp16 halves of a split row: hi at dst, lo 32 halfs behind it (split-fp16 kernels: conv_igemm_f16.hip <.., SPLIT>, conv_gather_s16.hip)
// `s` = the launch's power-of-two operand scale (s16_operand_scale below; 1 for GroupNorm'd operands).  The MFMA honours
// fp16 subnormals, so the absolute error of hi + lo is <= 2^-25 for |s v| < 0.25 and <= 2^-22 |v| above.
#define DDNM_S16_ASCALE 1.0f       // compile-time pre-scale of ABI 4, kept at 1: the scale is per launch and image now
__device__ __forceinline__ void split_store(_Float16* dst, f32x4 v, const float s) {
    v = v * s;
    const half4 h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
    const half4 l = {(_Float16)(v.x - (float)h.x), (_Float16)(v.y - (float)h.y), (_Float16)(v.z - (float)h.z),
                     (_Float16)(v.w - (float)h.w)};
    *reinterpret_cast<half4*>(dst) = h;
    *reinterpret_cast<half4*>(dst + 32) = l;
}
__device__ __forceinline__ void split_store(_Float16* dst, f32x4 v) {      // unscaled (GroupNorm'd operand, no raw shortcut)
    const half4 h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
    const half4 l = {(_Float16)(v.x - (float)h.x), (_Float16)(v.y - (float)h.y), (_Float16)(v.z - (float)h.z),
                     (_Float16)(v.w - (float)h.w)};
    *reinterpret_cast<half4*>(dst) = h;
    *reinterpret_cast<half4*>(dst + 32) = l;
}

// Operand-range guard of the split forms (include/ddnm_hip.h::ddnm_conv_desc::amax_in): the DDNM_AMAX_N bound words of
// image `img` are wave-uniform scalar loads (s_load: they count on lgkmcnt, not on the vmcnt the main loops count), their
// maximum is taken on the bit patterns (non-negative floats order like unsigned integers), and the scale is the power of
// two that brings the bound into [2^14, 2^15).  `down_only`: the raw operand shares its accumulator with a GroupNorm'd
// one (fused shortcut), which is O(1): scale down for huge inputs, never up.  The exponent is clamped so that the scale,
// its inverse and their product with acc_scale stay normal numbers; an all-zero (or non-finite) operand needs no care.
__device__ __forceinline__ void s16_operand_scale(const float* __restrict__ amax_in, int img, bool down_only,
                                                  float& scale, float& inv_scale) {
    scale = inv_scale = 1.f;
    if (amax_in == nullptr) return;
    const unsigned* __restrict__ w = reinterpret_cast<const unsigned*>(amax_in) + (size_t)img * DDNM_AMAX_N;
    unsigned m = 0u;
#pragma unroll
    for (int i = 0; i < DDNM_AMAX_N; ++i) m = w[i] > m ? w[i] : m;
    m = __builtin_amdgcn_readfirstlane(m);
    int e = (int)((m >> 23) & 0xffu);                 // biased exponent of the bound (sign bit is 0)
    e = e < 47 ? 47 : (e > 207 ? 207 : e);            // 2^-80 ... 2^80
    int k = 14 - (e - 127);
    if (down_only && k > 0) k = 0;
    scale = __uint_as_float((unsigned)(127 + k) << 23);
    inv_scale = __uint_as_float((unsigned)(127 - k) << 23);
}

__device__ __forceinline__ f32x4 gn_act(f32x4 v, const f32x4 gsc, const f32x4 gsh, int silu) {
    v = v * gsc + gsh;
    if (silu) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
    return v;
}


// =====================================================================================
// split-K reduction + epilogue:  out = sum_s ws[s] + bias + badd + res   (fixed order)
// =====================================================================================
static __global__ __launch_bounds__(256) void conv_splitk_reduce_kernel(const ConvArgs p, size_t total4) {
    const ddnm_conv_desc& d = p.d;
    const size_t slab4 = total4;
    const int c4n = d.Cout >> 2;
    const size_t hw = (size_t)d.Ho * d.Wo;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        f32x4 v = reinterpret_cast<const f32x4*>(p.ws)[i];
        for (int s = 1; s < p.ksplit; ++s) v = v + reinterpret_cast<const f32x4*>(p.ws)[i + s * slab4];
        const int n = (int)(i % c4n) * 4;
        const size_t pix = i / c4n;
        const size_t b = pix / hw;
        if (d.bias) v = v + *reinterpret_cast<const f32x4*>(d.bias + n);
        if (d.badd) v = v + *reinterpret_cast<const f32x4*>(d.badd + b * d.badd_stride + n);
        if (d.res) {
            if (d.res_ups) {
                const size_t p2 = pix - b * hw;
                const size_t oy = p2 / d.Wo, ox = p2 - oy * d.Wo;
                v = v + *reinterpret_cast<const f32x4*>(d.res + ((b * (d.Ho >> 1) + (oy >> 1)) * (d.Wo >> 1) + (ox >> 1)) * d.Cout + n);
            } else {
                v = v + reinterpret_cast<const f32x4*>(d.res)[i];
            }
        }
        reinterpret_cast<f32x4*>(d.out)[i] = v;
    }
}

// Same reduction for launches whose consumer is a GroupNorm: one workgroup per (image, pixel tile), every thread owns
// one float4 channel column, so the per-(tile, channel) sum / sum of squares of the FINAL values come for free
// (layout of conv_epilogue's stats_out) and the low-resolution layers need no stand-alone statistics pass either.
static __global__ __launch_bounds__(256) void conv_splitk_reduce_stats_kernel(const ConvArgs p, int tpi) {
    __shared__ f32x4 red[2][256];
    const ddnm_conv_desc& d = p.d;
    const int c4n = d.Cout >> 2;
    const int rows = 256 / c4n, active = rows * c4n;
    const int hw = d.Ho * d.Wo, P = hw / tpi;
    const int b = blockIdx.x / tpi, t = blockIdx.x - b * tpi;
    const size_t slab4 = (size_t)d.B * hw * c4n;
    const int tid = threadIdx.x;
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, ss = {0.f, 0.f, 0.f, 0.f};
    if (tid < active) {
        const int c4 = tid % c4n, prow = tid / c4n, n = c4 * 4;
        f32x4 add = {0.f, 0.f, 0.f, 0.f};
        if (d.bias) add = add + *reinterpret_cast<const f32x4*>(d.bias + n);
        if (d.badd) add = add + *reinterpret_cast<const f32x4*>(d.badd + (size_t)b * d.badd_stride + n);
        for (int pp = prow; pp < P; pp += rows) {
            const int p2 = t * P + pp;
            const size_t i = ((size_t)b * hw + p2) * c4n + c4;
            f32x4 v = reinterpret_cast<const f32x4*>(p.ws)[i];
            for (int k = 1; k < p.ksplit; ++k) v = v + reinterpret_cast<const f32x4*>(p.ws)[i + k * slab4];
            v = v + add;
            if (d.res) {
                if (d.res_ups) {
                    const int oy = p2 / d.Wo, ox = p2 - oy * d.Wo;
                    v = v + *reinterpret_cast<const f32x4*>(d.res + (((size_t)b * (d.Ho >> 1) + (oy >> 1)) * (d.Wo >> 1) + (ox >> 1)) * d.Cout + n);
                } else {
                    v = v + reinterpret_cast<const f32x4*>(d.res)[i];
                }
            }
            reinterpret_cast<f32x4*>(d.out)[i] = v;
            s += v;
            ss += v * v;
        }
    }
    red[0][tid] = s;
    red[1][tid] = ss;
    __syncthreads();
    if (tid < c4n) {
        for (int r = 1; r < rows; ++r) { s += red[0][r * c4n + tid]; ss += red[1][r * c4n + tid]; }
        float* o = d.stats_out + ((size_t)blockIdx.x * d.Cout + tid * 4) * 2;
        f32x4 lo = {s.x, ss.x, s.y, ss.y}, hi = {s.z, ss.z, s.w, ss.w};
        reinterpret_cast<f32x4*>(o)[0] = lo;
        reinterpret_cast<f32x4*>(o)[1] = hi;
    }
}

// pixel tiles per image of the statistics-emitting reduction (0: this launch cannot emit them)
static inline int splitk_stats_tiles(const ddnm_conv_desc* d) {
    const int hw = d->Ho * d->Wo;
    if (d->out_nchw || d->Cout % 4 || d->Cout > 1024 || hw % 4) return 0;
    // enough workgroups to fill 256 CUs even at B = 1..4: <= 128 tiles per image, >= 4 pixels per tile
    int tpi = hw / 4 < 128 ? hw / 4 : 128;
    while (tpi > 1 && hw % tpi) --tpi;
    return tpi;
}

static inline int launch_splitk_reduce(const ConvArgs& p, hipStream_t s) {
    const ddnm_conv_desc& d = p.d;
    if (d.stats_out) {
        const int tpi = splitk_stats_tiles(&d);
        if (tpi <= 0) return DDNM_E_SHAPE;
        DDNM_LAUNCH(conv_splitk_reduce_stats_kernel, dim3(d.B * tpi), dim3(256), 0, s, p, tpi);
        return 0;
    }
    const size_t total4 = (size_t)d.B * d.Ho * d.Wo * d.Cout / 4;
    const unsigned g = (unsigned)((total4 + 255) / 256 < 2048 ? (total4 + 255) / 256 : 2048);
    DDNM_LAUNCH(conv_splitk_reduce_kernel, dim3(g), dim3(256), 0, s, p, total4);
    return 0;
}

