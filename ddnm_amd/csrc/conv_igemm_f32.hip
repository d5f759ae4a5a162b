// Implicit-GEMM convolution on v_mfma_f32_32x32x2_f32 (exact fp32), gfx950.
//
//   GEMM view: M = B*Ho*Wo output pixels, N = Cout, K = taps * Cin.
//   Workgroup = 256 threads = 4 waves arranged WM x WN; each wave owns MT x NT tiles of 32x32.
//   K loop: (channel chunk of 32) x (filter tap).  Per step the A tile [BM][32] is gathered
//   from the NHWC activation(s) -- through the optional nearest-x2 upsample, channel concat
//   of two sources and GroupNorm-affine(+swish) prologue -- and the B tile [BN][32] from the
//   (O, ky, kx, I)-packed weights; both land k-contiguous in LDS (row pitch 36 floats), so
//   each lane fetches its MFMA operands with one conflict-free ds_read_b128 per 4 MFMAs.
//   Global loads for step i+1 are issued before the MFMAs of step i (register staging).
//   Epilogue: + bias + per-sample addend (temb projection) + residual, NHWC or NCHW store.
//
// Replaces: nn.Conv2d call sites of guided_diffusion/models.py (see include/ddnm_hip.h).
#include "common.h"

struct ConvArgs {
    ddnm_conv_desc d;
    int Cin, ntaps, Hs, Ws;
    int m_tiles, n_tiles;
    int tiles_x, tiles_per_img;  // 2-D tiling of the output image (tiles_x == 0: flat strips)
    int TW;                      // tile width in pixels (power of two) in 2-D mode
};

template <int WM, int WN, int MT, int NT>
__global__ __launch_bounds__(256) void conv_igemm_f32_kernel(const ConvArgs p) {
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
    constexpr int AR = BM / 32, BR = BN / 32;  // rows per thread for the A / B tile copies
    __shared__ __attribute__((aligned(16))) float As[BM * LDT];
    __shared__ __attribute__((aligned(16))) float Bs[BN * LDT];

    const ddnm_conv_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tile_id = xcd_swizzle(blockIdx.x, gridDim.x);
    const int n_tile = tile_id % p.n_tiles, m_tile = tile_id / p.n_tiles;
    const int HWo = d.Ho * d.Wo;

    // pixel coordinates of local row r of this M tile
    const int img = (m_tile * BM) / HWo;           // one image per tile (host guarantees HWo % BM == 0)
    const int t_in_img = m_tile - img * (HWo / BM);
    auto pixel_of = [&](int r, int& oy, int& ox) {
        if (p.tiles_x) {
            const int ty = t_in_img / p.tiles_x, tx = t_in_img - ty * p.tiles_x;
            const int th = BM / p.TW;
            oy = ty * th + r / p.TW;
            ox = tx * p.TW + (r & (p.TW - 1));
        } else {
            const int pix = t_in_img * BM + r;
            oy = pix / d.Wo;
            ox = pix - oy * d.Wo;
        }
    };

    // ---- loader mapping: thread -> (float4 column c4, rows row0 + 32*i)
    const int c4 = tid & 7, row0 = tid >> 3;
    int iy0[AR], ix0[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        int oy, ox;
        pixel_of(row0 + 32 * i, oy, ox);
        iy0[i] = oy * d.stride - d.pad;
        ix0[i] = ox * d.stride - d.pad;
    }
    const float* wbase = d.weight + (size_t)(n_tile * BN + row0) * p.ntaps * p.Cin + c4 * 4;

    f32x4 a_st[AR], b_st[BR];
    f32x4 gsc = {1.f, 1.f, 1.f, 1.f}, gsh = {0.f, 0.f, 0.f, 0.f};
    unsigned a_valid = 0;
    const bool has_gn = d.gn_scale != nullptr;

    auto prefetch = [&](int it) {
        const int chunk = it / p.ntaps, tap = it - chunk * p.ntaps;
        const int ky = tap / d.ksize, kx = tap - ky * d.ksize;
        const int cb = chunk * KC;
        const float* src;
        int cs, coff;
        if (cb < d.C0) { src = d.src0; cs = d.C0; coff = cb; }
        else { src = d.src1; cs = d.C1; coff = cb - d.C0; }
        a_valid = 0;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            const int iy = iy0[i] + ky, ix = ix0[i] + kx;
            const bool ok = (unsigned)iy < (unsigned)d.Hin && (unsigned)ix < (unsigned)d.Win;
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (ok) {
                const int sy = d.ups ? (iy >> 1) : iy, sx = d.ups ? (ix >> 1) : ix;
                v = *reinterpret_cast<const f32x4*>(src + ((size_t)(img * p.Hs + sy) * p.Ws + sx) * cs + coff + c4 * 4);
                a_valid |= 1u << i;
            }
            a_st[i] = v;
        }
        if (has_gn && tap == 0) {
            gsc = *reinterpret_cast<const f32x4*>(d.gn_scale + (size_t)img * p.Cin + cb + c4 * 4);
            gsh = *reinterpret_cast<const f32x4*>(d.gn_shift + (size_t)img * p.Cin + cb + c4 * 4);
        }
        const float* wp = wbase + (size_t)tap * p.Cin + cb;
#pragma unroll
        for (int i = 0; i < BR; ++i)
            b_st[i] = *reinterpret_cast<const f32x4*>(wp + (size_t)(32 * i) * p.ntaps * p.Cin);
    };

    auto stage_to_lds = [&]() {
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            f32x4 v = a_st[i];
            if (has_gn) {
                if (a_valid & (1u << i)) {
                    v = v * gsc + gsh;
                    if (d.gn_silu) { v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w); }
                }
            }
            *reinterpret_cast<f32x4*>(&As[(row0 + 32 * i) * LDT + c4 * 4]) = v;
        }
#pragma unroll
        for (int i = 0; i < BR; ++i)
            *reinterpret_cast<f32x4*>(&Bs[(row0 + 32 * i) * LDT + c4 * 4]) = b_st[i];
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int n_iter = (p.Cin / KC) * p.ntaps;
    const int frag_off = (lane & 31) * LDT + (lane >> 5) * 4;
    const float* a_frag = As + (wm * MT * 32) * LDT + frag_off;
    const float* b_frag = Bs + (wn * NT * 32) * LDT + frag_off;

    prefetch(0);
    for (int it = 0; it < n_iter; ++it) {
        __syncthreads();           // all waves finished reading the previous tile
        stage_to_lds();
        __syncthreads();
        if (it + 1 < n_iter) prefetch(it + 1);
#pragma unroll
        for (int kk = 0; kk < KC / 8; ++kk) {
            f32x4 a[MT], b[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const f32x4*>(a_frag + i * 32 * LDT + kk * 8);
#pragma unroll
            for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const f32x4*>(b_frag + j * 32 * LDT + kk * 8);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = mfma_k8(a[i], b[j], acc[i][j]);
        }
    }

    // ---- epilogue.  C/D map of 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8*(reg >> 2) + 4*(lane >> 5)
    const int ncol = lane & 31, rsel = 4 * (lane >> 5);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = n_tile * BN + (wn * NT + j) * 32 + ncol;
        if (n >= d.Cout) continue;
        float add = d.bias ? d.bias[n] : 0.f;
        if (d.badd) add += d.badd[(size_t)img * d.badd_stride + n];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = (wm * MT + i) * 32 + (r & 3) + 8 * (r >> 2) + rsel;
                int oy, ox;
                pixel_of(row, oy, ox);
                float v = acc[i][j][r] + add;
                if (d.out_nchw) {
                    d.out[((size_t)(img * d.Cout + n) * d.Ho + oy) * d.Wo + ox] = v;
                } else {
                    const size_t o = ((size_t)(img * d.Ho + oy) * d.Wo + ox) * d.Cout + n;
                    if (d.res) v += d.res[o];
                    d.out[o] = v;
                }
            }
        }
    }
}

static int pick_tile(const ddnm_conv_desc* d) {
    if (d->tile) return d->tile;
    const int HWo = d->Ho * d->Wo;
    if (d->Cout <= 32) return (HWo % 128 == 0) ? 3 : 2;
    // 128x128 when it still yields a full wave of workgroups, else 64x64
    if (HWo % 128 == 0 && d->Cout % 128 == 0) {
        const long tiles = (long)d->B * (HWo / 128) * (d->Cout / 128);
        if (tiles >= 256) return 1;
    }
    return 2;
}

extern "C" int ddnm_conv2d_f32_tile_n(const ddnm_conv_desc* d) {
    const int t = pick_tile(d);
    return t == 1 ? 128 : (t == 2 ? 64 : 32);
}

extern "C" int ddnm_conv2d_f32(const ddnm_conv_desc* d, void* stream) {
    if (!d || !d->src0 || !d->weight || !d->out) return DDNM_E_BADARG;
    if (d->B <= 0 || d->Cout <= 0 || d->Ho <= 0 || d->Wo <= 0) return DDNM_E_BADARG;
    if (d->C0 <= 0 || d->C0 % KC || d->C1 % KC || (d->C1 > 0 && !d->src1)) return DDNM_E_SHAPE;
    if ((d->ksize != 1 && d->ksize != 3) || (d->stride != 1 && d->stride != 2)) return DDNM_E_SHAPE;
    if (d->gn_scale && !d->gn_shift) return DDNM_E_BADARG;
    if (d->ups && ((d->Hin | d->Win) & 1)) return DDNM_E_SHAPE;
    if (d->out_nchw && d->res) return DDNM_E_SHAPE;
    ConvArgs p;
    p.d = *d;
    p.Cin = d->C0 + d->C1;
    p.ntaps = d->ksize * d->ksize;
    p.Hs = d->ups ? d->Hin / 2 : d->Hin;
    p.Ws = d->ups ? d->Win / 2 : d->Win;
    const int tile = pick_tile(d);
    const int BM = tile == 2 ? 64 : 128, BN = tile == 1 ? 128 : (tile == 2 ? 64 : 32);
    const int HWo = d->Ho * d->Wo;
    if (HWo % BM) return DDNM_E_SHAPE;
    p.m_tiles = d->B * (HWo / BM);
    p.n_tiles = (d->Cout + BN - 1) / BN;
    // 2-D output tiles (16 wide) keep the 3x3 halo of a tile compact in L2/L1
    p.TW = 16;
    const int TH = BM / p.TW;
    if (d->Wo % p.TW == 0 && d->Ho % TH == 0) {
        p.tiles_x = d->Wo / p.TW;
    } else {
        p.tiles_x = 0;
    }
    p.tiles_per_img = HWo / BM;
    dim3 grid(p.m_tiles * p.n_tiles), block(256);
    hipStream_t s = (hipStream_t)stream;
    switch (tile) {
        case 1: hipLaunchKernelGGL((conv_igemm_f32_kernel<2, 2, 2, 2>), grid, block, 0, s, p); break;
        case 2: hipLaunchKernelGGL((conv_igemm_f32_kernel<2, 2, 1, 1>), grid, block, 0, s, p); break;
        case 3: hipLaunchKernelGGL((conv_igemm_f32_kernel<4, 1, 1, 1>), grid, block, 0, s, p); break;
        default: return DDNM_E_BADARG;
    }
    DDNM_LAUNCH_CHECK();
    return 0;
}
