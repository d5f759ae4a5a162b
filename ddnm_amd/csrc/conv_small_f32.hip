// Output convolution of the celeba_hq `Model` (guided_diffusion/models.py:295-299,338-341): 3x3 / pad 1, 128 -> 3
// channels, GroupNorm + swish fused in front, fp32 NCHW result.  HBM-bound (the input is read once: 268 MB at B = 8;
// 3.6 GFLOP), so it runs on the vector ALU: on the MFMA tile kernel the 3 output channels are padded to a 32-wide N
// tile and the launch is bound by 10.7x wasted matrix work (317 us measured; this kernel: the HBM read).
//
//   workgroup = 256 threads = an 8 x 32 pixel patch, one output pixel per thread, Cout <= 4 accumulators each;
//   per 32-channel chunk the (8+2) x (32+2) halo goes through LDS with the GroupNorm affine + swish applied on the
//   way (zero padding AFTER the activation, like the reference); the 9 taps read it at shifted rows (pitch 36 floats:
//   conflict-free float4).  The weights are wave-uniform: they are read with SCALAR loads (s_load_dwordx4 through the
//   constant cache) and enter the FMAs as SGPR operands -- one LDS read per 12 FMAs instead of four.
#include "conv_common.h"

constexpr int CS_TH = 8, CS_TW = 32, CS_KC = 32, CS_PITCH = 36, CS_MAXCO = 4;
constexpr int CS_NP = (CS_TH + 2) * (CS_TW + 2);

template <int COUT>
__global__ __launch_bounds__(256) void conv3x3_small_cout_f32_kernel(const ddnm_conv_desc d, int tiles_x, int tiles_per_img) {
    __shared__ __attribute__((aligned(16))) float Hs[CS_NP * CS_PITCH];
    const int tid = threadIdx.x;
    const int img = blockIdx.x / tiles_per_img, t = blockIdx.x - img * tiles_per_img;
    const int ty0 = (t / tiles_x) * CS_TH, tx0 = (t % tiles_x) * CS_TW;
    const int Cin = d.C0, H = d.Hin, W = d.Win;
    const int py = tid >> 5, px = tid & 31;
    // halo loader: 340 pixels x 8 float4; thread -> (float4 column c4, pixels prow + 32*i), 11 slots
    constexpr int SLOTS = (CS_NP + 31) / 32;
    const int c4 = tid & 7, prow = tid >> 3;
    int goff[SLOTS];
#pragma unroll
    for (int i = 0; i < SLOTS; ++i) {
        const int r = prow + 32 * i;
        const int hy = r / (CS_TW + 2), hx = r - hy * (CS_TW + 2);
        const int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;
        goff[i] = (r < CS_NP && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W) ? ((img * H + iy) * W + ix) : -1;
    }
    f32x4 st[SLOTS];
    auto fetch = [&](int cb) {
#pragma unroll
        for (int i = 0; i < SLOTS; ++i)
            st[i] = goff[i] >= 0 ? *reinterpret_cast<const f32x4*>(d.src0 + (size_t)goff[i] * Cin + cb + c4 * 4)
                                 : f32x4{0.f, 0.f, 0.f, 0.f};
    };
    float acc[COUT];
#pragma unroll
    for (int o = 0; o < COUT; ++o) acc[o] = 0.f;
    fetch(0);
    for (int cb = 0; cb < Cin; cb += CS_KC) {
        f32x4 gsc = {1.f, 1.f, 1.f, 1.f}, gsh = {0.f, 0.f, 0.f, 0.f};
        if (d.gn_scale) {
            gsc = *reinterpret_cast<const f32x4*>(d.gn_scale + (size_t)img * Cin + cb + c4 * 4);
            gsh = *reinterpret_cast<const f32x4*>(d.gn_shift + (size_t)img * Cin + cb + c4 * 4);
        }
        __syncthreads();                                   // previous chunk's reads are done
#pragma unroll
        for (int i = 0; i < SLOTS; ++i) {
            const int r = prow + 32 * i;
            if (r < CS_NP) {
                f32x4 v = st[i];
                if (d.gn_scale && goff[i] >= 0) v = gn_act(v, gsc, gsh, d.gn_silu);
                *reinterpret_cast<f32x4*>(&Hs[r * CS_PITCH + c4 * 4]) = v;
            }
        }
        __syncthreads();
        if (cb + CS_KC < Cin) fetch(cb + CS_KC);           // next chunk's halo is in flight under this chunk's FMAs
        const float* __restrict__ wchunk = d.weight + cb;  // W[o][tap][cb + c]: wave-uniform -> scalar loads
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const float* hp = &Hs[((py + tap / 3) * (CS_TW + 2) + px + tap % 3) * CS_PITCH];
            // 3 x 16 weights at a time (wide scalar loads, ONE wait: 48 SGPRs), then 4 LDS reads x 12 FMAs; a scalar load
            // per float4 with its own wait serialised the loop on the scalar-cache latency (271 us -> HBM-bound)
#pragma unroll
            for (int kh2 = 0; kh2 < 2; ++kh2) {
                float w[COUT][16];
#pragma unroll
                for (int o = 0; o < COUT; ++o)
#pragma unroll
                    for (int c = 0; c < 16; ++c) w[o][c] = wchunk[((size_t)o * 9 + tap) * Cin + kh2 * 16 + c];
#pragma unroll
                for (int k4 = 0; k4 < 4; ++k4) {
                    const f32x4 a = *reinterpret_cast<const f32x4*>(hp + kh2 * 16 + k4 * 4);
#pragma unroll
                    for (int o = 0; o < COUT; ++o) {
                        acc[o] = fmaf(a.x, w[o][k4 * 4 + 0], acc[o]);
                        acc[o] = fmaf(a.y, w[o][k4 * 4 + 1], acc[o]);
                        acc[o] = fmaf(a.z, w[o][k4 * 4 + 2], acc[o]);
                        acc[o] = fmaf(a.w, w[o][k4 * 4 + 3], acc[o]);
                    }
                }
            }
        }
    }
    const int oy = ty0 + py, ox = tx0 + px;
#pragma unroll
    for (int o = 0; o < COUT; ++o)
        d.out[(((size_t)img * COUT + o) * H + oy) * W + ox] = acc[o] + (d.bias ? d.bias[o] : 0.f);
}

static bool small_cout_ok(const ddnm_conv_desc* d) {
    return d->ksize == 3 && d->stride == 1 && d->pad == 1 && d->Ho == d->Hin && d->Wo == d->Win && d->Cout >= 1 &&
           d->Cout <= CS_MAXCO && d->out_nchw && d->C1 == 0 && !d->src1 && d->C0 % CS_KC == 0 && !d->ups && !d->res &&
           !d->badd && !d->skip0 && !d->stats_out && d->Win % CS_TW == 0 && d->Hin % CS_TH == 0;
}

extern "C" int ddnm_conv3x3_small_cout_f32_supported(const ddnm_conv_desc* d) { return d && small_cout_ok(d) ? 1 : 0; }

extern "C" int ddnm_conv3x3_small_cout_f32(const ddnm_conv_desc* d, void* stream) {
    if (!d || !d->src0 || !d->weight || !d->out || d->B <= 0) return DDNM_E_BADARG;
    if (d->gn_scale && !d->gn_shift) return DDNM_E_BADARG;
    if (!small_cout_ok(d)) return DDNM_E_SHAPE;
    const int tiles_x = d->Win / CS_TW, tpi = tiles_x * (d->Hin / CS_TH);
    const dim3 grid(d->B * tpi);
    hipStream_t s = (hipStream_t)stream;
    switch (d->Cout) {
        case 1: DDNM_LAUNCH(conv3x3_small_cout_f32_kernel<1>, grid, dim3(256), 0, s, *d, tiles_x, tpi); break;
        case 2: DDNM_LAUNCH(conv3x3_small_cout_f32_kernel<2>, grid, dim3(256), 0, s, *d, tiles_x, tpi); break;
        case 3: DDNM_LAUNCH(conv3x3_small_cout_f32_kernel<3>, grid, dim3(256), 0, s, *d, tiles_x, tpi); break;
        default: DDNM_LAUNCH(conv3x3_small_cout_f32_kernel<4>, grid, dim3(256), 0, s, *d, tiles_x, tpi); break;
    }
    return 0;
}
