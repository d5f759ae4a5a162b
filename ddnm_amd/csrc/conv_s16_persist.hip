// Persistent form of the split-fp16 3x3 convolution (round 6): one workgroup per CU walks SEVERAL output tiles.
//
// Why (profiles/r06_mfma_ceiling.md, tools/r06/tile_phases.py): in the one-tile-per-workgroup kernel
// (conv_igemm_f16.hip, <4,2,2,2,..,SPLIT>) a 256 x 128 tile of a 128 -> 128 @256^2 launch spends 47.6 us in its tap loop and
// 11.8 us outside it -- 4.6 us from workgroup entry to the first MFMA (index arithmetic, the first halo from HBM, its
// GroupNorm / split staging), 6.8 us in an epilogue whose stores all 256 CUs issue in lock-step (33 MB burst = the HBM write
// rate), 0.3 us of workgroup turnover -- and with 151 KB of LDS per workgroup nothing else runs on the CU meanwhile: 20 % of
// the launch with an idle matrix pipe.  Here the chunk stream simply continues across tiles:
//   * the LAST chunk of a tile prefetches the NEXT tile's first halo (taps 0 / 3), stages it into the free halo buffer
//     (taps 3 / 6) and requests the next tile's first two weight tiles (taps 7 / 8): the next tile's first MFMA follows the
//     last one of this tile after ~1 us of register work (scale + bias, statistics, accumulator reset);
//   * the epilogue is DEFERRED: the finished tile's 64 values per lane move to a second register set and leave as buffer
//     stores spread over taps 0..6 of the next tile's first chunk (9-10 per tap and wave: ~0.7 TB/s chip-wide instead of
//     a burst), its residual tile arrives the same way during the LAST chunk (loads into that register set), bias /
//     per-sample addend with it; the GroupNorm partials cross the waves through LDS under the next chunk's first barrier.
// Every extra request is part of a branch-free, compile-time-known request stream, so the main loop keeps its counted
// `s_waitcnt vmcnt(N)` (VMEM retires in order; N = requests issued behind the awaited weight tile, p_wait() below;
// tests/test_isa_waits.py replays the stream from the ISA).  Arithmetic, summation order and statistics are those of the
// one-tile kernel: results are bit-identical (tests/test_gpu_s16.py).
//
// Replaces: the same nn.Conv2d 3x3 + Normalize / swish prologue + concat + nearest x2 + temb addend + residual of the
// celeba `Model` (/root/reference/guided_diffusion/models.py:36-134) as ddnm_conv3x3_s16_f32, for launches of >= 2 tiles per
// CU without a fused shortcut (the 256^2 / 128^2 levels: 62 % of the forward's kernel time).
#include <type_traits>

#include "conv_common.h"

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

namespace {
constexpr int P_WM = 4, P_WN = 2, P_MT = 2, P_NT = 2;
constexpr int P_NTHREADS = 512, P_BN = 128;      // block tile 256 pixels (8 x 32) x 128 channels
constexpr int P_LDH = 72, P_KCH = 32;                 // LDS halo row pitch in halfs ([hi 32 | lo 32] + 8), channels per chunk
constexpr int P_MAXH = 340, P_HWD = 34;               // 8 x 32 output patch, halo 10 x 34
constexpr int P_HCOLS = 8, P_HRPP = P_NTHREADS / P_HCOLS, P_HR = (P_MAXH + P_HRPP - 1) / P_HRPP, P_HSPLIT = (P_HR + 1) / 2;
constexpr int P_BR = P_BN / (P_NTHREADS / 8);         // LDS-DMA instructions per wave and weight tile
constexpr int P_NWB = 3, P_WTILE = P_BN * 128;
constexpr int P_HBYTES = 2 * P_MAXH * P_LDH * 2;
constexpr int P_NOUT = P_MT * P_NT * 16;              // output values per lane and tile
enum { K_MID = 0, K_LAST = 1, K_FIRST = 2 };

// Requests a wave issues at the END of a tap's issue block (behind the weight-tile DMA and the halo loads) on top of the
// one-tile kernel's stream: FIRST chunk of a tile = the previous tile's deferred stores (+ 1 statistics store at tap 0),
// LAST chunk = bias / per-sample addend (4 at tap 0), the next tile's operand bound (1 at tap 0, ASCALE instances) and, with
// a residual, its 64 loads.  Nothing at taps 7 / 8, so a chunk's table does not depend on its neighbours.
constexpr int p_spread(int tap) { return tap == 0 ? 10 : (tap <= 6 ? 9 : 0); }
constexpr int p_first_of(int tap) { return tap == 0 ? 0 : (tap <= 7 ? 10 + 9 * (tap - 1) : P_NOUT); }
static_assert(p_first_of(7) == P_NOUT, "64 values over taps 0..6");
constexpr int p_extra(int kind, int tap, bool has_res, bool ascale) {
    return kind == K_FIRST ? p_spread(tap) + (tap == 0 ? 1 : 0)
                           : (kind == K_LAST ? (tap == 0 ? 4 + (ascale ? 1 : 0) : 0) + (has_res ? p_spread(tap) : 0) : 0);
}
constexpr int p_base(int tap) {
    return (tap == 1 || tap == 2) ? P_BR + P_HSPLIT + 2 : ((tap == 4 || tap == 5) ? P_BR + P_HR - P_HSPLIT : P_BR);
}
// vmcnt immediate of the wait in front of tap `tap`: everything issued behind W(tap) (requested first thing at tap - 2)
constexpr int p_wait(int kind, int tap, bool has_res, bool ascale) {
    return p_base(tap) + (tap >= 1 ? p_extra(kind, tap - 1, has_res, ascale) : 0) + (tap >= 2 ? p_extra(kind, tap - 2, has_res, ascale) : 0);
}
static_assert(p_wait(K_FIRST, 2, true, true) < 64 && p_wait(K_LAST, 2, true, true) < 64, "vmcnt is a 6-bit field");

template <int V>
using ic = std::integral_constant<int, V>;

}  // namespace

// HAS_SKIP: the fused 1x1 shortcut of a ResnetBlock (models.py:109,128-132) as extra K chunks at the centre tap, between a
// tile's LAST chunk and its register epilogue -- in the halo / weight buffers the chunk stream is not using at that moment
// (the other halo buffer holds the next tile's first chunk, weight buffers 0 / 1 its first two tiles).
template <bool ASCALE, bool HAS_RES, bool HAS_SKIP = false>
__global__ __launch_bounds__(P_NTHREADS, 1) void conv3x3_s16_persist_kernel(const ConvArgs p, const int total) {
    static_assert(!HAS_SKIP || (ASCALE && !HAS_RES), "a fused shortcut reads a raw operand (bound required) and replaces the residual");
    constexpr int WM = P_WM, WN = P_WN, MT = P_MT, NT = P_NT, BN = P_BN, HR = P_HR, HSPLIT = P_HSPLIT, LDH = P_LDH, KCH = P_KCH;
    constexpr int MAXH = P_MAXH, HWd = P_HWD, NWB = P_NWB, WTILE = P_WTILE, BR = P_BR, NTHREADS = P_NTHREADS;
    __shared__ __attribute__((aligned(1024))) char lds_all[NWB * WTILE + P_HBYTES + WM * BN * 2 * 4];
    char* const Bs = lds_all;
    _Float16* const Hs = reinterpret_cast<_Float16*>(lds_all + NWB * WTILE);
    float* const stat_lds = reinterpret_cast<float*>(lds_all + NWB * WTILE + P_HBYTES);

    const ddnm_conv_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int G = gridDim.x;
    const int per_img = (d.Ho * d.Wo) >> 8;
    const int Cout = d.Cout, Wo = d.Wo;

    // ---- tile bookkeeping (wave-uniform): virtual block v -> the tile the one-tile launch would give block v
    int t_img, t_ty0, t_tx0, t_ntile, t_mtile;
    auto tile_of = [&](int v) {
        const int tile_id = xcd_swizzle(v, total);
        t_ntile = tile_id % p.n_tiles;
        t_mtile = tile_id / p.n_tiles;
        t_img = t_mtile / per_img;
        const int t = t_mtile - t_img * per_img;
        const int ty = t / p.tiles_x;
        t_ty0 = ty * 8;
        t_tx0 = (t - ty * p.tiles_x) * 32;
    };

    // ---- halo loader mapping: thread -> (16-byte column hc of 8, halo rows prow + 64 i); hoff[] = source pixel per row slot
    const int hc = tid % P_HCOLS, prow = tid / P_HCOLS;
    int hoff[HR];
    auto set_hoff = [&]() {
#pragma unroll
        for (int i = 0; i < HR; ++i) {
            const int row = prow + P_HRPP * i;
            const int hy = row / HWd, hx = row - hy * HWd;
            const int iy = t_ty0 - 1 + hy, ix = t_tx0 - 1 + hx;
            const bool ok = row < MAXH && (unsigned)iy < (unsigned)d.Hin && (unsigned)ix < (unsigned)d.Win;
            const int sy = d.ups ? (iy >> 1) : iy, sx = d.ups ? (ix >> 1) : ix;
            hoff[i] = ok ? (t_img * p.Hs + sy) * p.Ws + sx : -1;
        }
    };

    // ---- weight tiles by LDS-DMA (the one-tile kernel's layout: lane-linear rows, XOR swizzle on the SOURCE address)
    const int lrow = lane >> 3, lpiece = lane & 7;
    const int wswz = (((wave & 1) << 2) | (lrow >> 1));
    const unsigned w_rowlen = 9u * (unsigned)p.Cin * 4u;
    const __amdgpu_buffer_rsrc_t r_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(reinterpret_cast<const void*>(d.weight)), 0, (unsigned)p.n_tiles * BN * w_rowlen, 0x00020000);
    const unsigned w_lane = (unsigned)(wave * 8 + lrow) * w_rowlen + (unsigned)((lpiece ^ wswz) * 16);
    unsigned w_soff = 0;                              // n_tile * BN * w_rowlen of the tile whose weights are being requested
    auto issue_w = [&](int chunk, int tap, int buf) {
        char* dst = Bs + buf * WTILE + wave * 1024;
        const unsigned so = w_soff + ((unsigned)tap * p.Cin + (unsigned)chunk * KCH) * 4u;
#pragma unroll
        for (int j = 0; j < BR; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w, (__attribute__((address_space(3))) void*)(dst + j * (NTHREADS / 64) * 1024), 16,
                                                     w_lane, so + (unsigned)j * (NTHREADS / 8) * w_rowlen, 0, 0);
    };

    const int nchunks = p.Cin / KCH;
    uint4 h_st[HR];
    f32x4 gsc = {1.f, 1.f, 1.f, 1.f}, gsh = {0.f, 0.f, 0.f, 0.f};
    const bool has_gn = d.gn_scale != nullptr;
    constexpr unsigned HOOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t r_s0 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(reinterpret_cast<const void*>(d.src0)), 0, (unsigned)d.B * p.Hs * p.Ws * d.C0 * 4, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_s1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(reinterpret_cast<const void*>(d.C1 > 0 ? d.src1 : d.src0)), 0,
        (unsigned)d.B * p.Hs * p.Ws * (d.C1 > 0 ? d.C1 : d.C0) * 4, 0x00020000);
    // Operand-range guard (ASCALE: raw operands; conv_common.h::s16_operand_scale restated for a request stream that must
    // stay countable): the DDNM_AMAX_N bound words of an image are ONE buffer load per wave (lane l < 32 fetches word l; the
    // one-tile kernel's scalar loads would become vector loads here -- the kernel stores, so nothing is provably
    // unclobbered -- in a number the compiler chooses), the maximum is a wave reduction on the bit patterns.
    const __amdgpu_buffer_rsrc_t r_amax = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(reinterpret_cast<const void*>(d.amax_in)), 0, d.amax_in ? (unsigned)d.B * DDNM_AMAX_N * 4u : 0u, 0x00020000);
    auto amax_request = [&](int img, bool live) {
        return __builtin_amdgcn_raw_buffer_load_b32(r_amax, (lane < DDNM_AMAX_N && live) ? (unsigned)lane * 4u : HOOB, (unsigned)img * DDNM_AMAX_N * 4u, 0);
    };
    auto amax_scales = [&](unsigned m, bool down_only, float& scale, float& inv_scale) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) {
            const unsigned t = (unsigned)__shfl_xor((int)m, o);
            m = t > m ? t : m;
        }
        m = __builtin_amdgcn_readfirstlane(m);
        int e = (int)((m >> 23) & 0xffu);
        e = e < 47 ? 47 : (e > 207 ? 207 : e);
        int k = 14 - (e - 127);
        if (down_only && k > 0) k = 0;
        scale = __uint_as_float((unsigned)(127 + k) << 23);
        inv_scale = __uint_as_float((unsigned)(127 - k) << 23);
    };
    unsigned amax_bits = 0;                           // the next tile's bound words, requested at the LAST chunk's tap 0
    int gn_img = 0;                                   // image whose GroupNorm vectors the next prefetch fetches
    float ascale_stage = 1.f;                         // operand scale of the halo being staged (ASCALE)
    auto prefetch_halo_part = [&](int chunk, int i0, int i1, bool live = true) {
        const int cb = chunk * KCH;
        const bool first = cb < d.C0;
        const unsigned cs = first ? d.C0 : d.C1, coff = first ? cb : cb - d.C0;
        const __amdgpu_buffer_rsrc_t r_s = first ? r_s0 : r_s1;
#pragma unroll
        for (int i = 0; i < HR; ++i) {
            if (i < i0 || i >= i1) continue;
            const unsigned vo = (hoff[i] >= 0 && live) ? ((unsigned)hoff[i] * cs + hc * 4) * 4 : HOOB;
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r_s, vo, coff * 4, 0);
            h_st[i] = uint4{v.x, v.y, v.z, v.w};
        }
        if (i0 == 0) {
            const float* gb = has_gn ? d.gn_scale + (size_t)gn_img * p.Cin + hc * 4 + cb : reinterpret_cast<const float*>(d.weight);
            const float* hb2 = has_gn ? d.gn_shift + (size_t)gn_img * p.Cin + hc * 4 + cb : reinterpret_cast<const float*>(d.weight);
            gsc = *reinterpret_cast<const f32x4*>(gb);
            gsh = *reinterpret_cast<const f32x4*>(hb2);
        }
    };
    auto stage_halo_part = [&](int hbuf, int i0, int i1) {
#pragma unroll
        for (int i = 0; i < HR; ++i) {
            if (i < i0 || i >= i1) continue;
            const int row = prow + P_HRPP * i;
            if (row < MAXH) {
                _Float16* dst = &Hs[hbuf * MAXH * LDH + row * LDH + hc * 4];
                f32x4 v = __builtin_bit_cast(f32x4, h_st[i]);
                if (has_gn && hoff[i] >= 0) v = gn_act(v, gsc, gsh, d.gn_silu);
                if constexpr (ASCALE) split_store(dst, v, ascale_stage);
                else split_store(dst, v);
            }
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    int a_off[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = (wm * MT + i) * 32 + (lane & 31);
        a_off[i] = ((m >> 5) * HWd + (m & 31)) * LDH + (lane >> 5) * 8;
    }
    const int b_frag = ((wn * NT * 32 + (lane & 31)) * 128) + ((((lane >> 5) ^ (((lane & 31) >> 1) & 7))) << 4);
    auto mfma_tap = [&](int tap, int buf, int hbuf) {
        const int ky = tap / 3, kx = tap - 3 * ky;
        const int tap_off = (ky * HWd + kx) * LDH + hbuf * MAXH * LDH;
        const char* bf = Bs + buf * WTILE;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) {
                ah[i] = *reinterpret_cast<const half8*>(Hs + a_off[i] + tap_off + ks * 16);
                al[i] = *reinterpret_cast<const half8*>(Hs + a_off[i] + tap_off + ks * 16 + 32);
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                bh[j] = *reinterpret_cast<const half8*>(bf + ((b_frag ^ (ks << 5)) + j * 32 * 128));
                bl[j] = *reinterpret_cast<const half8*>(bf + ((b_frag ^ ((ks + 2) << 5)) + j * 32 * 128));
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    };

    // ---- output side: value k = (i * NT + j) * 16 + r of a lane is row (wm*MT + i)*32 + xr(r) + 4 (lane >> 5) of the tile
    // (TW = 32: tile row wm*MT + i, x = xr(r) + 4 (lane >> 5)), channel (wn*NT + j)*32 + (lane & 31).  Lane part in a VGPR,
    // tile / (i, j, r) part in the scalar offset of the buffer instruction.
    const int ncol = lane & 31, rsel = 4 * (lane >> 5);
    const unsigned out_bytes = (unsigned)d.B * d.Ho * Wo * Cout * 4u;
    const __amdgpu_buffer_rsrc_t r_out = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(d.out), 0, out_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_res = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(reinterpret_cast<const void*>(d.res)), 0, d.res ? out_bytes : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_bias = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(reinterpret_cast<const void*>(d.bias)), 0, d.bias ? (unsigned)Cout * 4u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_badd = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(reinterpret_cast<const void*>(d.badd)), 0, d.badd ? (unsigned)d.B * d.badd_stride * 4u : 0u, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_stats = __builtin_amdgcn_make_buffer_rsrc(
        reinterpret_cast<void*>(d.stats_out), 0, d.stats_out ? (unsigned)p.m_tiles * Cout * 8u : 0u, 0x00020000);
    const unsigned o_lane = (unsigned)(((wm * MT * Wo + rsel) * Cout + wn * NT * 32 + ncol) * 4);
    auto o_soff = [&](unsigned base, int k) {        // scalar byte offset of value k relative to the tile base
        const int i = k >> 5, j = (k >> 4) & 1, r = k & 15;
        return base + (unsigned)(((i * Wo + (r & 3) + 8 * (r >> 2)) * Cout + j * 32) * 4);
    };
    auto tile_base = [&]() { return (unsigned)((((t_img * d.Ho + t_ty0) * Wo + t_tx0) * Cout + t_ntile * BN) * 4); };
    const unsigned c_lane = (unsigned)((wn * NT * 32 + ncol) * 4);      // channel of (j = 0) in bias / badd
    const unsigned st_lane = tid < 2 * BN ? (unsigned)(tid * 4) : HOOB;  // statistics: thread -> (channel tid >> 1, sum / sum of squares)

    float outv[P_NOUT];                               // residual tile (LAST chunk) -> finished values (deferred stores)
    float addv[NT][2];                                // bias, per-sample addend of the lane's channels
#pragma unroll
    for (int k = 0; k < P_NOUT; ++k) outv[k] = 0.f;
#pragma unroll
    for (int j = 0; j < NT; ++j) addv[j][0] = addv[j][1] = 0.f;
    unsigned pend_base = 0, pend_stats = 0;           // deferred tile: output base, statistics row
    unsigned cur_base = 0, cur_stats = 0;
    float epi_cur = d.acc_scale;

    // tile end, register work only: scale, bias, residual; per-channel partial sums for the consumer's GroupNorm (the
    // summation order of conv_epilogue: per channel column i-major over the wave's rows, halves combined, waves in order)
    auto finalize = [&]() {
        float cs[NT], cq[NT];
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            cs[j] = cq[j] = 0.f;
            const float add = addv[j][0];            // bias + per-sample addend (summed at the LAST chunk's tap 4)
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int k = (i * NT + j) * 16 + r;
                    const float v = acc[i][j][r] * epi_cur + add + (HAS_RES ? outv[k] : 0.f);
                    outv[k] = v;
                    cs[j] += v;
                    cq[j] = __builtin_fmaf(v, v, cq[j]);
                    acc[i][j][r] = 0.f;
                }
            cs[j] += __shfl_xor(cs[j], 32);
            cq[j] += __shfl_xor(cq[j], 32);
        }
        if (lane < 32) {
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                const int c = (wn * NT + j) * 32 + lane;
                stat_lds[(wm * BN + c) * 2 + 0] = cs[j];
                stat_lds[(wm * BN + c) * 2 + 1] = cq[j];
            }
        }
        pend_base = cur_base;
        pend_stats = cur_stats;
    };
    auto store_stats = [&]() {                        // behind a barrier that follows finalize(): one store per wave
        const int c = (tid >> 1) & (BN - 1), which = tid & 1;
        float a = 0.f;
#pragma unroll
        for (int w = 0; w < WM; ++w) a += stat_lds[(w * BN + c) * 2 + which];
        __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(a), r_stats, st_lane, pend_stats, 0);
    };

    // ---- first tile: the one-tile kernel's prologue
    int v = blockIdx.x;
    tile_of(v);
    set_hoff();
    gn_img = t_img;
    w_soff = (unsigned)t_ntile * BN * w_rowlen;
    cur_base = tile_base();
    cur_stats = (unsigned)((t_mtile * Cout + t_ntile * BN) * 8);
    if constexpr (ASCALE) {
        float inv;
        amax_scales(amax_request(t_img, true), has_gn, ascale_stage, inv);       // nothing else is in flight yet
        epi_cur = d.acc_scale * inv;
    }
    prefetch_halo_part(0, 0, HR);
    issue_w(0, 0, 0);
    issue_w(0, 1, 1);
    stage_halo_part(0, 0, HR);
    if (wave >= WM * WN / 2) __builtin_amdgcn_s_setprio(1);       // one wave of each SIMD pair at raised priority (conv_igemm_f16.hip)

    int hb = 0, n_img_next = 0;
    bool have_next = false;
    float epi_next = epi_cur;
    unsigned next_base = 0, next_stats = 0, next_wsoff = 0;

    // one tap of chunk `chunk` of the current tile; KIND / TAP are compile-time (the vmcnt immediates are constants)
    auto tap_body = [&](auto KIND, auto TAP, int chunk) {
        constexpr int kind = decltype(KIND)::value, tap = decltype(TAP)::value;
        constexpr int cur = tap % NWB;
        asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(p_wait(kind, tap, HAS_RES, ASCALE)) : "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        // weight tile of step + 2 into the buffer W(step - 1) just left; the LAST chunk's taps 7 / 8 request the next tile's
        // first two (without a next tile: a harmless re-request of this tile's, which keeps the stream uniform)
        if constexpr (kind == K_LAST && tap >= 7) {
            if (tap == 7 && have_next) w_soff = next_wsoff;
            issue_w(0, tap - 7, cur >= 1 ? cur - 1 : NWB - 1);
        } else {
            issue_w(tap + 2 < 9 ? chunk : chunk + 1, (tap + 2) % 9, cur >= 1 ? cur - 1 : NWB - 1);
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // the next chunk's halo (LAST: the next tile's first chunk; hoff / gn_img / ascale_stage already describe that tile)
        const int nchunk = kind == K_LAST ? 0 : chunk + 1;
        const bool live = kind == K_LAST ? have_next : true;
        if constexpr (tap == 0) prefetch_halo_part(nchunk, 0, HSPLIT, live);
        if constexpr (tap == 3) {
            if constexpr (kind == K_LAST && ASCALE) {         // the next tile's operand scale, from the words requested at tap 0
                if (have_next) {
                    float inv;
                    amax_scales(amax_bits, has_gn, ascale_stage, inv);
                    epi_next = d.acc_scale * inv;
                }
            }
            if (live) stage_halo_part(hb ^ 1, 0, HSPLIT);
            prefetch_halo_part(nchunk, HSPLIT, HR, live);
        }
        if constexpr (tap == 6) {
            if (live) stage_halo_part(hb ^ 1, HSPLIT, HR);
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        // extras, always the same number per (kind, tap): p_extra()
        if constexpr (kind == K_FIRST) {
            if constexpr (tap == 0) store_stats();
#pragma unroll
            for (int k = p_first_of(tap); k < p_first_of(tap + 1); ++k)
                __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(outv[k]), r_out, o_lane, o_soff(pend_base, k), 0);
        }
        if constexpr (kind == K_LAST) {
            if constexpr (tap == 0) {
                if constexpr (ASCALE) amax_bits = amax_request(n_img_next, have_next);
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    addv[j][0] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_bias, c_lane + j * 128, (unsigned)(t_ntile * BN * 4), 0));
                    addv[j][1] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_badd, c_lane + j * 128, (unsigned)((t_img * d.badd_stride + t_ntile * BN) * 4), 0));
                }
            }
            if constexpr (tap == 4) {              // (requested four taps ago: no wait to speak of; frees two registers)
#pragma unroll
                for (int j = 0; j < NT; ++j) addv[j][0] = addv[j][0] + addv[j][1];
            }
            if constexpr (HAS_RES) {
#pragma unroll
                for (int k = p_first_of(tap); k < p_first_of(tap + 1); ++k)
                    outv[k] = __uint_as_float(__builtin_amdgcn_raw_buffer_load_b32(r_res, o_lane, o_soff(cur_base, k), 0));
            }
        }
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
        mfma_tap(tap, cur, hb);
    };
    auto run_chunk = [&](auto KIND, int chunk) {
        tap_body(KIND, ic<0>{}, chunk);
        tap_body(KIND, ic<1>{}, chunk);
        tap_body(KIND, ic<2>{}, chunk);
        tap_body(KIND, ic<3>{}, chunk);
        tap_body(KIND, ic<4>{}, chunk);
        tap_body(KIND, ic<5>{}, chunk);
        tap_body(KIND, ic<6>{}, chunk);
        tap_body(KIND, ic<7>{}, chunk);
        tap_body(KIND, ic<8>{}, chunk);
        hb ^= 1;
    };

    // ---- fused 1x1 shortcut (HAS_SKIP): thread -> (float4 column sc of 8, interior pixels srow + 64 i) of the raw input,
    // staged at the pixel's halo position so that the centre tap reads it; its [128][hi 32 | lo 32] weight tile arrives by
    // LDS-DMA like the main ones (same swizzled image; the one-tile kernel stages it through registers)
    constexpr int SR = 4;
    const int sc8 = tid & 7, srow = tid >> 3;
    int s_img = 0, s_ty0 = 0, s_tx0 = 0;
    f32x4 s_st[SR];
    const int SCin = d.SC0 + d.SC1;
    const unsigned sw_rowlen = (unsigned)SCin * 4u;
    const __amdgpu_buffer_rsrc_t r_sw = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(reinterpret_cast<const void*>(d.skip_weight)), 0, HAS_SKIP ? (unsigned)p.n_tiles * BN * sw_rowlen : 0u, 0x00020000);
    const unsigned sw_lane = (unsigned)(wave * 8 + lrow) * sw_rowlen + (unsigned)((lpiece ^ wswz) * 16);
    unsigned sw_soff = 0;
    auto skip_setup = [&](int img, int ty0, int tx0, int ntile) {      // the tile that has just finished its LAST chunk
        s_img = img;
        s_ty0 = ty0;
        s_tx0 = tx0;
        sw_soff = (unsigned)ntile * BN * sw_rowlen;
    };
    auto skip_prefetch = [&](int ch) {
        const int cb = ch * KCH;
        const float* src;
        int cs, coff;
        if (cb < d.SC0) { src = d.skip0; cs = d.SC0; coff = cb; }
        else { src = d.skip1; cs = d.SC1; coff = cb - d.SC0; }
#pragma unroll
        for (int i = 0; i < SR; ++i) {
            const int m = srow + 64 * i;
            const int so = (s_img * p.Hs + s_ty0 + (m >> 5)) * p.Ws + s_tx0 + (m & 31);
            s_st[i] = *reinterpret_cast<const f32x4*>(src + (size_t)so * cs + coff + sc8 * 4);
        }
    };
    auto skip_issue_w = [&](int ch) {                // weight tile of shortcut chunk `ch` -> weight buffer 2
        char* dst = Bs + 2 * WTILE + wave * 1024;
#pragma unroll
        for (int j = 0; j < BR; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_sw, (__attribute__((address_space(3))) void*)(dst + j * (NTHREADS / 64) * 1024), 16,
                                                     sw_lane, sw_soff + (unsigned)ch * KCH * 4u + (unsigned)j * (NTHREADS / 8) * sw_rowlen, 0, 0);
    };
    auto skip_phase = [&](int hbuf, float ascale_cur, int img, int ty0, int tx0, int ntile) {
        const int nsk = SCin / KCH;
        skip_setup(img, ty0, tx0, ntile);
        skip_prefetch(0);
        for (int ch = 0; ch < nsk; ++ch) {
            __syncthreads();                          // everybody is done with the previous reads of these two buffers
            skip_issue_w(ch);
#pragma unroll
            for (int i = 0; i < SR; ++i) {
                const int m = srow + 64 * i;
                split_store(&Hs[hbuf * MAXH * LDH + (((m >> 5) + 1) * HWd + (m & 31) + 1) * LDH + sc8 * 4], s_st[i], ascale_cur);
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // this wave's pieces of the weight tile have landed
            __syncthreads();
            if (ch + 1 < nsk) skip_prefetch(ch + 1);
            mfma_tap(4, 2, hbuf);
        }
    };

    bool pending = false;
    for (;;) {
        int c0 = 0;
        if (pending) {
            run_chunk(ic<K_FIRST>{}, 0);
            c0 = 1;
        }
        for (int chunk = c0; chunk < nchunks - 1; ++chunk) run_chunk(ic<K_MID>{}, chunk);
        // ---- LAST chunk: the bias / residual loads address THIS tile (t_*, cur_base), the halo prefetch the NEXT one
        const int vn = v + G;
        have_next = vn < total;
        const int c_img = t_img, c_ntile = t_ntile;
        const float ascale_cur = ascale_stage;
        const int c_ty0 = t_ty0, c_tx0 = t_tx0;
        if (have_next) {
            tile_of(vn);
            set_hoff();
            gn_img = t_img;
            next_wsoff = (unsigned)t_ntile * BN * w_rowlen;
            next_base = tile_base();
            next_stats = (unsigned)((t_mtile * Cout + t_ntile * BN) * 8);
            n_img_next = t_img;
        }
        {   // bias / addend of the CURRENT tile: tap_body reads t_img / t_ntile
            const int n_img = t_img, n_ntile = t_ntile;
            t_img = c_img;
            t_ntile = c_ntile;
            run_chunk(ic<K_LAST>{}, nchunks - 1);
            t_img = n_img;
            t_ntile = n_ntile;
        }
        if constexpr (HAS_SKIP) skip_phase(hb ^ 1, ascale_cur, c_img, c_ty0, c_tx0, c_ntile);      // run_chunk toggled hb: hb = the next tile's halo, hb ^ 1 is free
        finalize();
        if (!have_next) break;
        v = vn;
        cur_base = next_base;
        cur_stats = next_stats;
        epi_cur = epi_next;
        pending = true;
    }
    // ---- the last tile's values leave directly
    __syncthreads();
    store_stats();
#pragma unroll
    for (int k = 0; k < P_NOUT; ++k) __builtin_amdgcn_raw_buffer_store_b32(__float_as_uint(outv[k]), r_out, o_lane, o_soff(pend_base, k), 0);
}

// Launches of at least two tiles per CU, 32-pixel-wide patches, no upsampled residual / split-K.
bool conv3x3_s16_persist_eligible(const ConvArgs& p) {
    const ddnm_conv_desc& d = p.d;
    const long tiles = (long)p.m_tiles * p.n_tiles;
    if (d.skip0 && (!d.amax_in || d.res)) return false;        // (the only fused-shortcut instance: raw shortcut operand, no residual)
    return p.ksplit == 1 && tiles >= 512 && p.TW == 32 && !d.res_ups && !d.out_nchw && p.Cin / P_KCH >= 2 &&
           (int64_t)d.B * d.Ho * d.Wo * d.Cout * 4 < ((int64_t)1 << 31) && (int64_t)p.m_tiles * d.Cout * 8 < ((int64_t)1 << 31);
}

int conv3x3_s16_persist_launch(const ConvArgs& p, hipStream_t s) {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0)
        cus = 256;
    const int total = p.m_tiles * p.n_tiles;
    cus -= cus % 8;                                     // the XCD-contiguous tile order wants a multiple of 8 workgroups
    const dim3 grid(total < cus ? total : cus);
    const bool asc = p.d.amax_in != nullptr, res = p.d.res != nullptr;
    if (p.d.skip0) { DDNM_LAUNCH((conv3x3_s16_persist_kernel<true, false, true>), grid, dim3(P_NTHREADS), 0, s, p, total); }
    else if (asc && res) { DDNM_LAUNCH((conv3x3_s16_persist_kernel<true, true>), grid, dim3(P_NTHREADS), 0, s, p, total); }
    else if (asc) { DDNM_LAUNCH((conv3x3_s16_persist_kernel<true, false>), grid, dim3(P_NTHREADS), 0, s, p, total); }
    else if (res) { DDNM_LAUNCH((conv3x3_s16_persist_kernel<false, true>), grid, dim3(P_NTHREADS), 0, s, p, total); }
    else { DDNM_LAUNCH((conv3x3_s16_persist_kernel<false, false>), grid, dim3(P_NTHREADS), 0, s, p, total); }
    return 0;
}
