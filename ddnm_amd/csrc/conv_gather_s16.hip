// Per-tap gather convolution in the split-fp16 arithmetic (fp32-grade products on v_mfma_f32_32x32x16_f16), gfx950.
//
// The layers of the celeba `Model` that the 3x3 halo kernel (conv_igemm_f16.hip, SPLIT form) does not take:
//   * Downsample: 3x3 stride 2 with the reference's asymmetric (0,1,0,1) padding (guided_diffusion/models.py:61-71),
//   * 1x1 convolutions: the attention blocks' q / k / v and proj_out, un-fused nin_shortcut (models.py:109,143-162),
//   * 3x3 / stride 1 on the 8 x 8 level, where a 256-pixel tile would span several images.
// Arithmetic and operand formats are those of ddnm_conv3x3_s16_f32 (include/ddnm_hip.h): fp32 tensors, every operand
// value carried as hi + lo fp16 halves, product = hi*hi' + hi*lo' + lo*hi' with fp32 accumulation, weights in the
// split packing ([hi 32 | lo 32] halfs per (row, tap, 32-channel chunk), pre-scaled by a power of two).
//
// Workgroup 256 threads = 4 waves (2 x 2), tile 128 x 128 or 64 x 64; per (chunk, tap) step the A tile [BM][32 channels]
// is gathered straight from global memory through registers -- GroupNorm affine (+ swish) applied, split into
// [hi 32 | lo 32], 144-byte LDS row pitch (conflict-free ds_read_b128), requested TWO steps ahead (two register sets) --
// and the B tile [BN][hi 32 | lo 32] arrives by LDS-DMA two steps ahead (unpadded swizzled image, three buffers); all
// requests are counted (`s_waitcnt vmcnt(N)`, raw barriers): these launches are chains of L2 / HBM round trips.
// Split-K over channel chunks for launches that cannot fill the chip.
#include "conv_common.h"

constexpr int GS_KC = 32;            // channels per chunk
constexpr int GS_LDH = 72;           // LDS row pitch in halfs: [hi 32 | lo 32 | pad 8] = 144 B

template <int WM, int WN, int MT, int NT>
__global__ __launch_bounds__(256) void conv_gather_s16_kernel(const ConvArgs p) {
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
    constexpr int AR = BM / 32, BR = BN / 32;  // A: rows per thread; B: LDS-DMA instructions per wave and tile
    constexpr int NWB = 3;                     // weight tiles in LDS: step, step+1, step+2
    constexpr int WTILE = BN * 128;            // bytes of one weight tile: [BN rows][hi 32 | lo 32] unpadded, swizzled
    // ONE shared object (a second one makes the compiler drain the LDS-DMA queue in front of every fragment read)
    __shared__ __attribute__((aligned(1024))) char lds_all[NWB * WTILE + BM * GS_LDH * 2];
    char* const Bs = lds_all;
    _Float16* const As = reinterpret_cast<_Float16*>(lds_all + NWB * WTILE);

    const ddnm_conv_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tile_id = xcd_swizzle(blockIdx.x, gridDim.x);
    const int n_tile = tile_id % p.n_tiles, m_tile = tile_id / p.n_tiles;
    const int slice = blockIdx.y;
    const TileMap tm = make_tilemap<BM>(p, m_tile);
    const int img = tm.img;
    // operand-range guard: per-launch, per-image power-of-two scale of a RAW operand (Downsample, nin_shortcut, proj_out);
    // GroupNorm'd operands (qkv) carry no bound and run with 1 (scalar loads: no effect on the vmcnt bookkeeping below)
    float ascale, inv_ascale;
    s16_operand_scale(d.gn_scale == nullptr ? d.amax_in : nullptr, img, false, ascale, inv_ascale);

    const int c4 = tid & 7, row0 = tid >> 3;   // A: float4 column (4 channels) of rows row0 + 32 i
    int iy0[AR], ix0[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        int oy, ox;
        tm.pixel(row0 + 32 * i, oy, ox);
        iy0[i] = oy * d.stride - d.pad;
        ix0[i] = ox * d.stride - d.pad;
    }
    // ---- weight tile of step (chunk, tap) -> Bs[buf] by LDS-DMA (the weights are the cold stream of these launches: at
    // 8 x 8 / 16 x 16 every weight byte is used by a handful of pixel tiles and comes from HBM, so the tile of step + 2 is
    // requested while step is multiplied).  One instruction moves 8 rows x 128 B; lane -> (row = 8 g + lane / 8, piece
    // lane % 8) and FETCHES piece ^ swizzle(row); rows of wave w: 8 w + lrow + 32 j, so (row >> 1) & 7 = 4 (w & 1) + (lrow >> 1).
    // Split packing: a row is ntaps * Cin * 4 bytes, (tap, chunk) starts at (tap * Cin + chunk * 32) * 4.
    const int lrow = lane >> 3, lpiece = lane & 7;
    const int wswz = (((wave & 1) << 2) | (lrow >> 1));
    const unsigned w_rowlen = (unsigned)p.ntaps * (unsigned)p.Cin * 4u;
    const int cout_pad = (d.Cout + 127) / 128 * 128;
    const __amdgpu_buffer_rsrc_t r_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(reinterpret_cast<const void*>(d.weight)), 0, (unsigned)cout_pad * w_rowlen, 0x00020000);
    const unsigned w_voff = (unsigned)(n_tile * BN + wave * 8 + lrow) * w_rowlen + (unsigned)((lpiece ^ wswz) * 16);
    auto issue_b = [&](int it, int buf) {
        const int chunk = it / p.ntaps, tap = it - chunk * p.ntaps;
        char* dst = Bs + buf * WTILE + wave * 1024;
        const unsigned so = ((unsigned)tap * p.Cin + (unsigned)chunk * GS_KC) * 4u;
#pragma unroll
        for (int j = 0; j < BR; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w, (__attribute__((address_space(3))) void*)(dst + j * 4096), 16, w_voff,
                                                     so + (unsigned)j * 32u * w_rowlen, 0, 0);
    };

    // ---- A tile: gathered through registers TWO steps ahead (two register sets used in turn).  Buffer loads with an
    // out-of-range offset for padding (the load returns zero) and GroupNorm vectors fetched unconditionally (a dummy
    // address without GroupNorm): every call issues EXACTLY AR + 2 requests per wave, so the loop can leave a whole younger
    // step in flight behind the tile it waits for (`s_waitcnt vmcnt(N)` needs exact request counts).
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    constexpr unsigned AOOB = 0x80000000u;          // every tensor here is < 2 GB (checked by the host)
    const bool has_gn = d.gn_scale != nullptr;
    const __amdgpu_buffer_rsrc_t r_s0 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(d.src0), 0, (unsigned)d.B * p.Hs * p.Ws * d.C0 * 4u, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_s1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(d.C1 > 0 ? d.src1 : d.src0), 0, (unsigned)d.B * p.Hs * p.Ws * (d.C1 > 0 ? d.C1 : d.C0) * 4u, 0x00020000);
    const float* const gn_sc = has_gn ? d.gn_scale + (size_t)img * p.Cin + c4 * 4 : reinterpret_cast<const float*>(d.weight);
    const float* const gn_sh = has_gn ? d.gn_shift + (size_t)img * p.Cin + c4 * 4 : reinterpret_cast<const float*>(d.weight);
    f32x4 a_st0[AR], a_st1[AR];
    f32x4 gsc0, gsh0, gsc1, gsh1;
    unsigned a_valid0 = 0, a_valid1 = 0;

    auto prefetch_a = [&](int it, f32x4 (&a_st)[AR], f32x4& gsc, f32x4& gsh, unsigned& a_valid) {
        const int chunk = it / p.ntaps, tap = it - chunk * p.ntaps;
        const int ky = tap / d.ksize, kx = tap - ky * d.ksize;
        const int cb = chunk * GS_KC;
        const bool first = cb < d.C0;
        const unsigned cs = first ? d.C0 : d.C1, coff = first ? cb : cb - d.C0;
        const __amdgpu_buffer_rsrc_t r_s = first ? r_s0 : r_s1;            // wave-uniform: a scalar select, no branch
        a_valid = 0;
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            // branch-free: a divergent `ok ? load : zero` makes hipcc duplicate the load into both arms and guard the
            // second one with a full `vmcnt(0)` (same destination registers), which drains the requests in flight
            const int iy = iy0[i] + ky, ix = ix0[i] + kx;
            const bool ok = (unsigned)iy < (unsigned)d.Hin && (unsigned)ix < (unsigned)d.Win;
            const int sy = d.ups ? (iy >> 1) : iy, sx = d.ups ? (ix >> 1) : ix;
            const unsigned vo = (((unsigned)((img * p.Hs + sy) * p.Ws + sx) * cs + c4 * 4) * 4u) | (ok ? 0u : AOOB);
            a_st[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(r_s, vo, coff * 4u, 0));
            a_valid |= ok ? (1u << i) : 0u;
        }
        gsc = *reinterpret_cast<const f32x4*>(gn_sc + (has_gn ? cb : 0));
        gsh = *reinterpret_cast<const f32x4*>(gn_sh + (has_gn ? cb : 0));
    };

    auto stage_a = [&](f32x4 (&a_st)[AR], const f32x4& gsc, const f32x4& gsh, unsigned a_valid) {
#pragma unroll
        for (int i = 0; i < AR; ++i) {
            f32x4 v = a_st[i];
            if (has_gn && (a_valid & (1u << i))) v = gn_act(v, gsc, gsh, d.gn_silu);
            split_store(&As[(row0 + 32 * i) * GS_LDH + c4 * 4], v, ascale);   // hi at channel c4*4, lo 32 halfs behind it
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nchunks = p.Cin / GS_KC;
    const int c_begin = (int)((long)nchunks * slice / p.ksplit), c_end = (int)((long)nchunks * (slice + 1) / p.ksplit);
    const int it_begin = c_begin * p.ntaps, it_end = c_end * p.ntaps;
    const _Float16* a_frag = As + (wm * MT * 32) * GS_LDH + (lane & 31) * GS_LDH + (lane >> 5) * 8;
    // B fragment of 16-byte piece q (+ lane >> 5): row n = wn*NT*32 + j*32 + (lane & 31), swizzled with (n >> 1) & 7
    const int b_frag = ((wn * NT * 32 + (lane & 31)) * 128) + ((((lane >> 5) ^ (((lane & 31) >> 1) & 7))) << 4);

    // Per step:  [barrier: As and the buffer of W(it-1) are free | A(it): registers -> GroupNorm / split -> LDS |
    //             my pieces of W(it) landed | barrier | request A(it+2) (registers) and W(it+2) (LDS-DMA) | MFMA(it)].
    // Requests beyond the last step are clamped to it (re-fetching what nobody reads keeps the `vmcnt` bookkeeping
    // uniform): behind W(it) exactly the AR + 2 + BR requests of step it+1 are in flight when it is awaited.
    if (it_begin < it_end) {
        const int last = it_end - 1;
        auto clampi = [&](int q) { return q < last ? q : last; };
        prefetch_a(it_begin, a_st0, gsc0, gsh0, a_valid0);
        issue_b(it_begin, 0);
        prefetch_a(clampi(it_begin + 1), a_st1, gsc1, gsh1, a_valid1);
        issue_b(clampi(it_begin + 1), 1);
        int cur = 0;
        auto step = [&](int it, f32x4 (&a_st)[AR], f32x4& gsc, f32x4& gsh, unsigned& a_valid) {
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            stage_a(a_st, gsc, gsh, a_valid);
            asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(AR + 2 + BR) : "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            prefetch_a(clampi(it + 2), a_st, gsc, gsh, a_valid);
            issue_b(clampi(it + 2), cur >= 1 ? cur - 1 : NWB - 1);
            __builtin_amdgcn_sched_barrier(0);
            const char* bf = Bs + cur * WTILE;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                half8 ah[MT], al[MT], bh[NT], bl[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    ah[i] = *reinterpret_cast<const half8*>(a_frag + i * 32 * GS_LDH + ks * 16);
                    al[i] = *reinterpret_cast<const half8*>(a_frag + i * 32 * GS_LDH + ks * 16 + 32);
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
            cur = cur == NWB - 1 ? 0 : cur + 1;
        };
        // pairs in the loop, an odd last step behind it: with `if (it + 1 < it_end) step(..)` inside, hipcc sees a back
        // edge with only one step's requests in flight and waits for the A registers with `vmcnt(BR)` -- one step of
        // lookahead instead of two
        int it = it_begin;
        for (; it + 1 < it_end; it += 2) {
            step(it, a_st0, gsc0, gsh0, a_valid0);
            step(it + 1, a_st1, gsc1, gsh1, a_valid1);
        }
        if (it < it_end) step(it, a_st0, gsc0, gsh0, a_valid0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the tail requests
    }
    __syncthreads();               // the statistics epilogue reuses As
    conv_epilogue<WM, WN, MT, NT>(p, tm, n_tile, m_tile, slice, acc, reinterpret_cast<float*>(As), d.acc_scale * inv_ascale);
}

// ---------------------------------------------------------------------------------------------
struct PlanGS {
    int tile;       // 1: 128x128, 2: 64x64
    int BM, BN, ksplit;
};

static bool plan_gs(const ddnm_conv_desc* d, PlanGS* pl) {
    const int HWo = d->Ho * d->Wo;
    const int Cin = d->C0 + d->C1;
    if ((d->ksize != 1 && d->ksize != 3) || (d->stride != 1 && d->stride != 2)) return false;
    if (d->C0 <= 0 || d->C0 % GS_KC || d->C1 % GS_KC || d->out_nchw || d->src_f16 || d->skip0) return false;
    if (d->Cout % 64 || HWo % 64) return false;
    {   // the loader addresses both sources with 32-bit buffer offsets and an out-of-range sentinel for padding
        const int64_t px = (int64_t)d->B * (d->ups ? d->Hin / 2 : d->Hin) * (d->ups ? d->Win / 2 : d->Win);
        const int64_t cmax = d->C0 > d->C1 ? d->C0 : d->C1;
        if (px * cmax * 4 >= (int64_t)1 << 31) return false;
    }
    const int nchunks = Cin / GS_KC;
    int tile = 2;
    if (HWo % 128 == 0 && d->Cout % 128 == 0) {
        const long tiles = (long)d->B * (HWo / 128) * (d->Cout / 128);
        if (tiles * nchunks >= 256) tile = 1;
    }
    pl->tile = tile;
    pl->BM = pl->BN = tile == 1 ? 128 : 64;
    const long tiles = (long)d->B * (HWo / pl->BM) * (d->Cout / pl->BN);
    int ks = 1;
    if (d->Cout % 4 == 0 && tiles < 192) {
        ks = (int)((512 + tiles - 1) / tiles);     // (256 / 1024 workgroups, ks up to 32: forward time unchanged within 0.3 %)
        if (ks > nchunks) ks = nchunks;
        if (ks > 16) ks = 16;
        if (ks < 1) ks = 1;
    }
    pl->ksplit = ks;
    return true;
}

extern "C" int ddnm_conv_gather_s16_supported(const ddnm_conv_desc* d) {
    PlanGS pl;
    return d && plan_gs(d, &pl) ? 1 : 0;
}

extern "C" int64_t ddnm_conv_gather_s16_workspace_floats(const ddnm_conv_desc* d) {
    PlanGS pl;
    if (!d || !plan_gs(d, &pl)) return DDNM_E_SHAPE;
    return pl.ksplit > 1 ? (int64_t)pl.ksplit * d->B * d->Ho * d->Wo * d->Cout : 0;
}

extern "C" int ddnm_conv_gather_s16_stats_tiles(const ddnm_conv_desc* d) {
    PlanGS pl;
    if (!d || !plan_gs(d, &pl)) return DDNM_E_SHAPE;
    return pl.ksplit > 1 ? splitk_stats_tiles(d) : d->Ho * d->Wo / pl.BM;
}

extern "C" int ddnm_conv_gather_s16_f32(const ddnm_conv_desc* d, void* stream) {
    if (!d || !d->src0 || !d->weight || !d->out) return DDNM_E_BADARG;
    if (d->B <= 0 || d->Cout <= 0 || d->Ho <= 0 || d->Wo <= 0) return DDNM_E_BADARG;
    if (!conv_sizes_addressable(d)) return DDNM_E_SHAPE;
    if (d->C1 > 0 && !d->src1) return DDNM_E_BADARG;
    if (d->gn_scale && !d->gn_shift) return DDNM_E_BADARG;
    if (!(d->acc_scale > 0.f)) return DDNM_E_BADARG;
    if (!d->gn_scale && !d->amax_in) return DDNM_E_BADARG;      // a raw operand needs the operand bound (fp16 range)
    if (d->ups && ((d->Hin | d->Win) & 1)) return DDNM_E_SHAPE;
    if (d->res_ups && ((d->Ho | d->Wo) & 1)) return DDNM_E_SHAPE;
    PlanGS pl;
    if (!plan_gs(d, &pl)) return DDNM_E_SHAPE;
    if (pl.ksplit > 1) {
        const int64_t need = (int64_t)pl.ksplit * d->B * d->Ho * d->Wo * d->Cout;
        if (!d->workspace || d->workspace_floats < need) {
            if (d->stats_out) return DDNM_E_BADARG;       // the caller sized stats_out for the split plan
            pl.ksplit = 1;
        }
    }
    ConvArgs p;
    p.d = *d;
    p.Cin = d->C0 + d->C1;
    p.ntaps = d->ksize * d->ksize;
    p.Hs = d->ups ? d->Hin / 2 : d->Hin;
    p.Ws = d->ups ? d->Win / 2 : d->Win;
    p.m_tiles = d->B * (d->Ho * d->Wo / pl.BM);
    p.n_tiles = d->Cout / pl.BN;
    p.TW = 32;
    p.TW_log2 = 5;
    p.tiles_x = 0;                 // flat strips of BM consecutive pixels of one image
    p.ksplit = pl.ksplit;
    p.ws = d->workspace;
    const dim3 grid(p.m_tiles * p.n_tiles, pl.ksplit);
    hipStream_t s = (hipStream_t)stream;
    if (pl.tile == 1) { DDNM_LAUNCH((conv_gather_s16_kernel<2, 2, 2, 2>), grid, dim3(256), 0, s, p); }
    else { DDNM_LAUNCH((conv_gather_s16_kernel<2, 2, 1, 1>), grid, dim3(256), 0, s, p); }
    if (pl.ksplit > 1) return launch_splitk_reduce(p, s);
    return 0;
}
