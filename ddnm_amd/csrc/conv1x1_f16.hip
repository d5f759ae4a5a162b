// 1x1 convolution (= GEMM over pixels) on v_mfma_f32_32x32x16_f16 for the ADM `use_fp16` torso, gfx950:
// the attention blocks' qkv / proj_out Conv1d(k=1) and un-fused 1x1 shortcuts (guided_diffusion/unet.py:222,
// 283-289, 301-308).  out[m][n] = bias[n] + res[m][n] + sum_k A[m][k] * W[n][k],  m = pixel, k = input channel.
//
// Same building blocks as conv_igemm_f16.hip without a halo: workgroup 512 threads = 8 waves (4 x 2), block tile
// 256 pixels x 128 channels, K chunk 64; A tile [256][64+8] and W tile [128][64+8] fp16, BOTH double-buffered in
// LDS (110 KB); one barrier per chunk, global loads two chunks ahead.  A is fp32 (rounded while staged) or the
// fp16 tensor written by ddnm_gn_apply_f16 (src_f16 = 1: GroupNorm affine of the qkv input applied once).
// Epilogue, split-K and the GroupNorm statistics are the shared ones of conv_common.h.
#include "conv_common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

constexpr int GK = 64;             // channels per chunk
constexpr int GLD = GK + 8;        // LDS row pitch in halfs (144 B = 36 dwords: conflict-free ds_read_b128)

template <bool SRC16>
__global__ __launch_bounds__(512) void conv1x1_f16_kernel(const ConvArgs p) {
    constexpr int WM = 4, WN = 2, MT = 2, NT = 2;
    constexpr int BM = 256, BN = 128;
    __shared__ __attribute__((aligned(16))) _Float16 As[2 * BM * GLD];
    __shared__ __attribute__((aligned(16))) _Float16 Bs[2 * BN * GLD];
    __shared__ __attribute__((aligned(16))) float stat_lds[WM * BN * 2];

    const ddnm_conv_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tile_id = xcd_swizzle(blockIdx.x, gridDim.x);
    const int n_tile = tile_id % p.n_tiles, m_tile = tile_id / p.n_tiles;
    const int slice = blockIdx.y;
    const TileMap tm = make_tilemap<BM>(p, m_tile);
    const size_t row0 = (size_t)m_tile * BM;                       // first pixel (flat over batch) of this tile

    // A loader: fp32 rows are 16 float4, fp16 rows 8 uint4; every thread owns one 16-byte column of AR rows
    constexpr int ACOLS = SRC16 ? 8 : 16, AROWS_PER_PASS = 512 / ACOLS, AR = BM / AROWS_PER_PASS;
    const int ac = tid % ACOLS, arow = tid / ACOLS;
    const int bc = tid & 7, brow = tid >> 3;                        // W loader: 8 uint4 per row, 64 rows per pass
    const _Float16* wbase = reinterpret_cast<const _Float16*>(d.weight) + (size_t)(n_tile * BN + brow) * p.Cin + bc * 8;

    const int nchunks = p.Cin / GK;
    const int c_begin = (int)((long)nchunks * slice / p.ksplit), c_end = (int)((long)nchunks * (slice + 1) / p.ksplit);

    // named staging registers (arrays of them get demoted to LDS / scratch by the compiler, see conv_igemm_f16.hip)
    static_assert(AR == 4 || AR == 8, "A staging registers");
    uint4 a0, a1, a2, a3, a4, a5, a6, a7;
    a0 = a1 = a2 = a3 = a4 = a5 = a6 = a7 = uint4{0u, 0u, 0u, 0u};
    uint4 b_st0 = {0u, 0u, 0u, 0u}, b_st1 = {0u, 0u, 0u, 0u};
    auto prefetch = [&](int chunk) {
        const int cb = chunk * GK;
        const char* src;
        int cs, coff;
        if (cb < d.C0) { src = reinterpret_cast<const char*>(d.src0); cs = d.C0; coff = cb; }
        else { src = reinterpret_cast<const char*>(d.src1); cs = d.C1; coff = cb - d.C0; }
        constexpr int ESZ = SRC16 ? 2 : 4, AVEC = SRC16 ? 8 : 4;
        auto ld = [&](int i) {
            return *reinterpret_cast<const uint4*>(src + ((row0 + arow + AROWS_PER_PASS * i) * cs + coff + ac * AVEC) * ESZ);
        };
        a0 = ld(0); a1 = ld(1); a2 = ld(2); a3 = ld(3);
        if constexpr (AR == 8) { a4 = ld(4); a5 = ld(5); a6 = ld(6); a7 = ld(7); }
        b_st0 = *reinterpret_cast<const uint4*>(wbase + cb);
        b_st1 = *reinterpret_cast<const uint4*>(wbase + (size_t)64 * p.Cin + cb);
    };
    auto stage = [&](int buf) {
        auto st = [&](int i, const uint4 r) {
            _Float16* dst = &As[buf * BM * GLD + (arow + AROWS_PER_PASS * i) * GLD];
            if constexpr (SRC16) {
                *reinterpret_cast<uint4*>(dst + ac * 8) = r;
            } else {
                const f32x4 v = __builtin_bit_cast(f32x4, r);
                half4 h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
                *reinterpret_cast<half4*>(dst + ac * 4) = h;
            }
        };
        st(0, a0); st(1, a1); st(2, a2); st(3, a3);
        if constexpr (AR == 8) { st(4, a4); st(5, a5); st(6, a6); st(7, a7); }
        _Float16* bd = &Bs[buf * BN * GLD + brow * GLD + bc * 8];
        *reinterpret_cast<uint4*>(bd) = b_st0;
        *reinterpret_cast<uint4*>(bd + 64 * GLD) = b_st1;
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const _Float16* a_frag = As + ((wm * MT * 32) + (lane & 31)) * GLD + (lane >> 5) * 8;
    const _Float16* b_frag = Bs + ((wn * NT * 32) + (lane & 31)) * GLD + (lane >> 5) * 8;

    if (c_begin < c_end) {
        prefetch(c_begin);
        stage(0);
        if (c_begin + 1 < c_end) prefetch(c_begin + 1);
        __syncthreads();
        int cur = 0;
        for (int chunk = c_begin; chunk < c_end; ++chunk) {
            if (chunk + 1 < c_end) {
                stage(cur ^ 1);
                if (chunk + 2 < c_end) prefetch(chunk + 2);
            }
            __builtin_amdgcn_sched_barrier(0);        // keep the prefetch in front of the MFMAs (see conv_igemm_f16.hip)
            const _Float16* af = a_frag + cur * BM * GLD;
            const _Float16* bf = b_frag + cur * BN * GLD;
#pragma unroll
            for (int ks = 0; ks < GK / 16; ++ks) {
                half8 a[MT], b[NT];
#pragma unroll
                for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const half8*>(af + i * 32 * GLD + ks * 16);
#pragma unroll
                for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const half8*>(bf + j * 32 * GLD + ks * 16);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
            }
            __syncthreads();
            cur ^= 1;
        }
    }
    conv_epilogue<WM, WN, MT, NT, true>(p, tm, n_tile, m_tile, slice, acc, stat_lds);
}

// ---------------------------------------------------------------------------------------------
struct Plan1x1 {
    int fold;        // 1: the batch is folded into one "image" (pixels per image < 256): no statistics, no badd
    int ksplit;
};

static bool plan_1x1(const ddnm_conv_desc* d, Plan1x1* pl) {
    const int HW = d->Ho * d->Wo, Cin = d->C0 + d->C1;
    if (d->ksize != 1 || d->stride != 1 || d->pad != 0 || d->ups || d->Ho != d->Hin || d->Wo != d->Win) return false;
    if (Cin % GK || d->C0 % GK || d->Cout % 128 || d->out_nchw || d->gn_scale || d->skip0 || d->res_ups) return false;
    if (((long)d->B * HW) % 256) return false;
    pl->fold = HW % 256 ? 1 : 0;
    if (pl->fold && d->badd) return false;
    const long tiles = ((long)d->B * HW / 256) * (d->Cout / 128);
    int ks = 1;
    const int nchunks = Cin / GK;
    if (tiles < 128) {
        ks = (int)((256 + tiles - 1) / tiles);
        if (ks > nchunks / 2) ks = nchunks / 2;
        if (ks > 32) ks = 32;            // K = 9*C of an im2col'd 8x8 layer: up to 288 chunks
        if (ks < 1) ks = 1;
    }
    pl->ksplit = ks;
    return true;
}

extern "C" int ddnm_conv1x1_f16_supported(const ddnm_conv_desc* d) {
    Plan1x1 pl;
    return d && plan_1x1(d, &pl) ? 1 : 0;
}

extern "C" int64_t ddnm_conv1x1_f16_workspace_floats(const ddnm_conv_desc* d) {
    Plan1x1 pl;
    if (!d || !plan_1x1(d, &pl)) return DDNM_E_SHAPE;
    return pl.ksplit > 1 ? (int64_t)pl.ksplit * d->B * d->Ho * d->Wo * d->Cout : 0;
}

extern "C" int ddnm_conv1x1_f16_stats_tiles(const ddnm_conv_desc* d) {
    Plan1x1 pl;
    if (!d || !plan_1x1(d, &pl)) return DDNM_E_SHAPE;
    if (pl.fold) return 0;
    return pl.ksplit > 1 ? splitk_stats_tiles(d) : d->Ho * d->Wo / 256;
}

extern "C" int ddnm_conv1x1_f16_f32(const ddnm_conv_desc* d, void* stream) {
    if (!d || !d->src0 || !d->weight || !d->out) return DDNM_E_BADARG;
    if (d->B <= 0 || d->Cout <= 0 || d->Ho <= 0 || d->Wo <= 0) return DDNM_E_BADARG;
    if (!conv_sizes_addressable(d)) return DDNM_E_SHAPE;
    if (d->C1 > 0 && !d->src1) return DDNM_E_BADARG;
    Plan1x1 pl;
    if (!plan_1x1(d, &pl)) return DDNM_E_SHAPE;
    if (pl.fold && d->stats_out) return DDNM_E_SHAPE;
    if (pl.ksplit > 1) {
        const int64_t need = (int64_t)pl.ksplit * d->B * d->Ho * d->Wo * d->Cout;
        if (!d->workspace || d->workspace_floats < need) {
            if (d->stats_out) return DDNM_E_BADARG;
            pl.ksplit = 1;
        }
    }
    ConvArgs p;
    p.d = *d;
    if (pl.fold) {                      // pixels of all images as one flat strip
        p.d.Ho = 1;
        p.d.Wo = d->B * d->Ho * d->Wo;
        p.d.Hin = 1;
        p.d.Win = p.d.Wo;
        p.d.B = 1;
    }
    p.Cin = d->C0 + d->C1;
    p.ntaps = 1;
    p.Hs = p.d.Hin;
    p.Ws = p.d.Win;
    p.m_tiles = p.d.B * (p.d.Ho * p.d.Wo / 256);
    p.n_tiles = d->Cout / 128;
    p.TW = 256;
    p.TW_log2 = 8;
    p.tiles_x = 0;                      // flat strips of 256 pixels
    p.ksplit = pl.ksplit;
    p.ws = d->workspace;
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(p.m_tiles * p.n_tiles, pl.ksplit);
    if (d->src_f16) { DDNM_LAUNCH(conv1x1_f16_kernel<true>, grid, dim3(512), 0, s, p); }
    else { DDNM_LAUNCH(conv1x1_f16_kernel<false>, grid, dim3(512), 0, s, p); }
    if (pl.ksplit > 1) return launch_splitk_reduce(p, s);
    return 0;
}
