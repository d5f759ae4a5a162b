// Implicit-GEMM convolution on v_mfma_f32_32x32x2_f32 (exact fp32), gfx950.
//
//   GEMM view: M = B*Ho*Wo output pixels, N = Cout, K = taps * Cin.
//   Workgroup = 256 threads = 4 waves arranged WM x WN; each wave owns MT x NT tiles of 32x32.
//   Operands land k-contiguous in LDS (row pitch 36 floats), so each lane fetches its MFMA
//   operands with one conflict-free ds_read_b128 per 4 MFMAs; global loads for step i+1 are
//   issued before the MFMAs of step i (register staging).
//
// Two kernels share the MFMA core and the epilogue:
//   * conv3x3_halo  (3x3, stride 1, pad 1 -- >97 % of the UNet's FLOPs): the (TH+2)x(TW+2)
//     input halo of a TH x TW output tile is staged ONCE per 32-channel chunk -- through the
//     optional nearest-x2 upsample, channel concat of two sources and GroupNorm-affine(+swish)
//     prologue -- and the 9 taps read it at shifted LDS rows; only the [BN][32] weight tile is
//     re-staged per tap.  Activation loads and GN/swish VALU work drop 9x vs per-tap gathering.
//   * conv_gather   (1x1, strided 3x3 with asymmetric padding): the A tile [BM][32] is gathered
//     per tap straight from global memory.
// Split-K: layers with too few output tiles to fill 256 CUs (8x8 .. 32x32 levels) split the
//   channel chunks over `ksplit` workgroups per tile; partial tiles go to a workspace slab and a
//   second kernel reduces them in a fixed order (deterministic) and applies the epilogue.
// Epilogue: + bias + per-sample addend (temb projection) + residual, NHWC or NCHW store.
//
// Replaces: nn.Conv2d call sites of guided_diffusion/models.py (see include/ddnm_hip.h).
#include "conv_common.h"

template <int MT, int NT>
__device__ __forceinline__ void mfma_tile_step(const float* a_frag, const float* b_frag, f32x16 (&acc)[MT][NT]) {
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

// =====================================================================================
// 3x3 stride-1 pad-1 convolution with an LDS-resident input halo
// =====================================================================================
template <int WM, int WN, int MT, int NT>
__global__ __launch_bounds__(256) void conv3x3_halo_f32_kernel(const ConvArgs p) {
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
    constexpr int MAXH = BM == 128 ? 204 : 136;          // (TH+2)*(TW+2) for TW in {8,16,32}
    constexpr int HR = (MAXH + 31) / 32;                 // halo rows staged per thread
    constexpr int BR = BN / 32;
    constexpr int NBUF = 2;        // weight tile double-buffered: one barrier per tap
    // ONE shared object (a second one makes the compiler drain the LDS-DMA queue in front of every fragment read).
    // Weight tiles first: [NBUF][BN rows][32 floats = 128 B], UNPADDED -- they arrive by LDS-DMA (lane-linear
    // destination), so the bank-conflict fix is an XOR swizzle of the 16-byte piece index with (row >> 1) & 7, applied
    // to the per-lane SOURCE address and to the fragment read address; the halo keeps its 36-float pitch (it is
    // written through registers, behind the GroupNorm + swish prologue).
    constexpr int BSF = BN * KC;                          // floats per weight buffer
    __shared__ __attribute__((aligned(1024))) float lds_all[NBUF * BSF + MAXH * LDT
    ];
    float* const Bs = lds_all;
    float* const Hs = lds_all + NBUF * BSF;
    const ddnm_conv_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tile_id = xcd_swizzle(blockIdx.x, gridDim.x);
    const int n_tile = tile_id % p.n_tiles, m_tile = tile_id / p.n_tiles;
    const int slice = blockIdx.y;
    const TileMap tm = make_tilemap<BM>(p, m_tile);
    const int img = tm.img;
    const int TH = BM >> p.TW_log2, HWd = p.TW + 2;
    const int NP = (TH + 2) * HWd;

    // ---- halo loader mapping: thread -> (float4 column c4, halo rows prow + 32*i)
    const int c4 = tid & 7, prow = tid >> 3;
    int hoff[HR];                                         // source pixel index, -1: zero (padding / unused row)
#pragma unroll
    for (int i = 0; i < HR; ++i) {
        const int row = prow + 32 * i;
        const int hy = row / HWd, hx = row - hy * HWd;
        const int iy = tm.ty0 - 1 + hy, ix = tm.tx0 - 1 + hx;
        const bool ok = row < NP && (unsigned)iy < (unsigned)d.Hin && (unsigned)ix < (unsigned)d.Win;
        const int sy = d.ups ? (iy >> 1) : iy, sx = d.ups ? (ix >> 1) : ix;
        hoff[i] = ok ? (img * p.Hs + sy) * p.Ws + sx : -1;
    }

    // K range of this workgroup (channel chunks)
    const int nchunks = p.Cin / KC;
    const int c_begin = (int)((long)nchunks * slice / p.ksplit), c_end = (int)((long)nchunks * (slice + 1) / p.ksplit);

    f32x4 h_st[HR], b_st[BR];
    f32x4 gsc = {1.f, 1.f, 1.f, 1.f}, gsh = {0.f, 0.f, 0.f, 0.f};
    const bool has_gn = d.gn_scale != nullptr;

    auto prefetch_halo = [&](int chunk) {
        const int cb = chunk * KC;
        const float* src;
        int cs, coff;
        if (cb < d.C0) { src = d.src0; cs = d.C0; coff = cb; }
        else { src = d.src1; cs = d.C1; coff = cb - d.C0; }
#pragma unroll
        for (int i = 0; i < HR; ++i) {
            f32x4 v = {0.f, 0.f, 0.f, 0.f};
            if (hoff[i] >= 0) v = *reinterpret_cast<const f32x4*>(src + (size_t)hoff[i] * cs + coff + c4 * 4);
            h_st[i] = v;
        }
        if (has_gn) {
            gsc = *reinterpret_cast<const f32x4*>(d.gn_scale + (size_t)img * p.Cin + cb + c4 * 4);
            gsh = *reinterpret_cast<const f32x4*>(d.gn_shift + (size_t)img * p.Cin + cb + c4 * 4);
        }
    };
    // ---- weight tile of (chunk, tap) -> Bs[buf] by LDS-DMA: one instruction moves 8 rows x 128 B; lane ->
    // (row = 8*g + lane/8, piece lane%8) and the piece it FETCHES is piece ^ swizzle(row), so the linear image holds the
    // swizzled layout.  Rows of wave w: 8w + lrow + 32 j, so (row >> 1) & 7 = 4 (w & 1) + (lrow >> 1) for every j.
    const int lrow = lane >> 3, lpiece = lane & 7;
    const int wswz = (((wave & 1) << 2) | (lrow >> 1));
    const unsigned w_rowlen = 9u * (unsigned)p.Cin;
    const __amdgpu_buffer_rsrc_t r_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(d.weight), 0, (unsigned)p.n_tiles * BN * w_rowlen * 4u, 0x00020000);
    const unsigned w_voff = (((unsigned)(n_tile * BN + wave * 8 + lrow)) * w_rowlen + (unsigned)((lpiece ^ wswz) * 4)) * 4u;
    auto issue_w = [&](int chunk, int tap, int buf) {
        char* dst = reinterpret_cast<char*>(Bs + buf * BSF) + wave * 1024;
        const unsigned so = ((unsigned)tap * p.Cin + (unsigned)chunk * KC) * 4u;
#pragma unroll
        for (int j = 0; j < BR; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w, (__attribute__((address_space(3))) void*)(dst + j * 4096), 16,
                                                     w_voff, so + (unsigned)j * 32u * w_rowlen * 4u, 0, 0);
    };
    auto stage_halo = [&](bool apply_gn = true) {
#pragma unroll
        for (int i = 0; i < HR; ++i) {
            const int row = prow + 32 * i;
            if (row < MAXH) {
                f32x4 v = h_st[i];
                if (apply_gn && has_gn && hoff[i] >= 0) v = gn_act(v, gsc, gsh, d.gn_silu);
                *reinterpret_cast<f32x4*>(&Hs[row * LDT + c4 * 4]) = v;
            }
        }
    };
    auto stage_b = [&](int buf) {          // register-staged weights (fused shortcut): the same swizzled image
#pragma unroll
        for (int i = 0; i < BR; ++i) {
            const int row = prow + 32 * i;
            *reinterpret_cast<f32x4*>(&Bs[buf * BSF + row * KC + ((c4 ^ ((row >> 1) & 7)) << 2)]) = b_st[i];
        }
    };

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // A-fragment base (tap 0,0): lane's output pixel -> halo row
    int a_off[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = (wm * MT + i) * 32 + (lane & 31);
        const int ty = m >> p.TW_log2, tx = m & (p.TW - 1);
        a_off[i] = (ty * HWd + tx) * LDT + (lane >> 5) * 4;
    }
    // B fragment of k-step kk: row n = wn*NT*32 + j*32 + (lane & 31), piece (lane >> 5) + 2 kk, swizzled with
    // (n >> 1) & 7 = ((lane & 31) >> 1) & 7; the k-step enters as an XOR of bits 5-6 of the byte address
    const int b_frag = ((wn * NT * 32 + (lane & 31)) * KC * 4) + ((((lane >> 5) ^ (((lane & 31) >> 1) & 7))) << 4);

    auto mfma_tap = [&](int tap, int buf) {
        const int ky = tap / 3, kx = tap - 3 * ky;
        const int tap_off = (ky * HWd + kx) * LDT;
        const char* bbase = reinterpret_cast<const char*>(Bs + buf * BSF);
        // explicit two-deep fragment pipeline: the reads of k-step kk+1 are in flight under the 16 MFMAs of kk
        f32x4 a[2][MT], b[2][NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) a[0][i] = *reinterpret_cast<const f32x4*>(Hs + a_off[i] + tap_off);
#pragma unroll
        for (int j = 0; j < NT; ++j) b[0][j] = *reinterpret_cast<const f32x4*>(bbase + b_frag + j * 32 * KC * 4);
#pragma unroll
        for (int kk = 0; kk < KC / 8; ++kk) {
            const int cur = kk & 1, nxt = cur ^ 1;
            if (kk + 1 < KC / 8) {
#pragma unroll
                for (int i = 0; i < MT; ++i) a[nxt][i] = *reinterpret_cast<const f32x4*>(Hs + a_off[i] + tap_off + (kk + 1) * 8);
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    b[nxt][j] = *reinterpret_cast<const f32x4*>(bbase + ((b_frag ^ ((kk + 1) << 5)) + j * 32 * KC * 4));
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = mfma_k8(a[cur][i], b[cur][j], acc[i][j]);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // ---- weight tile double-buffered, by LDS-DMA: per tap  [barrier: B(s) landed, buffer of B(s-1) free |
    //      request B(s+1) into it | MFMA(s)];  the halo is re-staged between two barriers at every chunk boundary.
    // The request flies under the 64 MFMAs per wave of a tap and is drained by the `vmcnt(0)` of the next tap's
    // barrier; no staging registers, no ds_write pass, no wait on a register destination inside the loop.
    if (c_begin < c_end) {
        prefetch_halo(c_begin);
        issue_w(c_begin, 0, 0);
        stage_halo();
        int cur = 0;
        for (int chunk = c_begin; chunk < c_end; ++chunk) {
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const bool last_tap = tap == 8, more = chunk + 1 < c_end;
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's pieces of B(s) (and of the halo prefetch)
                __syncthreads();                                        // everybody's; B(s-1) and (tap 0) the halo are free
                if (!last_tap) issue_w(chunk, tap + 1, cur ^ 1);
                else if (more) issue_w(chunk + 1, 0, cur ^ 1);
                if (tap == 7 && more) prefetch_halo(chunk + 1);         // registers; staged after tap 8
                __builtin_amdgcn_sched_barrier(0);      // requests in FRONT of the MFMAs (the scheduler sinks them otherwise)
                mfma_tap(tap, cur);
                if (last_tap && more) {                     // chunk boundary: every wave must be done with Hs
                    __syncthreads();
                    stage_halo();
                }
                cur ^= 1;
            }
        }
    }
    // ---- fused 1x1 shortcut (nin_shortcut / skip_connection of a residual block): extra K chunks that read
    // the block's RAW input (no GroupNorm) at the centre tap and accumulate into the same tile, so the
    // shortcut needs neither its own launch nor an HBM round trip of its output.
    if (d.skip0 != nullptr) {
        // split-K launches share the shortcut's chunks like the main ones (slice 0 alone would run 2-3x longer)
        const int SCin = d.SC0 + d.SC1, nsk_all = SCin / KC;
        const int s_begin = (int)((long)nsk_all * slice / p.ksplit), s_end = (int)((long)nsk_all * (slice + 1) / p.ksplit);
        const float* swbase = d.skip_weight + (size_t)(n_tile * BN + prow) * SCin + c4 * 4;
        auto prefetch_skip = [&](int ch) {
            const int cb = ch * KC;
            const float* src;
            int cs, coff;
            if (cb < d.SC0) { src = d.skip0; cs = d.SC0; coff = cb; }
            else { src = d.skip1; cs = d.SC1; coff = cb - d.SC0; }
#pragma unroll
            for (int i = 0; i < HR; ++i) {
                f32x4 v = {0.f, 0.f, 0.f, 0.f};
                if (hoff[i] >= 0) v = *reinterpret_cast<const f32x4*>(src + (size_t)hoff[i] * cs + coff + c4 * 4);
                h_st[i] = v;
            }
#pragma unroll
            for (int i = 0; i < BR; ++i) b_st[i] = *reinterpret_cast<const f32x4*>(swbase + (size_t)(32 * i) * SCin + cb);
        };
        if (s_begin < s_end) prefetch_skip(s_begin);
        for (int ch = s_begin; ch < s_end; ++ch) {
            __syncthreads();
            stage_halo(false);
            stage_b(0);
            __syncthreads();
            if (ch + 1 < s_end) prefetch_skip(ch + 1);
            mfma_tap(4, 0);                      // centre tap: the tile's own pixels
        }
    }
    conv_epilogue<WM, WN, MT, NT>(p, tm, n_tile, m_tile, slice, acc, Hs);
}

// =====================================================================================
// generic per-tap gather (1x1, strided 3x3)
// =====================================================================================
template <int WM, int WN, int MT, int NT>
__global__ __launch_bounds__(256, 3) void conv_gather_f32_kernel(const ConvArgs p) {
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
    constexpr int AR = BM / 32, BR = BN / 32;  // rows per thread for the A / B tile copies
    __shared__ __attribute__((aligned(16))) float As[BM * LDT];
    __shared__ __attribute__((aligned(16))) float Bs[BN * LDT];

    const ddnm_conv_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;
    const int tile_id = xcd_swizzle(blockIdx.x, gridDim.x);
    const int n_tile = tile_id % p.n_tiles, m_tile = tile_id / p.n_tiles;
    const int slice = blockIdx.y;
    const TileMap tm = make_tilemap<BM>(p, m_tile);
    const int img = tm.img;

    const int c4 = tid & 7, row0 = tid >> 3;
    int iy0[AR], ix0[AR];
#pragma unroll
    for (int i = 0; i < AR; ++i) {
        int oy, ox;
        tm.pixel(row0 + 32 * i, oy, ox);
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
            if (has_gn && (a_valid & (1u << i))) v = gn_act(v, gsc, gsh, d.gn_silu);
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

    const int nchunks = p.Cin / KC;
    const int c_begin = (int)((long)nchunks * slice / p.ksplit), c_end = (int)((long)nchunks * (slice + 1) / p.ksplit);
    const int it_begin = c_begin * p.ntaps, it_end = c_end * p.ntaps;
    const int frag_off = (lane & 31) * LDT + (lane >> 5) * 4;
    const float* a_frag = As + (wm * MT * 32) * LDT + frag_off;
    const float* b_frag = Bs + (wn * NT * 32) * LDT + frag_off;

    if (it_begin < it_end) prefetch(it_begin);
    for (int it = it_begin; it < it_end; ++it) {
        __syncthreads();           // all waves finished reading the previous tile
        stage_to_lds();
        __syncthreads();
        if (it + 1 < it_end) prefetch(it + 1);
        mfma_tile_step<MT, NT>(a_frag, b_frag, acc);
    }
    conv_epilogue<WM, WN, MT, NT>(p, tm, n_tile, m_tile, slice, acc, As);
}

// =====================================================================================
// host side: plan + launch
// =====================================================================================
struct ConvPlan {
    int tile;       // 1: 128x128, 2: 64x64, 3: 128x32
    int BM, BN;
    bool halo;
    int ksplit;
    int TW, TW_log2, tiles_x;
};

static bool make_plan(const ddnm_conv_desc* d, ConvPlan* pl) {
    const int HWo = d->Ho * d->Wo;
    const int nchunks = (d->C0 + d->C1) / KC;
    int tile = d->tile;
    if (!tile) {
        if (d->Cout <= 32) {
            tile = (HWo % 128 == 0) ? 3 : 2;
        } else {
            tile = 2;
            if (HWo % 128 == 0 && d->Cout % 128 == 0) {
                // 128x128 whenever it can still occupy the chip (directly or through split-K)
                const long tiles = (long)d->B * (HWo / 128) * (d->Cout / 128);
                if (tiles * nchunks >= 256) tile = 1;
            }
        }
    }
    pl->tile = tile;
    pl->BM = tile == 2 ? 64 : 128;
    pl->BN = tile == 1 ? 128 : (tile == 2 ? 64 : 32);
    if (HWo % pl->BM) return false;
    pl->halo = d->ksize == 3 && d->stride == 1 && d->pad == 1 && d->Ho == d->Hin && d->Wo == d->Win;
    // 2-D output tile: width 32 keeps the 32 rows of one MFMA tile on consecutive halo rows
    // (conflict-free ds_read_b128); narrower images fall back to 16 / 8
    int tw = 32;
    while (tw > 8 && (d->Wo % tw || d->Ho % (pl->BM / tw) || pl->BM / tw < 1)) tw >>= 1;
    const bool two_d = (d->Wo % tw == 0) && (d->Ho % (pl->BM / tw) == 0);
    if (pl->halo && !two_d) pl->halo = false;
    pl->TW = tw;
    pl->TW_log2 = tw == 32 ? 5 : (tw == 16 ? 4 : 3);
    pl->tiles_x = two_d ? d->Wo / tw : 0;
    // split-K over channel chunks when the tile grid cannot fill the chip
    const long tiles = (long)d->B * (HWo / pl->BM) * ((d->Cout + pl->BN - 1) / pl->BN);
    int ks = 1;
    if (!d->out_nchw && d->Cout % 4 == 0 && tiles < 192) {
        ks = (int)((512 + tiles - 1) / tiles);
        if (ks > nchunks) ks = nchunks;
        if (ks > 16) ks = 16;
        if (ks < 1) ks = 1;
    }
    pl->ksplit = ks;
    return true;
}

extern "C" int ddnm_conv2d_f32_tile_n(const ddnm_conv_desc* d) {
    ConvPlan pl;
    if (!d || !make_plan(d, &pl)) return DDNM_E_SHAPE;
    return pl.BN;
}

extern "C" int ddnm_conv2d_f32_fuses_skip(const ddnm_conv_desc* d) {
    ConvPlan pl;
    if (!d || !make_plan(d, &pl)) return 0;
    return pl.halo && !d->ups ? 1 : 0;
}

extern "C" int ddnm_conv2d_f32_stats_tiles(const ddnm_conv_desc* d) {
    ConvPlan pl;
    if (!d || !make_plan(d, &pl)) return DDNM_E_SHAPE;
    if (d->out_nchw) return 0;                          // NCHW launches do not emit statistics
    if (pl.ksplit > 1) return splitk_stats_tiles(d);    // ... split-K launches emit them from the reduction pass
    return d->Ho * d->Wo / pl.BM;
}

extern "C" int64_t ddnm_conv2d_f32_workspace_floats(const ddnm_conv_desc* d) {
    ConvPlan pl;
    if (!d || !make_plan(d, &pl)) return DDNM_E_SHAPE;
    return pl.ksplit > 1 ? (int64_t)pl.ksplit * d->B * d->Ho * d->Wo * d->Cout : 0;
}

template <int WM, int WN, int MT, int NT>
static void launch_conv(bool halo, dim3 grid, hipStream_t s, const ConvArgs& p) {
    if (halo)
        hipLaunchKernelGGL((conv3x3_halo_f32_kernel<WM, WN, MT, NT>), grid, dim3(256), 0, s, p);
    else
        hipLaunchKernelGGL((conv_gather_f32_kernel<WM, WN, MT, NT>), grid, dim3(256), 0, s, p);
}

extern "C" int ddnm_conv2d_f32(const ddnm_conv_desc* d, void* stream) {
    if (!d || !d->src0 || !d->weight || !d->out) return DDNM_E_BADARG;
    if (d->B <= 0 || d->Cout <= 0 || d->Ho <= 0 || d->Wo <= 0) return DDNM_E_BADARG;
    if (!conv_sizes_addressable(d)) return DDNM_E_SHAPE;
    if (d->C0 <= 0 || d->C0 % KC || d->C1 % KC || (d->C1 > 0 && !d->src1)) return DDNM_E_SHAPE;
    if ((d->ksize != 1 && d->ksize != 3) || (d->stride != 1 && d->stride != 2)) return DDNM_E_SHAPE;
    if (d->gn_scale && !d->gn_shift) return DDNM_E_BADARG;
    if (d->ups && ((d->Hin | d->Win) & 1)) return DDNM_E_SHAPE;
    if (d->out_nchw && d->res) return DDNM_E_SHAPE;
    if (d->res_ups && ((d->Ho | d->Wo) & 1)) return DDNM_E_SHAPE;
    ConvPlan pl;
    if (!make_plan(d, &pl)) return DDNM_E_SHAPE;
    if (d->skip0) {
        if (!pl.halo || d->ups || !d->skip_weight || d->SC0 <= 0 || d->SC0 % KC || d->SC1 % KC ||
            (d->SC1 > 0 && !d->skip1))
            return DDNM_E_SHAPE;
    }
    if (pl.ksplit > 1) {
        const int64_t need = (int64_t)pl.ksplit * d->B * d->Ho * d->Wo * d->Cout;
        if (!d->workspace || d->workspace_floats < need) {
            if (d->stats_out) return DDNM_E_BADARG;       // the caller sized stats_out for the split plan
            pl.ksplit = 1;                                // no scratch: run unsplit
        }
    }
    if (d->stats_out && d->out_nchw) return DDNM_E_SHAPE;                     // see ddnm_conv2d_f32_stats_tiles
    ConvArgs p;
    p.d = *d;
    p.Cin = d->C0 + d->C1;
    p.ntaps = d->ksize * d->ksize;
    p.Hs = d->ups ? d->Hin / 2 : d->Hin;
    p.Ws = d->ups ? d->Win / 2 : d->Win;
    p.m_tiles = d->B * (d->Ho * d->Wo / pl.BM);
    p.n_tiles = (d->Cout + pl.BN - 1) / pl.BN;
    p.TW = pl.TW;
    p.TW_log2 = pl.TW_log2;
    p.tiles_x = pl.tiles_x;
    p.ksplit = pl.ksplit;
    p.ws = d->workspace;
    dim3 grid(p.m_tiles * p.n_tiles, pl.ksplit);
    hipStream_t s = (hipStream_t)stream;
    (void)hipGetLastError();      // drop stale errors of unrelated runtime calls
    switch (pl.tile) {
        case 1: launch_conv<2, 2, 2, 2>(pl.halo, grid, s, p); break;
        case 2: launch_conv<2, 2, 1, 1>(pl.halo, grid, s, p); break;
        case 3: launch_conv<4, 1, 1, 1>(pl.halo, grid, s, p); break;
        default: return DDNM_E_BADARG;
    }
    DDNM_LAUNCH_CHECK();
    if (pl.ksplit > 1) return launch_splitk_reduce(p, s);
    return 0;
}
