// 3x3 implicit-GEMM convolution on v_mfma_f32_32x32x16_f16 (fp16 operands, fp32 accumulate), gfx950.
//
// Precision contract of the reference's ImageNet path: conv weights and torso activations are fp16,
// GroupNorm / softmax / embeddings fp32 (guided_diffusion/fp16_util.py:15-22, nn.py:17-19,
// unet.py:352).  Here activations stay fp32 in HBM; they are rounded to fp16 only while the
// (GroupNorm+swish'd) halo is staged into LDS, weights are fp16 in HBM, accumulation is fp32 and the
// output (+ bias + residual) is written in fp32 -- i.e. never less precise than the reference.
// Keeping HBM tensors fp32 leaves every other kernel of the engine untouched; at >= 500 TFLOP/s the
// 256-channel 256x256 layers remain compute-bound (0.3 ms of MFMA vs 0.2 ms of HBM traffic).
//
// Structure = the fp32 halo kernel scaled for a 16x faster matrix pipe:
//   workgroup 512 threads = 8 waves (4 x 2), wave tile 64 x 64 (2 x 2 MFMA tiles), block tile
//   256 pixels (8 x 32 patch, halo 10 x 34) x 128 output channels, K chunk = 64 channels;
//   LDS: halo [340][64+8] fp16 double-buffered (2 x 49 KB; row pitch 144 B = 36 dwords keeps ds_read_b128
//   conflict-free, one read = 8 k-values of a row) + THREE weight tiles [128][64] fp16 (3 x 16 KB) that arrive by
//   LDS-DMA (buffer_load_dwordx4 ... lds: no staging registers, no ds_write pass): unpadded lane-linear image, bank
//   conflicts removed by an XOR swizzle of the 16-byte piece index with (row >> 1) & 7 applied to the per-lane SOURCE
//   address and to the fragment read address; the tile of tap+2 is requested while tap is multiplied (an L2 round
//   trip is longer than one tap of MFMAs: with register staging one tap ahead the loads cost 20 % of the kernel);
//   per tap and wave: 16 (split form: 24) MFMAs between barriers.
#include "conv_common.h"

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

constexpr int KC16 = 64;            // halfs per LDS row of one chunk
constexpr int F16_BM = 256;         // pixels per block tile
constexpr int F16_KS_TARGET = 256;  // split-K: workgroups a low-resolution launch is spread over.  ONE workgroup fits a CU
                                    // (151 KB of LDS), so 256 = a single round; 512 (the fp32 kernel's figure, two
                                    // workgroups per CU) doubles the fp32 slab traffic for nothing: 32^2 / 16^2 layers
                                    // of the celeba UNet at B = 8 run 12-29 % faster with 256
constexpr int LDH = KC16 + 8;       // LDS row pitch in halfs (144 B)

// SRC16 = the activation operand is already fp16 in HBM (written by ddnm_gn_apply_f16: GroupNorm affine +
// swish applied ONCE per element instead of once per (output-channel tile x halo overlap) inside this kernel,
// where the v_exp/v_rcp work of a 1024-output-channel layer is repeated 10x and costs 25 % of the kernel).
//
// SPLIT = fp32-grade products on the fp16 matrix pipe (the celeba `Model`, whose reference runs in fp32): every fp32
// operand value v is carried as TWO fp16 numbers hi = rn16(v), lo = rn16(v - hi) (|v - hi - lo| <= 2^-22 |v|) and a
// product is hi*hi' + hi*lo' + lo*hi' -- three MFMAs whose fp16 x fp16 products are exact in the fp32 accumulator;
// the dropped lo*lo' term is 2^-22 relative.  A chunk is then 32 channels: an LDS row holds [hi 32 | lo 32] halfs,
// i.e. the same 128 + 16 bytes as a 64-channel fp16 row, the weights arrive packed the same way ([hi 32 | lo 32]
// per (row, tap, chunk), pre-scaled by a power of two so that `lo` stays a normal fp16 number) and
// ddnm_conv_desc::acc_scale undoes the scaling in the epilogue.
//
// ASCALE (split form only) = the launch carries an operand bound (ddnm_conv_desc::amax_in): raw operands -- the main
// operand of a launch without GroupNorm (Upsample convolution) and the fused shortcut's input -- are multiplied by a
// per-launch, per-image power of two while they are split, and the accumulator by its inverse (conv_common.h::
// s16_operand_scale).  Launches whose operands are all GroupNorm'd run the ASCALE = false instance (no multiply).
template <int WM, int WN, int MT, int NT, bool SRC16, bool SPLIT = false, bool ASCALE = false>
__global__ __launch_bounds__(WM * WN * 64, 1) void conv3x3_halo_f16_kernel(const ConvArgs p) {
    static_assert(!(SPLIT && SRC16), "the split form reads fp32 activations");
    static_assert(SPLIT || !ASCALE, "operand scaling belongs to the split form");
    constexpr int NTHREADS = WM * WN * 64;
    constexpr int BM = WM * MT * 32, BN = WN * NT * 32;
    constexpr int MAXH = BM == 512 ? 612 : (BM == 256 ? 340 : (BM == 128 ? 204 : 136));
    constexpr int KCH = SPLIT ? KC16 / 2 : KC16;                  // CHANNELS per chunk (an LDS row is always KC16 halfs)
    constexpr int WE = SPLIT ? 2 : 1;                             // weight halfs per channel
    constexpr int HVEC = SRC16 ? 8 : 4;                           // channels per 16-byte global load
    constexpr int HCOLS = KCH / HVEC;                             // loads per halo row of one chunk
    constexpr int HROWS_PER_PASS = NTHREADS / HCOLS;
    constexpr int HR = (MAXH + HROWS_PER_PASS - 1) / HROWS_PER_PASS;
    static_assert(KC16 == 64, "an LDS weight row is 128 bytes");
    constexpr int BCOLS = 8;                                      // 16-byte pieces per weight row of one chunk
    constexpr int BROWS_PER_PASS = NTHREADS / BCOLS;
    constexpr int BR = BN / BROWS_PER_PASS;                       // LDS-DMA instructions per wave and weight tile
    static_assert(BR >= 1 && BN % BROWS_PER_PASS == 0, "weight tile / thread mapping");
    constexpr int NWB = 3;                                        // weight tiles in LDS (tap, tap+1, tap+2)
    constexpr int WTILE = BN * 128;                               // bytes of one weight tile
    constexpr int HBYTES = 2 * MAXH * LDH * 2;                    // halo, double-buffered (see main loop)
    // ONE shared object (a second one makes the compiler drain the LDS-DMA queue in front of every fragment read)
    __shared__ __attribute__((aligned(1024))) char lds_all[NWB * WTILE + HBYTES + WM * BN * 2 * 4];
    char* const Bs = lds_all;
    _Float16* const Hs = reinterpret_cast<_Float16*>(lds_all + NWB * WTILE);
    float* const stat_lds = reinterpret_cast<float*>(lds_all + NWB * WTILE + HBYTES);

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
    // operand-range guard (scalar loads: nothing here touches the vmcnt bookkeeping of the main loop)
    float ascale = 1.f, epi_scale = SPLIT ? d.acc_scale : 1.f;
    if constexpr (ASCALE) {
        float inv;
        s16_operand_scale(d.amax_in, img, /*down_only=*/d.gn_scale != nullptr, ascale, inv);
        epi_scale = d.acc_scale * inv;
    }

    // ---- halo loader mapping: thread -> (16-byte column hc of HCOLS, halo rows prow + HROWS_PER_PASS*i)
    const int hc = tid % HCOLS, prow = tid / HCOLS;
    int hoff[HR];
#pragma unroll
    for (int i = 0; i < HR; ++i) {
        const int row = prow + HROWS_PER_PASS * i;
        const int hy = row / HWd, hx = row - hy * HWd;
        const int iy = tm.ty0 - 1 + hy, ix = tm.tx0 - 1 + hx;
        const bool ok = row < NP && (unsigned)iy < (unsigned)d.Hin && (unsigned)ix < (unsigned)d.Win;
        const int sy = d.ups ? (iy >> 1) : iy, sx = d.ups ? (ix >> 1) : ix;
        hoff[i] = ok ? (img * p.Hs + sy) * p.Ws + sx : -1;
    }
    // ---- weight tile of (chunk, tap) -> Bs[buf] by LDS-DMA: one instruction moves 8 rows x 128 B; lane ->
    // (row = 8*g + lane/8, piece lane%8) and the piece it FETCHES is piece ^ swizzle(row), so the linear image holds the
    // swizzled layout.  Rows of wave w: 8w + lrow + 64 j, so (row >> 1) & 7 = 4 (w & 1) + (lrow >> 1) for every j.
    const int lrow = lane >> 3, lpiece = lane & 7;
    const int wswz = (((wave & 1) << 2) | (lrow >> 1));
    const unsigned w_rowlen = 9u * (unsigned)p.Cin * WE * 2u;                 // bytes per output-channel row
    const __amdgpu_buffer_rsrc_t r_w = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(reinterpret_cast<const void*>(d.weight)), 0, (unsigned)p.n_tiles * BN * w_rowlen, 0x00020000);
    const unsigned w_voff = (unsigned)(n_tile * BN + wave * 8 + lrow) * w_rowlen + (unsigned)((lpiece ^ wswz) * 16);
    auto issue_w = [&](int chunk, int tap, int buf) {
        char* dst = Bs + buf * WTILE + wave * 1024;
        const unsigned so = ((unsigned)tap * p.Cin + (unsigned)chunk * KCH) * WE * 2u;
#pragma unroll
        for (int j = 0; j < BR; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w, (__attribute__((address_space(3))) void*)(dst + j * (NTHREADS / 64) * 1024),
                                                     16, w_voff, so + (unsigned)j * (NTHREADS / 8) * w_rowlen, 0, 0);
    };
    // register-staged weights (fused shortcut only): thread -> (16-byte column c8 of 8, rows brow + BROWS_PER_PASS*i)
    const int c8 = tid % BCOLS, brow = tid / BCOLS;

    const int nchunks = p.Cin / KCH;
    const int c_begin = (int)((long)nchunks * slice / p.ksplit), c_end = (int)((long)nchunks * (slice + 1) / p.ksplit);

    uint4 h_st[HR];                                 // fp32 mode: 4 floats (bit-cast); fp16 mode: 8 halfs
    // named registers, not an array: the compiler kept `uint4 b_st[BR]` in memory (promoted to a 16 KB LDS array),
    // which turned every weight prefetch into load -> wait -> LDS -> barrier -> LDS -> LDS and exposed the full
    // global-load latency once per tap
    static_assert(BR == 1 || BR == 2 || BR == 4, "weight staging registers");
    uint4 b_st0 = {0u, 0u, 0u, 0u}, b_st1 = {0u, 0u, 0u, 0u}, b_st2 = {0u, 0u, 0u, 0u}, b_st3 = {0u, 0u, 0u, 0u};
    f32x4 gsc = {1.f, 1.f, 1.f, 1.f}, gsh = {0.f, 0.f, 0.f, 0.f};
    const bool has_gn = !SRC16 && d.gn_scale != nullptr;

    constexpr int HSPLIT = (HR + 1) / 2;            // row slots [0, HSPLIT) and [HSPLIT, HR) are loaded / staged separately
    // Halo loads are buffer loads with an out-of-range offset for padding / unused rows (the load returns zero): every
    // wave then issues EXACTLY i1 - i0 (+ 2 GroupNorm vectors with part 0) requests per call, which is what lets the
    // main loop leave them in flight across a barrier with a counted `s_waitcnt vmcnt(N)`.
    constexpr int ESZ = SRC16 ? 2 : 4;
    constexpr unsigned HOOB = 0x80000000u;          // every tensor here is < 2 GB (checked by the host)
    const __amdgpu_buffer_rsrc_t r_s0 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(reinterpret_cast<const void*>(d.src0)), 0, (unsigned)d.B * p.Hs * p.Ws * d.C0 * ESZ, 0x00020000);
    const __amdgpu_buffer_rsrc_t r_s1 = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(reinterpret_cast<const void*>(d.C1 > 0 ? d.src1 : d.src0)), 0,
        (unsigned)d.B * p.Hs * p.Ws * (d.C1 > 0 ? d.C1 : d.C0) * ESZ, 0x00020000);
    // `live` = false turns every request of the call into an out-of-range one (returns zero, no memory traffic): the last
    // chunk of a slice still issues the SAME number of requests per tap as every other chunk, so the `vmcnt` immediates of
    // the main loop are compile-time constants of a branch-free request stream (tests/test_isa_waits.py replays the
    // stream from the ISA and checks every immediate against it).  The GroupNorm vectors are fetched unconditionally
    // (from the weights when the launch has no GroupNorm, like the gather form).
    const float* const gn_sc_base = has_gn ? d.gn_scale + (size_t)img * p.Cin + hc * 4 : reinterpret_cast<const float*>(d.weight);
    const float* const gn_sh_base = has_gn ? d.gn_shift + (size_t)img * p.Cin + hc * 4 : reinterpret_cast<const float*>(d.weight);
    auto prefetch_halo_part = [&](int chunk, int i0, int i1, bool live = true) {
        const int cb = chunk * KCH;
        const bool first = cb < d.C0;
        const unsigned cs = first ? d.C0 : d.C1, coff = first ? cb : cb - d.C0;
        const __amdgpu_buffer_rsrc_t r_s = first ? r_s0 : r_s1;              // wave-uniform: a scalar select, no branch
#pragma unroll
        for (int i = 0; i < HR; ++i) {
            if (i < i0 || i >= i1) continue;
            const unsigned vo = (hoff[i] >= 0 && live) ? ((unsigned)hoff[i] * cs + hc * HVEC) * ESZ : HOOB;
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r_s, vo, coff * ESZ, 0);
            h_st[i] = uint4{v.x, v.y, v.z, v.w};
        }
        if constexpr (!SRC16) {
            if (i0 == 0) {
                gsc = *reinterpret_cast<const f32x4*>(gn_sc_base + (has_gn ? cb : 0));
                gsh = *reinterpret_cast<const f32x4*>(gn_sh_base + (has_gn ? cb : 0));
            }
        }
    };
    auto prefetch_halo = [&](int chunk) { prefetch_halo_part(chunk, 0, HR); };
    auto stage_halo_part = [&](int hbuf, int i0, int i1) {
#pragma unroll
        for (int i = 0; i < HR; ++i) {
            if (i < i0 || i >= i1) continue;
            const int row = prow + HROWS_PER_PASS * i;
            if (row < MAXH) {
                _Float16* dst = &Hs[hbuf * MAXH * LDH + row * LDH + hc * HVEC];
                if constexpr (SRC16) {
                    *reinterpret_cast<uint4*>(dst) = h_st[i];
                } else {
                    f32x4 v = __builtin_bit_cast(f32x4, h_st[i]);
                    if (has_gn && hoff[i] >= 0) v = gn_act(v, gsc, gsh, d.gn_silu);
                    if constexpr (SPLIT && ASCALE) {
                        split_store(dst, v, ascale);
                    } else if constexpr (SPLIT) {
                        split_store(dst, v);
                    } else {
                        half4 h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
                        *reinterpret_cast<half4*>(dst) = h;
                    }
                }
            }
        }
    };
    auto stage_b = [&](int buf) {          // register-staged weights (fused shortcut): the same swizzled image
        char* dst = Bs + buf * WTILE;
        auto put = [&](int r, const uint4& v) { *reinterpret_cast<uint4*>(dst + r * 128 + ((c8 ^ ((r >> 1) & 7)) << 4)) = v; };
        put(brow, b_st0);
        if constexpr (BR >= 2) put(brow + BROWS_PER_PASS, b_st1);
        if constexpr (BR == 4) {
            put(brow + 2 * BROWS_PER_PASS, b_st2);
            put(brow + 3 * BROWS_PER_PASS, b_st3);
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
        const int ty = m >> p.TW_log2, tx = m & (p.TW - 1);
        a_off[i] = (ty * HWd + tx) * LDH + (lane >> 5) * 8;
    }
    // B fragment of 16-byte piece q (+ lane >> 5): row n = wn*NT*32 + j*32 + (lane & 31), swizzled with (n >> 1) & 7 =
    // ((lane & 31) >> 1) & 7; the piece pair 2 q enters as an XOR of bits 5-6 of the byte address
    const int b_frag = ((wn * NT * 32 + (lane & 31)) * 128) + ((((lane >> 5) ^ (((lane & 31) >> 1) & 7))) << 4);

    auto mfma_tap = [&](int tap, int buf, int hbuf = 0) {
        const int ky = tap / 3, kx = tap - 3 * ky;
        const int tap_off = (ky * HWd + kx) * LDH + hbuf * MAXH * LDH;
        const char* bf = Bs + buf * WTILE;
        if constexpr (SPLIT) {
            // [hi 32 | lo 32] rows: k-step ks reads the hi fragment at ks*16 and the lo fragment 32 halfs behind it;
            // the three products of one (i, j) tile are spread over the loop so that no MFMA waits on its predecessor
#pragma unroll
            for (int ks = 0; ks < KC16 / 32; ++ks) {
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
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
            }
            return;
        }
#pragma unroll
        for (int ks = 0; ks < KC16 / 16; ++ks) {
            half8 a[MT], b[NT];
#pragma unroll
            for (int i = 0; i < MT; ++i) a[i] = *reinterpret_cast<const half8*>(Hs + a_off[i] + tap_off + ks * 16);
#pragma unroll
            for (int j = 0; j < NT; ++j) b[j] = *reinterpret_cast<const half8*>(bf + ((b_frag ^ (ks << 5)) + j * 32 * 128));
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
    };

    // Main loop, one barrier per tap:  [my pieces of W(s) landed | barrier: everybody's; the buffer of W(s-1) is free |
    // request W(s+2) into it | MFMA(s)].  Halo: double-buffered -- the next chunk's halo is loaded in two halves
    // (taps 0 and 3) and converted / GroupNorm'd / written to the OTHER halo buffer at taps 3 and 6, between MFMA
    // batches, so a chunk boundary costs nothing (re-staging it between two barriers used to be 28 % of the kernel).
    // Static priority for one of the two waves that share a SIMD (waves w and w + 4; MI355X_MICROARCH.md): the pair stops
    // competing for the matrix pipe in phase, the low-priority wave fills the other's barrier / fragment-read gaps.
    // Split form: +2.4 ... 2.9 % on the 256^2 layers, nothing elsewhere (tools/s16_probe.py vtime base prio_half); no
    // effect was ever measured for the fp16-operand form, which keeps the default.
    if (SPLIT && wave >= WM * WN / 2) __builtin_amdgcn_s_setprio(1);
    if (c_begin < c_end) {
        prefetch_halo(c_begin);
        const int last_step = (c_end - c_begin) * 9 - 1;
        // request of step q (clamped to the last one: the tail re-requests a tile nobody reads, which keeps the
        // `vmcnt` bookkeeping of the loop uniform)
        auto issue_step = [&](int q, int buf) {
            q = q < last_step ? q : last_step;
            const int ch = q / 9;
            issue_w(c_begin + ch, q - 9 * ch, buf);
        };
        issue_step(0, 0);
        issue_step(1, 1);
        stage_halo_part(0, 0, HR);
        int hb = 0, step0 = 0;
        for (int chunk = c_begin; chunk < c_end; ++chunk, step0 += 9) {
            const bool more = chunk + 1 < c_end;
            // unrolled: in a rolled loop the compiler guards the halo loads into registers (taps 0 / 3) with a full
            // `vmcnt(0)` in front of every tap's fragment reads, which would drain the two tiles in flight
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int cur = tap % NWB, step = step0 + tap;      // 9 % 3 == 0: W(step) sits in buffer tap % 3
                // in order behind W(step): W(step+1) = BR requests (+ halo loads into registers, waited for where used)
                // (raw s_barrier: __syncthreads() would drain the whole queue -- `vmcnt(0)` -- in front of it; lgkmcnt(0) =
                // this wave's halo ds_writes of taps 3 / 6 are in LDS before anybody can be past the barrier)
                // the halo loads of taps 0 / 3 (requested behind W(step+2) of those taps, i.e. younger than the tile awaited
                // here at the two following taps) stay in flight as well: they come from HBM and are not needed before
                // taps 3 / 6, where the compiler waits for their registers.  Every immediate is a compile-time constant of
                // a branch-free request stream; tests/test_isa_waits.py replays the stream from the ISA.
                constexpr int GNV = SRC16 ? 0 : 2;         // GroupNorm vectors requested with halo part 0
                if (tap == 1 || tap == 2) {
                    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(BR + HSPLIT + GNV) : "memory");
                } else if (tap == 4 || tap == 5) {
                    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(BR + HR - HSPLIT) : "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(BR) : "memory");
                }
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                issue_step(step + 2, cur >= 1 ? cur - 1 : NWB - 1);     // (step + 2) % 3: the buffer W(step - 1) just left
                // pin the issue order the counted waits assume: this tap's DMA requests first, then its halo requests
                // (the scheduler may not move anything across; the compiler barrier keeps IR passes from sinking the loads)
                asm volatile("" ::: "memory");
                __builtin_amdgcn_sched_barrier(0);
                // the next chunk's halo: requested at taps 0 / 3 (on the last chunk as out-of-range requests that return
                // zero: the request stream stays uniform), GroupNorm'd / split / written to the other buffer at taps 3 / 6
                const int nchunk = more ? chunk + 1 : chunk;
                if (tap == 0) prefetch_halo_part(nchunk, 0, HSPLIT, more);
                if (tap == 3) {
                    if (more) stage_halo_part(hb ^ 1, 0, HSPLIT);
                    prefetch_halo_part(nchunk, HSPLIT, HR, more);
                }
                if (tap == 6 && more) stage_halo_part(hb ^ 1, HSPLIT, HR);
                // keep the requests issued above in front of this tap's MFMAs: left alone, the scheduler sinks
                // them behind the MFMAs (VGPR pressure) and their latency lands on the barrier of every tap
                __builtin_amdgcn_sched_barrier(0);
                mfma_tap(tap, cur, hb);
            }
            hb ^= 1;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the tail requests
        __syncthreads();
        // the shortcut phase and the statistics epilogue below use halo buffer 0 / weight buffer 0
    }
    // ---- fused 1x1 shortcut (skip_connection of a ResBlock, unet.py:222,256): extra K chunks over the block's
    // raw input at the centre tap (see conv_igemm_f32.hip)
    if (d.skip0 != nullptr) {
        // split-K launches share the shortcut's chunks like the main ones (slice 0 alone would run 2-3x longer)
        const int SCin = d.SC0 + d.SC1, nsk_all = SCin / KCH;
        const int s_begin = (int)((long)nsk_all * slice / p.ksplit), s_end = (int)((long)nsk_all * (slice + 1) / p.ksplit);
        const _Float16* swbase = reinterpret_cast<const _Float16*>(d.skip_weight) + (size_t)(n_tile * BN + brow) * SCin * WE + c8 * 8;
        // the raw input is fp32: thread -> (float4 column sc of 16, interior pixels srow + 32*i), staged at the
        // pixel's halo position so that mfma_tap(4) (centre tap) reads it
        constexpr int SCOLS = KCH / 4;                             // float4 pieces per raw-input row of one chunk
        constexpr int SR = BM * SCOLS / NTHREADS;
        const int sc = tid % SCOLS, srow = tid / SCOLS;
        int soff[SR], sdst[SR];
#pragma unroll
        for (int i = 0; i < SR; ++i) {
            const int m = srow + (NTHREADS / SCOLS) * i;
            const int ty = m >> p.TW_log2, tx = m & (p.TW - 1);
            const int iy = tm.ty0 + ty, ix = tm.tx0 + tx;
            const int sy = d.ups ? (iy >> 1) : iy, sx = d.ups ? (ix >> 1) : ix;
            soff[i] = (img * p.Hs + sy) * p.Ws + sx;
            sdst[i] = ((ty + 1) * HWd + tx + 1) * LDH + sc * 4;
        }
        f32x4 s_st[SR];
        auto prefetch_skip = [&](int ch) {
            const int cb = ch * KCH;
            const float* src;
            int cs, coff;
            if (cb < d.SC0) { src = d.skip0; cs = d.SC0; coff = cb; }
            else { src = d.skip1; cs = d.SC1; coff = cb - d.SC0; }
#pragma unroll
            for (int i = 0; i < SR; ++i) s_st[i] = *reinterpret_cast<const f32x4*>(src + (size_t)soff[i] * cs + coff + sc * 4);
            b_st0 = *reinterpret_cast<const uint4*>(swbase + cb * WE);
            if constexpr (BR >= 2) b_st1 = *reinterpret_cast<const uint4*>(swbase + ((size_t)BROWS_PER_PASS * SCin + cb) * WE);
            if constexpr (BR == 4) {
                b_st2 = *reinterpret_cast<const uint4*>(swbase + ((size_t)(2 * BROWS_PER_PASS) * SCin + cb) * WE);
                b_st3 = *reinterpret_cast<const uint4*>(swbase + ((size_t)(3 * BROWS_PER_PASS) * SCin + cb) * WE);
            }
        };
        if (s_begin < s_end) prefetch_skip(s_begin);
        for (int ch = s_begin; ch < s_end; ++ch) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < SR; ++i) {
                const f32x4 v = s_st[i];
                if constexpr (SPLIT && ASCALE) {
                    split_store(&Hs[sdst[i]], v, ascale);
                } else if constexpr (SPLIT) {
                    split_store(&Hs[sdst[i]], v);
                } else {
                    half4 h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
                    *reinterpret_cast<half4*>(&Hs[sdst[i]]) = h;
                }
            }
            stage_b(0);
            __syncthreads();
            if (ch + 1 < s_end) prefetch_skip(ch + 1);
            mfma_tap(4, 0);
        }
    }
    conv_epilogue<WM, WN, MT, NT, (MT * NT <= 4)>(p, tm, n_tile, m_tile, slice, acc, stat_lds, epi_scale);
}

// ---------------------------------------------------------------------------------------------
struct PlanF16 {
    int BM, TW, TW_log2, tiles_x, ksplit;
};

// kch = channels per K chunk: KC16 for fp16 operands, KC16 / 2 for the split (hi | lo) form
static bool plan_f16(const ddnm_conv_desc* d, PlanF16* pl, int kch = KC16) {
    const int HWo = d->Ho * d->Wo;
    const int Cin = d->C0 + d->C1;
    if (d->ksize != 3 || d->stride != 1 || d->pad != 1 || d->Ho != d->Hin || d->Wo != d->Win) return false;
    if (Cin % kch || d->C0 % kch || d->Cout % 128 || d->out_nchw) return false;
    {   // the loader addresses both sources with 32-bit buffer offsets and an out-of-range sentinel for padding
        const int64_t px = (int64_t)d->B * (d->ups ? d->Hin / 2 : d->Hin) * (d->ups ? d->Win / 2 : d->Win);
        const int64_t cmax = d->C0 > d->C1 ? d->C0 : d->C1;
        if (px * cmax * (d->src_f16 ? 2 : 4) >= (int64_t)1 << 31) return false;
    }
    pl->BM = F16_BM;
    if (HWo % pl->BM) return false;
    int tw = 32;
    while (tw > 8 && (d->Wo % tw || d->Ho % (pl->BM / tw))) tw >>= 1;
    if (d->Wo % tw || d->Ho % (pl->BM / tw)) return false;
    pl->TW = tw;
    pl->TW_log2 = tw == 32 ? 5 : (tw == 16 ? 4 : 3);
    pl->tiles_x = d->Wo / tw;
    const long tiles = (long)d->B * (HWo / pl->BM) * (d->Cout / 128);
    int ks = 1;
    const int nchunks = Cin / kch;
    if (tiles < 192) {
        ks = (int)((F16_KS_TARGET + tiles - 1) / tiles);
        if (ks > nchunks) ks = nchunks;
        if (ks > 16) ks = 16;
    }
    pl->ksplit = ks;
    return true;
}

extern "C" int ddnm_conv3x3_f16_supported(const ddnm_conv_desc* d) {
    PlanF16 pl;
    return d && plan_f16(d, &pl) ? 1 : 0;
}

extern "C" int64_t ddnm_conv3x3_f16_workspace_floats(const ddnm_conv_desc* d) {
    PlanF16 pl;
    if (!d || !plan_f16(d, &pl)) return DDNM_E_SHAPE;
    return pl.ksplit > 1 ? (int64_t)pl.ksplit * d->B * d->Ho * d->Wo * d->Cout : 0;
}

extern "C" int ddnm_conv3x3_f16_stats_tiles(const ddnm_conv_desc* d) {
    PlanF16 pl;
    if (!d || !plan_f16(d, &pl)) return DDNM_E_SHAPE;
    return pl.ksplit > 1 ? splitk_stats_tiles(d) : d->Ho * d->Wo / pl.BM;
}

static int run_f16(const ddnm_conv_desc* d, void* stream, bool split) {
    const int kch = split ? KC16 / 2 : KC16;
    if (!d || !d->src0 || !d->weight || !d->out) return DDNM_E_BADARG;
    if (split && (d->src_f16 || !(d->acc_scale > 0.f))) return DDNM_E_BADARG;
    // raw operands (no GroupNorm in front of the main operand, or a fused shortcut) need the operand bound: fp16 range
    if (split && !d->amax_in && (!d->gn_scale || d->skip0)) return DDNM_E_BADARG;
    if (d->B <= 0 || d->Cout <= 0 || d->Ho <= 0 || d->Wo <= 0) return DDNM_E_BADARG;
    if (!conv_sizes_addressable(d)) return DDNM_E_SHAPE;
    if (d->C1 > 0 && !d->src1) return DDNM_E_BADARG;
    if (d->gn_scale && !d->gn_shift) return DDNM_E_BADARG;
    if (d->src_f16 && d->gn_scale) return DDNM_E_BADARG;
    if (d->ups && ((d->Hin | d->Win) & 1)) return DDNM_E_SHAPE;
    if (d->res_ups && ((d->Ho | d->Wo) & 1)) return DDNM_E_SHAPE;
    PlanF16 pl;
    if (!plan_f16(d, &pl, kch)) return DDNM_E_SHAPE;
    if (d->skip0) {
        if (d->ups || !d->skip_weight || d->SC0 <= 0 || d->SC0 % kch || d->SC1 % kch || (d->SC1 > 0 && !d->skip1))
            return DDNM_E_SHAPE;
    }
    if (pl.ksplit > 1) {
        const int64_t need = (int64_t)pl.ksplit * d->B * d->Ho * d->Wo * d->Cout;
        if (!d->workspace || d->workspace_floats < need) {
            if (d->stats_out) return DDNM_E_BADARG;
            pl.ksplit = 1;
        }
    }
    ConvArgs p;
    p.d = *d;
    p.Cin = d->C0 + d->C1;
    p.ntaps = 9;
    p.Hs = d->ups ? d->Hin / 2 : d->Hin;
    p.Ws = d->ups ? d->Win / 2 : d->Win;
    p.m_tiles = d->B * (d->Ho * d->Wo / pl.BM);
    p.n_tiles = d->Cout / 128;
    p.TW = pl.TW;
    p.TW_log2 = pl.TW_log2;
    p.tiles_x = pl.tiles_x;
    p.ksplit = pl.ksplit;
    p.ws = d->workspace;
    hipStream_t s = (hipStream_t)stream;
    // launches of >= 2 tiles per CU: the persistent form (conv_s16_persist.hip; bit-identical results), unless the caller
    // asks for the one-tile kernel (ddnm_conv_desc::flags & DDNM_CONV_ONE_TILE: A/B timing and the bit-identity tests)
    if (split && !(d->flags & DDNM_CONV_ONE_TILE) && conv3x3_s16_persist_eligible(p)) return conv3x3_s16_persist_launch(p, s);
    const dim3 grid(p.m_tiles * p.n_tiles, pl.ksplit);
    if (split && d->amax_in) { DDNM_LAUNCH((conv3x3_halo_f16_kernel<4, 2, 2, 2, false, true, true>), grid, dim3(512), 0, s, p); }
    else if (split) { DDNM_LAUNCH((conv3x3_halo_f16_kernel<4, 2, 2, 2, false, true>), grid, dim3(512), 0, s, p); }
    else if (d->src_f16) { DDNM_LAUNCH((conv3x3_halo_f16_kernel<4, 2, 2, 2, true>), grid, dim3(512), 0, s, p); }
    else { DDNM_LAUNCH((conv3x3_halo_f16_kernel<4, 2, 2, 2, false>), grid, dim3(512), 0, s, p); }
    if (pl.ksplit > 1) return launch_splitk_reduce(p, s);
    return 0;
}

extern "C" int ddnm_conv3x3_f16_f32(const ddnm_conv_desc* d, void* stream) { return run_f16(d, stream, false); }

// ---- split form: fp32 tensors, fp32-grade products as three fp16 MFMAs (see the kernel's header comment)
extern "C" int ddnm_conv3x3_s16_f32(const ddnm_conv_desc* d, void* stream) { return run_f16(d, stream, true); }

extern "C" float ddnm_conv3x3_s16_act_scale(void) { return DDNM_S16_ASCALE; }

// 1: this launch would run the persistent form (conv_s16_persist.hip): >= 2 tiles per CU, no split-K (profiling / bench labels)
extern "C" int ddnm_conv3x3_s16_persistent(const ddnm_conv_desc* d) {
    PlanF16 pl;
    if (!d || d->src_f16 || (d->flags & DDNM_CONV_ONE_TILE) || !plan_f16(d, &pl, KC16 / 2) || pl.ksplit != 1) return 0;
    ConvArgs p;
    p.d = *d;
    p.Cin = d->C0 + d->C1;
    p.m_tiles = d->B * (d->Ho * d->Wo / pl.BM);
    p.n_tiles = d->Cout / 128;
    p.TW = pl.TW;
    p.ksplit = 1;
    return conv3x3_s16_persist_eligible(p) ? 1 : 0;
}

extern "C" int ddnm_conv3x3_s16_supported(const ddnm_conv_desc* d) {
    PlanF16 pl;
    return d && !d->src_f16 && plan_f16(d, &pl, KC16 / 2) ? 1 : 0;
}

extern "C" int64_t ddnm_conv3x3_s16_workspace_floats(const ddnm_conv_desc* d) {
    PlanF16 pl;
    if (!d || !plan_f16(d, &pl, KC16 / 2)) return DDNM_E_SHAPE;
    return pl.ksplit > 1 ? (int64_t)pl.ksplit * d->B * d->Ho * d->Wo * d->Cout : 0;
}

extern "C" int ddnm_conv3x3_s16_stats_tiles(const ddnm_conv_desc* d) {
    PlanF16 pl;
    if (!d || !plan_f16(d, &pl, KC16 / 2)) return DDNM_E_SHAPE;
    return pl.ksplit > 1 ? splitk_stats_tiles(d) : d->Ho * d->Wo / pl.BM;
}
