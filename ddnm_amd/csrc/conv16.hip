// fp16-activation implicit-GEMM convolution for the ADM `use_fp16` torso (guided_diffusion/unet.py:619-625,
// fp16_util.py:15-22), gfx950.  Second-generation kernel of the fp16 path:
//
//   * every tensor it touches in HBM is fp16 NHWC: the operand was normalised / activated ONCE by
//     ddnm_gn_apply_h16 (act16.hip), the output (+ bias + residual) is written as fp16 together with the fp32
//     GroupNorm partials of the ROUNDED values, so the consumer never re-reads it for statistics;
//   * block tile 256 (or 128) pixels x 256 output channels, 8 waves as 2 (pixels) x 4 (channels), wave tile
//     128 x 64 = 4 x 2 MFMA tiles of v_mfma_f32_32x32x16_f16: 6 ds_read_b128 per 8 MFMAs (the first-generation
//     64 x 64 wave tile needed 8 per 8 and was LDS-read bound);
//   * both operands go HBM/L2 -> LDS with buffer_load_dwordx4 ... lds (no staging registers, no ds_write pass):
//     the LDS image is lane-linear [row][64 channels = 128 B]; bank conflicts of the fragment reads are removed
//     by an XOR swizzle of the 16-byte piece index with (row >> 1) & 7, applied to the per-lane SOURCE address
//     and to the read address (the destination of an LDS-DMA cannot be permuted);
//   * zero padding and ragged tiles are out-of-range buffer offsets (the load returns zero) instead of branches;
//   * MFMA operands are swapped (A = weights, B = pixels): a lane then owns ONE pixel and 4 consecutive
//     channels per accumulator quad, which makes the epilogue's LDS transposition ds_write_b64 / ds_read_b128
//     and every global store a full 16 bytes per lane (8 lanes = one 128-byte line).
//
// K loop: chunk = 64 input channels; per chunk the (TH+2) x (TW+2) halo is resident (double-buffered, the next
// chunk's halo is requested at tap 0) and the 9 taps read it at shifted rows; the [256][64] weight tile of the
// next tap is requested right after the barrier that frees its buffer -- one barrier per 32 MFMAs per wave.
// 1x1 convolutions / the im2col'ed 8x8 level run the same loop with one tap and a flat row mapping.
#include "conv_common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_ptr_t;

struct Conv16Args {
    ddnm_conv16_desc d;
    int TW, TW_log2, tiles_x, tiles_per_img;   // 3x3: 2-D patch TH x TW of one image; 1x1: unused
    int m_tiles, n_tiles, ksplit, M;           // M = B*H*W output pixels
    int Hs, Ws;                                // source resolution (H/2 when ups)
    int wmajor;                                // workgroup order inside a slice: 1 = pixel tiles fastest (weight-heavy launch)
    int slab16;                                // split-K partial slabs in fp16 (see the epilogue) instead of fp32
};

constexpr int C16_KC = 64;          // channels per chunk (= 128 bytes per LDS row)
constexpr int C16_BN = 256;
constexpr int C16_ROWB = 128;       // LDS row pitch in bytes
constexpr int C16_EPITCH = 144;     // epilogue staging pitch (bytes) per pixel of a wave's 64-channel slice

// WNW = waves along the output-channel axis: 4 -> 2 x 4 waves, wave tile (MT*32) x 64, block (MT*64) x 256;
//                                             1 -> 8 x 1 waves, wave tile 32 x 32, block 256 x 32 (small-Cout output conv)
template <int TAPS, int MT, int WNW>
struct C16Geom {
    static constexpr int WMW = 8 / WNW, NT = WNW == 4 ? 2 : 1;
    static constexpr int BM = WMW * MT * 32, BN = WNW * NT * 32;
    // rows of one halo buffer, rounded up to whole 8-row (1 KB) LDS-DMA pieces
    static constexpr int HROWS = TAPS == 9 ? (BM == 256 ? 344 : 208) : BM;
    static constexpr int HGROUPS = HROWS / 8;
    static constexpr int HG_PER_WAVE = (HGROUPS + 7) / 8;
    static constexpr int HBYTES = HROWS * C16_ROWB;
    static constexpr int WBYTES = BN * C16_ROWB;                        // 32 KB (4 KB)
    // buffers in flight: the 128-pixel kernels have LDS to spare and run where the weights stream through L2 from HBM
    // (low resolutions: every weight byte is used by a handful of pixel tiles and a step's MFMAs are shorter than an L2
    // round trip), so they keep TWO weight tiles ahead of the one being multiplied; the 1-tap GEMM form needs a new
    // pixel tile per step as well
    // ... and the 64-pixel GEMM tile (1x1 convolutions of the 8^2 .. 32^2 attention blocks, the im2col'ed 8^2 level:
    // <= 16 steps per workgroup, every one a cold weight tile) keeps THREE tiles ahead: those launches are chains of
    // HBM round trips, and their rate is (tiles in flight) / latency
    static constexpr int NWB = (TAPS == 1 && MT == 1 && WNW == 4) ? 4 : ((MT == 2 && WNW == 4) ? 3 : 2);
    static constexpr int NHB = TAPS == 1 ? (MT == 1 ? 4 : (MT == 2 ? 3 : 2)) : 2;
    static constexpr int LDS_TILES = NHB * HBYTES + NWB * WBYTES;
    static constexpr int LDS_MAIN = LDS_TILES + (TAPS == 9 ? 512 : 0);  // + GroupNorm scale | shift of one 64-channel chunk
    static constexpr int LDS_EPI = WNW == 4 ? 8 * (MT * 32) * C16_EPITCH + 2 * C16_BN * 2 * 4 : 0;
    static constexpr int LDS_BYTES = LDS_MAIN > LDS_EPI ? LDS_MAIN : LDS_EPI;
};

// LDS-DMA of 16 bytes per lane through a raw buffer descriptor: per-lane 32-bit byte offset + wave-uniform SGPR
// offset; an offset beyond the descriptor's size returns ZERO (that is how padding and ragged rows are fetched).
__device__ __forceinline__ void bload16(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, char* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t*)lds_wave_base, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
constexpr unsigned C16_OOB = 0x80000000u;      // every tensor here is < 2 GB

template <int TAPS, int MT, int WNW>
__global__ __launch_bounds__(512) void conv16_kernel(const Conv16Args p) {
    using G = C16Geom<TAPS, MT, WNW>;
    constexpr int BM = G::BM, NT = G::NT, BN = G::BN;
    __shared__ __attribute__((aligned(1024))) char lds[G::LDS_BYTES];     // ONE shared object (keeps the DMA pipeline)
    char* const Hb = lds;
    char* const Wb = lds + G::NHB * G::HBYTES;

    const ddnm_conv16_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = WNW == 4 ? wave >> 2 : wave, wn = WNW == 4 ? wave & 3 : 0;
    const int kh = lane >> 5;
    // Workgroup -> (slice, pixel tile, channel tile).  Consecutive `q` share an XCD (= one L2): the split-K slice is
    // the slowest index, so an XCD reads only ITS slices' channel range of both operands; inside a slice the
    // operand that is larger for this launch is the one NOT replicated over XCDs (weight-heavy low-resolution
    // layers: the pixel tiles of one weight stream sit on one XCD; the old order made every XCD read every weight).
    const int q = xcd_swizzle(blockIdx.x, gridDim.x);
    const int tiles_mn = p.m_tiles * p.n_tiles;
    const int slice = q / tiles_mn, t_mn = q - slice * tiles_mn;
    const int n_tile = p.wmajor ? t_mn / p.m_tiles : t_mn % p.n_tiles;
    const int m_tile = p.wmajor ? t_mn % p.m_tiles : t_mn / p.n_tiles;
    const int Cin = d.Cin;

    // ---- tile geometry
    int img = 0, ty0 = 0, tx0 = 0;
    const int TW = p.TW, TWl = p.TW_log2, HWd = TW + 2;
    if (TAPS == 9) {
        img = m_tile / p.tiles_per_img;
        const int t = m_tile - img * p.tiles_per_img;
        const int ty = t / p.tiles_x, tx = t - ty * p.tiles_x;
        ty0 = ty * (BM >> TWl);
        tx0 = tx << TWl;
    }
    const int NP = TAPS == 9 ? ((BM >> TWl) + 2) * HWd : BM;

    // ---- LDS-DMA source mapping.  One instruction moves 8 rows x 128 B; lane -> (row = 8*g + lane/8, piece lane%8),
    // and the piece it FETCHES is piece ^ swizzle(row) so that the linear image holds the swizzled layout.
    const int lrow = lane >> 3, lpiece = lane & 7;
    int hoff[G::HG_PER_WAVE];           // source pixel index (element offset / Cin) or -1 -> out-of-range offset (zero)
    // logical piece (in halfs) this lane fetches: row = (wave + 8*gi)*8 + lrow, so (row >> 1) & 7 does not depend on gi
    const int lp8 = (lpiece ^ (((wave & 1) << 2) | (lrow >> 1))) * 8;
#pragma unroll
    for (int gi = 0; gi < G::HG_PER_WAVE; ++gi) {
        const int row = (wave + 8 * gi) * 8 + lrow;
        int off = -1;
        if (TAPS == 9) {
            if (row < NP) {
                const int hy = row / HWd, hx = row - hy * HWd;
                const int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;
                if ((unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W) {
                    const int sy = d.ups ? (iy >> 1) : iy, sx = d.ups ? (ix >> 1) : ix;
                    off = (img * p.Hs + sy) * p.Ws + sx;
                }
            }
        } else {
            const int pix = m_tile * BM + row;
            if (row < BM && pix < p.M) off = pix;
        }
        hoff[gi] = off;
    }
    // weight rows: 32 pieces of 8 rows, wave w takes pieces w, w+8, w+16, w+24
    const unsigned wrow0 = (unsigned)(n_tile * BN + wave * 8 + lrow);
    const unsigned src_pix = (unsigned)d.B * p.Hs * p.Ws, out_pix = (unsigned)p.M;
    const int C0 = d.src1 ? d.C0 : Cin, C1 = Cin - C0;           // operand = channel concat of two NHWC tensors
    const __amdgpu_buffer_rsrc_t r_src = make_rsrc(d.src, src_pix * C0 * 2u);
    const __amdgpu_buffer_rsrc_t r_src1 = make_rsrc(d.src1 ? d.src1 : d.src, src_pix * C1 * 2u);
    const __amdgpu_buffer_rsrc_t r_w = make_rsrc(d.weight, (unsigned)p.n_tiles * BN * TAPS * Cin * 2u);

    // halo rows of channels [coff, coff+64) of the NHWC tensor behind `rsrc` (`cstride` channels per pixel)
    auto issue_halo = [&](__amdgpu_buffer_rsrc_t rsrc, int cstride, int coff, int hb) {
        char* dst = Hb + hb * G::HBYTES + wave * 1024;
#pragma unroll
        for (int gi = 0; gi < G::HG_PER_WAVE; ++gi) {
            if (wave + 8 * gi < G::HGROUPS) {
                const unsigned vo = hoff[gi] >= 0 ? ((unsigned)hoff[gi] * (unsigned)cstride + lp8) * 2u : C16_OOB;
                bload16(rsrc, vo, (unsigned)coff * 2u, dst + gi * 8192);
            }
        }
    };
    // rows n = wrow0 + 64*j of the [Cout][rowlen] fp16 matrix behind `rsrc`, 64 channels at element offset `delta`
    auto issue_w = [&](__amdgpu_buffer_rsrc_t rsrc, unsigned rowlen, unsigned delta, int wb) {
        char* dst = Wb + wb * G::WBYTES + wave * 1024;
        const unsigned vo = (wrow0 * rowlen + lp8) * 2u;
        if (WNW == 4) {
#pragma unroll
            for (int j = 0; j < 4; ++j) bload16(rsrc, vo, (delta + 64u * j * rowlen) * 2u, dst + j * 8192);
        } else if (wave < BN / 8) {
            bload16(rsrc, vo, delta * 2u, dst);
        }
    };

    // halo of main-operand chunk `c` (64 channels of the concat)
    auto issue_main_halo = [&](int c, int hb) {
        const int cb = c * C16_KC;
        if (cb < C0) issue_halo(r_src, C0, cb, hb);
        else issue_halo(r_src1, C1, cb - C0, hb);
    };
    // ---- fused GroupNorm (+FiLM) affine + swish of the operand, applied IN LDS: each lane transforms exactly the
    // 16-byte pieces it fetched (so it needs no barrier, only its own vmcnt wait, knows which rows are zero padding --
    // the reference pads the ACTIVATED tensor -- and which 8 channels it holds: lp8 .. lp8+7 of the chunk).
    const bool fuse_gn = TAPS == 9 && d.gn_scale != nullptr;
    // The chunk's 64 scales and 64 shifts travel by LDS-DMA as well (wave 0, 16 lanes each) and are read back into
    // registers one barrier later: a VGPR load inside the rolled tap loop makes the compiler wait with vmcnt(0) in
    // front of every use, which would drain the request pipeline in each activated tap.
    char* const Gb = lds + G::LDS_TILES;
    f32x4 gsc0, gsc1, gsh0, gsh1;
    auto issue_gn = [&](int c) {
        if (wave == 0 && lane < 16) {
            const unsigned nb = (unsigned)d.B * Cin * 4u, vo = ((unsigned)(img * Cin + c * C16_KC) * 4u) + lane * 16u;
            bload16(make_rsrc(d.gn_scale, nb), vo, 0u, Gb);
            bload16(make_rsrc(d.gn_shift, nb), vo, 0u, Gb + 256);
        }
    };
    auto read_gn = [&]() {
        gsc0 = *reinterpret_cast<const f32x4*>(Gb + lp8 * 4); gsc1 = *reinterpret_cast<const f32x4*>(Gb + lp8 * 4 + 16);
        gsh0 = *reinterpret_cast<const f32x4*>(Gb + 256 + lp8 * 4); gsh1 = *reinterpret_cast<const f32x4*>(Gb + 256 + lp8 * 4 + 16);
    };
    auto act_group = [&](int gi, int hb) {
        if (wave + 8 * gi < G::HGROUPS && hoff[gi] >= 0) {
            char* pl = Hb + hb * G::HBYTES + (wave + 8 * gi) * 1024 + lane * 16;
            const half8 v = *reinterpret_cast<const half8*>(pl);
            f32x4 a = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
            f32x4 b = {(float)v[4], (float)v[5], (float)v[6], (float)v[7]};
            a = gn_act(a, gsc0, gsh0, d.gn_silu);
            b = gn_act(b, gsc1, gsh1, d.gn_silu);
            const half8 o = {(_Float16)a.x, (_Float16)a.y, (_Float16)a.z, (_Float16)a.w,
                             (_Float16)b.x, (_Float16)b.y, (_Float16)b.z, (_Float16)b.w};
            *reinterpret_cast<half8*>(pl) = o;
        }
    };

    // ---- fragment read addresses (bytes).  A operand = weights (rows = output channels), B operand = pixels.
    int wa[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = wn * (NT * 32) + j * 32 + (lane & 31);
        wa[j] = n * C16_ROWB + ((kh ^ ((n >> 1) & 7)) << 4);
    }
    int q0[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = (wm * MT + i) * 32 + (lane & 31);
        q0[i] = TAPS == 9 ? (m >> TWl) * HWd + (m & (TW - 1)) : m;
    }

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto mfma_step = [&](int toff, int hb, int wb, auto&& after_first_kstep) {
        // byte offsets into `lds`; the buffer bases are multiples of 128, so the k-step XOR (bits 5-6) commutes
        int pb[MT], wo[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int q = q0[i] + toff;
            pb[i] = q * C16_ROWB + ((kh ^ ((q >> 1) & 7)) << 4) + hb * G::HBYTES;
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) wo[j] = wa[j] + G::NHB * G::HBYTES + wb * G::WBYTES;
        // explicit two-deep fragment pipeline: the 6 reads of k-step ks+1 are in flight under the 8 MFMAs of ks
        half8 a[2][NT], b[2][MT];
#pragma unroll
        for (int j = 0; j < NT; ++j) a[0][j] = *reinterpret_cast<const half8*>(lds + wo[j]);
#pragma unroll
        for (int i = 0; i < MT; ++i) b[0][i] = *reinterpret_cast<const half8*>(lds + pb[i]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            if (ks + 1 < 4) {
#pragma unroll
                for (int j = 0; j < NT; ++j) a[nxt][j] = *reinterpret_cast<const half8*>(lds + (wo[j] ^ ((ks + 1) << 5)));
#pragma unroll
                for (int i = 0; i < MT; ++i) b[nxt][i] = *reinterpret_cast<const half8*>(lds + (pb[i] ^ ((ks + 1) << 5)));
            }
            // 9-tap kernels: one LDS read of the next k-step behind each of the first MFMAs (-4 % vs a read burst up
            // front); the 1-tap GEMM form measured better with the burst (its steps are dominated by the tile loads)
            constexpr bool ILV = TAPS == 9 && MT * NT >= MT + NT;
            if (!ILV) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[cur][j], b[cur][i], acc[i][j], 0, 0, 0);
            if (ILV && ks + 1 < 4) {
#pragma unroll
                for (int n = 0; n < MT + NT; ++n) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                if constexpr (MT * NT > MT + NT) __builtin_amdgcn_sched_group_barrier(0x008, MT * NT - (MT + NT), 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (ks == 0) {
                after_first_kstep();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // ---- K loop over (chunk, tap) steps of this split-K slice, then the fused 1x1 shortcut's chunks.
    // Tiles are requested LA steps ahead of the step that multiplies them, into the buffer the step's barrier has just
    // freed; the wait in front of a barrier therefore leaves exactly the younger request group in flight
    // (`s_waitcnt vmcnt(GRP)`: VMEM operations complete in order, and the weight tile -- 4 requests per wave,
    // unconditional -- is always the youngest part of a group).
    constexpr int LA = G::NWB - 1;
    constexpr int GRP = TAPS == 1 ? G::HG_PER_WAVE + 4 : 4;
    static_assert(LA == 1 || (WNW == 4 && (TAPS == 9 || G::HGROUPS % 8 == 0)), "counted waits need fixed-size request groups");
    const int nchunks = Cin / C16_KC;
    const int c_begin = (int)((long)nchunks * slice / p.ksplit), c_end = (int)((long)nchunks * (slice + 1) / p.ksplit);
    const int SC = d.SC0 + d.SC1, nsk = d.skip0 ? SC / C16_KC : 0;
    const int s_begin = (int)((long)nsk * slice / p.ksplit), s_end = (int)((long)nsk * (slice + 1) / p.ksplit);
    const int n_main = c_end - c_begin, n_skip = s_end - s_begin;
    const unsigned wrow = (unsigned)(TAPS * Cin);
    int hb = 0, wb = 0;
    bool pend = false;                 // the previous step requested a group that may still be in flight
    int ahead = 0;                     // 1-tap form: request groups younger than the one the next step multiplies
    auto wait_tiles = [&]() {
        if (TAPS == 1 && LA > 1) {
            if (ahead >= 2 && LA >= 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * GRP) : "memory");
            else if (ahead >= 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GRP) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        } else if (LA == 2 && pend) {
            if constexpr (GRP == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(GRP) : "memory");
        } else {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
    };
    if (n_main > 0) {
        issue_main_halo(c_begin, 0);
        if (fuse_gn) issue_gn(c_begin);
        issue_w(r_w, wrow, (unsigned)(c_begin * C16_KC), 0);
        if (TAPS == 1) {
            // tiles of the steps c_begin + 1 .. c_begin + LA - 1 (the first one was requested above)
#pragma unroll
            for (int a = 1; a < LA; ++a)
                if (c_begin + a < c_end) {
                    issue_main_halo(c_begin + a, a);
                    issue_w(r_w, wrow, (unsigned)((c_begin + a) * C16_KC), a);
                    ahead = a;
                }
        } else if (LA == 2) {
            issue_w(r_w, wrow, (unsigned)(Cin + c_begin * C16_KC), 1);
            pend = true;
        }
        if (fuse_gn) {
            wait_tiles();                           // the first halo (this wave's pieces) and the parameters have landed
            __builtin_amdgcn_s_barrier();
            read_gn();
#pragma unroll
            for (int gi = 0; gi < G::HG_PER_WAVE; ++gi) act_group(gi, 0);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // activated pieces are in LDS before the first step's barrier
        }
    }
#pragma unroll 1
    for (int c = c_begin; c < c_end; ++c) {
        const bool more = c + 1 < c_end;
        int toff = 0, kx = 0;
        // (rolled on purpose: unrolled, the compiler hoists 9 x 16 loop-invariant fragment addresses and spills)
#pragma unroll 1
        for (int tap = 0; tap < TAPS; ++tap) {
            wait_tiles();
            __builtin_amdgcn_s_barrier();           // this step's tiles have landed (every wave's); the oldest buffers are free
            auto issue_next = [&]() {
                pend = false;
                if (TAPS == 9) {
                    if (tap == 0 && more) {         // next chunk's halo (older than this group's weight tile: landed by tap 1)
                        issue_main_halo(c + 1, hb ^ 1);
                        if (fuse_gn) issue_gn(c + 1);
                    }
                } else if (c + LA < c_end) {
                    issue_main_halo(c + LA, hb + LA >= G::NHB ? hb + LA - G::NHB : hb + LA);
                }
                int t2 = tap + LA, c2 = c;
                while (t2 >= TAPS) { t2 -= TAPS; ++c2; }
                if (c2 < c_end) {
                    issue_w(r_w, wrow, (unsigned)(t2 * Cin + c2 * C16_KC), wb + LA >= G::NWB ? wb + LA - G::NWB : wb + LA);
                    pend = true;
                }
                // 1-tap form: the step after this one finds min(LA - 1, steps left after it) younger groups in flight
                if (TAPS == 1) { const int left = c_end - 2 - c; ahead = left < LA - 1 ? (left < 0 ? 0 : left) : LA - 1; }
            };
            // the next tiles are requested behind the first 8 MFMAs: the LDS-DMA issue (M0 set-up, 5-10 buffer
            // loads) no longer sits between the barrier and the first fragment reads
            mfma_step(toff, hb, wb, issue_next);
            // the next chunk's halo and parameters landed before this step's barrier (tap >= 1): one 8-row piece is
            // activated per tap
            if (fuse_gn && more && tap >= 1 && tap <= G::HG_PER_WAVE) {
                if (tap == 1) read_gn();
                act_group(tap - 1, hb ^ 1);
            }
            wb = wb + 1 == G::NWB ? 0 : wb + 1;
            if (++kx == 3) { kx = 0; toff += HWd - 2; } else { ++toff; }
        }
        hb = hb + 1 == G::NHB ? 0 : hb + 1;
    }
    if (TAPS == 9 && n_skip > 0) {
        // fused 1x1 shortcut (skip_connection of a ResBlock, unet.py:222,256): extra K chunks over the block's RAW
        // input, read at the centre tap of its halo.  (Started after the main loop has drained: one pipeline
        // bubble per launch, and the main loop carries no shortcut state.)
        const __amdgpu_buffer_rsrc_t r_skw = make_rsrc(d.skip_weight, (unsigned)p.n_tiles * BN * SC * 2u);
        const __amdgpu_buffer_rsrc_t r_sk0 = make_rsrc(d.skip0, out_pix * d.SC0 * 2u);
        const __amdgpu_buffer_rsrc_t r_sk1 = make_rsrc(d.skip1 ? d.skip1 : d.skip0, out_pix * d.SC1 * 2u);
        auto issue_skip = [&](int ch, int hbuf, int wbuf) {
            const int cb = ch * C16_KC;
            if (cb < d.SC0) issue_halo(r_sk0, d.SC0, cb, hbuf);
            else issue_halo(r_sk1, d.SC1, cb - d.SC0, hbuf);
            issue_w(r_skw, (unsigned)SC, (unsigned)cb, wbuf);
        };
        __syncthreads();
        issue_skip(s_begin, hb, wb);
#pragma unroll 1
        for (int ch = s_begin; ch < s_end; ++ch) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const int wbn = wb + 1 == G::NWB ? 0 : wb + 1;
            if (ch + 1 < s_end) issue_skip(ch + 1, hb ^ 1, wbn);
            mfma_step(HWd + 1, hb, wb, [] {});
            wb = wbn;
            hb ^= 1;
        }
    }
    // residual tile of the epilogue: requested NOW, before the barrier and the LDS transposition, so that its HBM
    // latency overlaps them (nothing else hides it with one workgroup per CU)
    constexpr int ITS = MT * 32 / 8;                 // lane -> (pixel = it*8 + lane/8, 8 channels = piece lane%8)
    // Split-K launches with fp16 slabs (`slab16`) write their partial tile through the SAME LDS transposition as a
    // finished tile -- 16-byte stores, 8 lanes = one 128-byte line, half the bytes of an fp32 slab -- into slab `slice`;
    // bias / residual / statistics belong to the reduction pass.  (The fp32 form stores f32x4 per lane and pixel: 32-byte
    // segments.)  A partial sum rounded to fp16 carries 2^-12 of ITS OWN magnitude, the same order as the final rounding
    // of the fp16 output tensor; with k slices the result's error grows by about sqrt(1 + k / 4) for partials of the sum's
    // magnitude, which is why the host uses fp16 slabs for k <= 8 only (full-configuration goldens: tests).
    const bool partial16 = p.ksplit > 1 && p.slab16 != 0;
    const _Float16* const res = partial16 ? nullptr : reinterpret_cast<const _Float16*>(d.res);
    _Float16* const out = partial16 ? reinterpret_cast<_Float16*>(d.workspace) + (size_t)slice * p.M * d.Cout
                                    : reinterpret_cast<_Float16*>(d.out);
    const float* const bias = partial16 ? nullptr : d.bias;
    const int cbase = n_tile * C16_BN + wn * 64;
    const bool wave_on = cbase < d.Cout;            // Cout % 64 == 0: a wave's 64-channel slice is all in or all out
    const int chn = cbase + lpiece * 8;
    int opix[ITS];
    uint4 rv[ITS];
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
        const int m = wm * MT * 32 + it * 8 + lrow;
        int pix, rpix;
        if (TAPS == 9) {
            const int oy = ty0 + (m >> TWl), ox = tx0 + (m & (TW - 1));
            pix = (img * d.H + oy) * d.W + ox;
            rpix = d.res_ups ? (img * (d.H >> 1) + (oy >> 1)) * (d.W >> 1) + (ox >> 1) : pix;
        } else {
            pix = m_tile * BM + m;
            rpix = pix;
            if (pix >= p.M) pix = -1;
        }
        if (!wave_on) pix = -1;
        opix[it] = pix;
        rv[it] = uint4{0u, 0u, 0u, 0u};
        if (p.ksplit == 1 && WNW == 4 && res && pix >= 0) rv[it] = *reinterpret_cast<const uint4*>(res + (size_t)rpix * d.Cout + chn);
    }
    __syncthreads();                   // all fragment reads done: LDS becomes the epilogue's staging area

    // ================================================================ epilogue
    // D layout (32x32 MFMA, A = weights): lane -> pixel = lane & 31 of M tile i, channels
    // wn*64 + j*32 + 8*(r>>2) + 4*(lane>>5) + (r&3).
    if constexpr (WNW == 1) {
        // small-Cout output convolution (out.2 of the UNet, unet.py:627-631): fp32 NCHW, D rows = channels
        // (r&3) + 8*(r>>2) + 4*kh, col = pixel; for a fixed channel 32 lanes store 32 consecutive x
        float* outf = reinterpret_cast<float*>(d.out);
        const int m = wm * 32 + (lane & 31);
        const int oy = ty0 + (m >> TWl), ox = tx0 + (m & (TW - 1));
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int ch = (r & 3) + 8 * (r >> 2) + 4 * kh;
            if (ch < d.Cout)
                outf[(((size_t)img * d.Cout + ch) * d.H + oy) * d.W + ox] = acc[0][0][r] + (d.bias ? d.bias[ch] : 0.f);
        }
        return;
    }
    if (p.ksplit > 1 && !partial16) {
        float* ws = d.workspace + (size_t)slice * p.M * d.Cout;
        if (!wave_on) return;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int m = (wm * MT + i) * 32 + (lane & 31);
            int pix;
            if (TAPS == 9) pix = (img * d.H + ty0 + (m >> TWl)) * d.W + tx0 + (m & (TW - 1));
            else pix = m_tile * BM + m;
            if (TAPS == 1 && pix >= p.M) continue;
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const int ch = cbase + j * 32 + 8 * rg + 4 * kh;
                    f32x4 v = {acc[i][j][4 * rg], acc[i][j][4 * rg + 1], acc[i][j][4 * rg + 2], acc[i][j][4 * rg + 3]};
                    *reinterpret_cast<f32x4*>(ws + (size_t)pix * d.Cout + ch) = v;
                }
        }
        return;
    }
    char* const stage = lds + wave * (MT * 32) * C16_EPITCH;
    float* const stat_lds = reinterpret_cast<float*>(lds + 8 * (MT * 32) * C16_EPITCH);
    if (wave_on) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            f32x4 bias4[4];
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int ch = cbase + j * 32 + 8 * rg + 4 * kh;
                bias4[rg] = bias ? *reinterpret_cast<const f32x4*>(bias + ch) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const f32x4 b4 = bias4[rg];
                    char* dst = stage + (i * 32 + (lane & 31)) * C16_EPITCH + (j * 32 + 8 * rg + 4 * kh) * 2;
                    half4 h = {(_Float16)(acc[i][j][4 * rg] + b4.x), (_Float16)(acc[i][j][4 * rg + 1] + b4.y),
                               (_Float16)(acc[i][j][4 * rg + 2] + b4.z), (_Float16)(acc[i][j][4 * rg + 3] + b4.w)};
                    *reinterpret_cast<half4*>(dst) = h;
                }
        }
    }
    // wave-local hand-off (each wave re-reads only its own staging region)
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");

    float cs[8], cq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) cs[e] = cq[e] = 0.f;
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
        half8 v = *reinterpret_cast<const half8*>(stage + (it * 8 + lrow) * C16_EPITCH + lpiece * 16);
        if (res) {
            const half8 r8 = __builtin_bit_cast(half8, rv[it]);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (_Float16)((float)v[e] + (float)r8[e]);
        }
        if (opix[it] >= 0) {
            *reinterpret_cast<half8*>(out + (size_t)opix[it] * d.Cout + chn) = v;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float f = (float)v[e];
                cs[e] += f;
                cq[e] += f * f;
            }
        }
    }
    if (d.stats_out && !partial16) {
        // GroupNorm partials of the tensor just written (of the ROUNDED values the consumer will read):
        // reduce over the 8 pixel rows of a wave (lanes with equal lane%8), then over the two pixel-halves (wm)
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            cs[e] += __shfl_xor(cs[e], 8);  cq[e] += __shfl_xor(cq[e], 8);
            cs[e] += __shfl_xor(cs[e], 16); cq[e] += __shfl_xor(cq[e], 16);
            cs[e] += __shfl_xor(cs[e], 32); cq[e] += __shfl_xor(cq[e], 32);
        }
        if (lane < 8 && wave_on) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = wn * 64 + lane * 8 + e;
                stat_lds[(wm * C16_BN + c) * 2 + 0] = cs[e];
                stat_lds[(wm * C16_BN + c) * 2 + 1] = cq[e];
            }
        }
        __syncthreads();
        if (tid < C16_BN) {
            const int n = n_tile * C16_BN + tid;
            if (n < d.Cout) {
                const float a = stat_lds[tid * 2] + stat_lds[(C16_BN + tid) * 2];
                const float q = stat_lds[tid * 2 + 1] + stat_lds[(C16_BN + tid) * 2 + 1];
                *reinterpret_cast<float2*>(d.stats_out + ((size_t)m_tile * d.Cout + n) * 2) = float2{a, q};
            }
        }
    }
}

// =====================================================================================
// 256 pixels x 128 output channels, FOUR waves per workgroup, TWO workgroups per CU (round 5).
//
// Launches with Cout = 128 (the classifier's 256^2 / 128^2 levels: 20 of its 42 forward + data-gradient 3x3 convolutions,
// 72 % of its FLOPs) ran on the 256-channel tile above with half of the waves multiplying zero weight rows.  Their K loop
// is short as well (Cin = 128: 18 taps), so with ONE workgroup per CU the tile prologue (halo fetch + GroupNorm) and the
// epilogue (store-bound) are exposed for a third of the launch.  This kernel keeps the 128 x 64 wave tile (6 fragment
// reads per 8 MFMAs) with 2 x 2 waves and fits TWO workgroups into a CU's LDS (77 KB each: ONE halo buffer, two weight
// buffers), so the prologue / chunk hand-over / epilogue of one workgroup run under the MFMAs of the other; the CU still
// holds 8 waves, 2 per SIMD.  Same LDS image, swizzle, fused GroupNorm-in-LDS and epilogue as conv16_kernel<9, 4, 4>;
// no split-K, no fused shortcut (neither occurs at Cout = 128).
// =====================================================================================
struct N128Geom {
    static constexpr int BM = 256, BN = 128, MT = 4, NT = 2, NWV = 4;
    static constexpr int HROWS = 344, HGROUPS = HROWS / 8, HG_PER_WAVE = (HGROUPS + NWV - 1) / NWV;
    static constexpr int HBYTES = HROWS * C16_ROWB, WBYTES = BN * C16_ROWB;
    static constexpr int LDS_TILES = HBYTES + 2 * WBYTES;
    static constexpr int LDS_MAIN = LDS_TILES + 512;
    static constexpr int LDS_EPI = NWV * (MT * 32) * C16_EPITCH + 2 * BN * 2 * 4;
    static constexpr int LDS_BYTES = LDS_MAIN > LDS_EPI ? LDS_MAIN : LDS_EPI;
};
static_assert(2 * N128Geom::LDS_BYTES <= 160 * 1024, "two workgroups per CU");

__global__ __launch_bounds__(256, 2) void conv16_n128_kernel(const Conv16Args p) {
    using G = N128Geom;
    constexpr int BM = G::BM, BN = G::BN, MT = G::MT, NT = G::NT;
    __shared__ __attribute__((aligned(1024))) char lds[G::LDS_BYTES];
    char* const Hb = lds;
    char* const Wb = lds + G::HBYTES;
    char* const Gb = lds + G::LDS_TILES;

    const ddnm_conv16_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int kh = lane >> 5;
    // workgroup -> (pixel tile, 128-channel tile); the channel tiles of one pixel tile are neighbours (same XCD: the
    // second one finds the halo in L2)
    const int q_wg = xcd_swizzle(blockIdx.x, gridDim.x);
    const int m_tile = q_wg / p.n_tiles, n_tile = q_wg - m_tile * p.n_tiles;
    const int Cin = d.Cin;

    const int img = m_tile / p.tiles_per_img;
    const int TW = p.TW, TWl = p.TW_log2, HWd = TW + 2;
    int ty0, tx0;
    {
        const int t = m_tile - img * p.tiles_per_img;
        const int ty = t / p.tiles_x, tx = t - ty * p.tiles_x;
        ty0 = ty * (BM >> TWl);
        tx0 = tx << TWl;
    }
    const int NP = ((BM >> TWl) + 2) * HWd;

    // ---- LDS-DMA source mapping: request group G = wave + 4*gi moves halo rows 8G .. 8G+7 (lane -> row 8G + lane/8,
    // piece lane%8); (row >> 1) & 7 = ((G & 1) << 2) | (lrow >> 1) and G & 1 == wave & 1
    const int lrow = lane >> 3, lpiece = lane & 7;
    const int lp8 = (lpiece ^ (((wave & 1) << 2) | (lrow >> 1))) * 8;
    // source pixel of halo row 8*(wave + 4*gi) + lrow, or -1 for zero padding / rows beyond the patch.  Evaluated where it
    // is used (once per 64-channel chunk), NOT kept in 11 registers across the MFMA loop: the accumulators (128), two
    // fragment sets (48) and the epilogue's residual tile leave no room (the first build spilled 34 registers); the empty
    // asm keeps the compiler from hoisting the arithmetic out of the chunk loop again.
    // (row / HWd by multiply-shift: the runtime division costs ~35 vector instructions, and this lambda runs 22 times per
    // chunk in a kernel whose K loop is 18 taps -- the first build spent 6.6 non-MFMA vector instructions per MFMA,
    // profiles/r05_pmc_stalls_n128.md; exact for row < 2^20 / (magic * HWd - 2^20), i.e. far beyond the 344 rows here)
    const unsigned hw_magic = (1u << 20) / (unsigned)HWd + 1u;
    auto halo_src = [&](int gi) -> int {
        int row = (wave + G::NWV * gi) * 8 + lrow;
        asm volatile("" : "+v"(row));
        int off = -1;
        if (row < NP) {
            const int hy = (int)(((unsigned)row * hw_magic) >> 20), hx = row - hy * HWd;
            const int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;
            if ((unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W) {
                const int sy = d.ups ? (iy >> 1) : iy, sx = d.ups ? (ix >> 1) : ix;
                off = (img * p.Hs + sy) * p.Ws + sx;
            }
        }
        return off;
    };
    // weight rows: 16 groups of 8 rows, wave w takes groups w, w+4, w+8, w+12
    const unsigned wrow0 = (unsigned)(n_tile * BN + wave * 8 + lrow);
    const unsigned src_pix = (unsigned)d.B * p.Hs * p.Ws;
    const int C0 = d.src1 ? d.C0 : Cin, C1 = Cin - C0;
    const __amdgpu_buffer_rsrc_t r_src = make_rsrc(d.src, src_pix * C0 * 2u);
    const __amdgpu_buffer_rsrc_t r_src1 = make_rsrc(d.src1 ? d.src1 : d.src, src_pix * C1 * 2u);
    const __amdgpu_buffer_rsrc_t r_w = make_rsrc(d.weight, (unsigned)p.n_tiles * BN * 9u * Cin * 2u);
    const unsigned wrow = 9u * (unsigned)Cin;

    auto issue_halo = [&](int c) {
        const int cb = c * C16_KC;
        const bool first = cb < C0;
        const __amdgpu_buffer_rsrc_t rsrc = first ? r_src : r_src1;
        const unsigned cstride = first ? C0 : C1, coff = first ? cb : cb - C0;
        char* dst = Hb + wave * 1024;
#pragma unroll
        for (int gi = 0; gi < G::HG_PER_WAVE; ++gi) {
            if (wave + G::NWV * gi < G::HGROUPS) {
                const int ho = halo_src(gi);
                const unsigned vo = ho >= 0 ? ((unsigned)ho * cstride + lp8) * 2u : C16_OOB;
                bload16(rsrc, vo, coff * 2u, dst + gi * (G::NWV * 1024));
            }
        }
    };
    auto issue_w = [&](unsigned delta, int wb) {
        char* dst = Wb + wb * G::WBYTES + wave * 1024;
        const unsigned vo = (wrow0 * wrow + lp8) * 2u;
#pragma unroll
        for (int j = 0; j < 4; ++j) bload16(r_w, vo, (delta + 32u * j * wrow) * 2u, dst + j * (G::NWV * 1024));
    };
    const bool fuse_gn = d.gn_scale != nullptr;
    f32x4 gsc0, gsc1, gsh0, gsh1;
    auto issue_gn = [&](int c) {
        if (wave == 0 && lane < 16) {
            const unsigned nb = (unsigned)d.B * Cin * 4u, vo = ((unsigned)(img * Cin + c * C16_KC) * 4u) + lane * 16u;
            bload16(make_rsrc(d.gn_scale, nb), vo, 0u, Gb);
            bload16(make_rsrc(d.gn_shift, nb), vo, 0u, Gb + 256);
        }
    };
    auto read_gn = [&]() {
        gsc0 = *reinterpret_cast<const f32x4*>(Gb + lp8 * 4); gsc1 = *reinterpret_cast<const f32x4*>(Gb + lp8 * 4 + 16);
        gsh0 = *reinterpret_cast<const f32x4*>(Gb + 256 + lp8 * 4); gsh1 = *reinterpret_cast<const f32x4*>(Gb + 256 + lp8 * 4 + 16);
    };
    // which of this lane's halo rows lie inside the image (bit gi): the same for every chunk, so one register instead of
    // 11 offsets -- the GroupNorm pass below must leave the zero padding alone
    unsigned in_image = 0u;
    if (fuse_gn) {
#pragma unroll
        for (int gi = 0; gi < G::HG_PER_WAVE; ++gi)
            if (wave + G::NWV * gi < G::HGROUPS && halo_src(gi) >= 0) in_image |= 1u << gi;
    }
    auto act_group = [&](int gi) {
        if (wave + G::NWV * gi < G::HGROUPS && ((in_image >> gi) & 1u)) {
            char* pl = Hb + (wave + G::NWV * gi) * 1024 + lane * 16;
            const half8 v = *reinterpret_cast<const half8*>(pl);
            f32x4 a = {(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
            f32x4 b = {(float)v[4], (float)v[5], (float)v[6], (float)v[7]};
            a = gn_act(a, gsc0, gsh0, d.gn_silu);
            b = gn_act(b, gsc1, gsh1, d.gn_silu);
            const half8 o = {(_Float16)a.x, (_Float16)a.y, (_Float16)a.z, (_Float16)a.w,
                             (_Float16)b.x, (_Float16)b.y, (_Float16)b.z, (_Float16)b.w};
            *reinterpret_cast<half8*>(pl) = o;
        }
    };

    // ---- fragment read addresses (bytes): A operand = weights (rows = output channels), B operand = pixels
    int wa[NT];
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        const int n = wn * (NT * 32) + j * 32 + (lane & 31);
        wa[j] = n * C16_ROWB + ((kh ^ ((n >> 1) & 7)) << 4);
    }
    int q0[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = (wm * MT + i) * 32 + (lane & 31);
        q0[i] = (m >> TWl) * HWd + (m & (TW - 1));
    }
    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    auto mfma_step = [&](int toff, int wb, auto&& after_first_kstep) {
        int pb[MT], wo[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int q = q0[i] + toff;
            pb[i] = q * C16_ROWB + ((kh ^ ((q >> 1) & 7)) << 4);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) wo[j] = wa[j] + G::HBYTES + wb * G::WBYTES;
        half8 a[2][NT], b[2][MT];
#pragma unroll
        for (int j = 0; j < NT; ++j) a[0][j] = *reinterpret_cast<const half8*>(lds + wo[j]);
#pragma unroll
        for (int i = 0; i < MT; ++i) b[0][i] = *reinterpret_cast<const half8*>(lds + pb[i]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            if (ks + 1 < 4) {
#pragma unroll
                for (int j = 0; j < NT; ++j) a[nxt][j] = *reinterpret_cast<const half8*>(lds + (wo[j] ^ ((ks + 1) << 5)));
#pragma unroll
                for (int i = 0; i < MT; ++i) b[nxt][i] = *reinterpret_cast<const half8*>(lds + (pb[i] ^ ((ks + 1) << 5)));
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[cur][j], b[cur][i], acc[i][j], 0, 0, 0);
            if (ks + 1 < 4) {
#pragma unroll
                for (int n = 0; n < MT + NT; ++n) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_group_barrier(0x008, MT * NT - (MT + NT), 0);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (ks == 0) {
                after_first_kstep();
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    // ---- K loop: chunk = 64 input channels; ONE halo buffer (re-filled between chunks, behind a barrier: the other
    // workgroup of the CU covers the gap), weight tiles double-buffered one tap ahead across chunk boundaries.
    const int nchunks = Cin / C16_KC;
    issue_halo(0);
    if (fuse_gn) issue_gn(0);
    issue_w(0u, 0);
    int wb = 0;
#pragma unroll 1
    for (int c = 0; c < nchunks; ++c) {
        if (c > 0) {
            __builtin_amdgcn_s_barrier();           // every wave has finished the previous chunk's fragment reads
            issue_halo(c);
            if (fuse_gn) issue_gn(c);
        }
        if (fuse_gn) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();           // the chunk's scale | shift (wave 0's requests) have landed
            read_gn();
#pragma unroll
            for (int gi = 0; gi < G::HG_PER_WAVE; ++gi) act_group(gi);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
        const bool more = c + 1 < nchunks;
        int toff = 0, kx = 0;
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();           // this tap's weight tile (tap 0: the halo too) is complete; wb ^ 1 is free
            auto issue_next = [&]() {
                if (tap + 1 < 9) issue_w((unsigned)((tap + 1) * Cin + c * C16_KC), wb ^ 1);
                else if (more) issue_w((unsigned)((c + 1) * C16_KC), wb ^ 1);
            };
            mfma_step(toff, wb, issue_next);
            wb ^= 1;
            if (++kx == 3) { kx = 0; toff += HWd - 2; } else { ++toff; }
        }
    }

    // ---- epilogue (as conv16_kernel, WNW = 4 form): residual tile requested before the LDS transposition
    constexpr int ITS = MT * 32 / 8;
    const _Float16* const res = reinterpret_cast<const _Float16*>(d.res);
    _Float16* const out = reinterpret_cast<_Float16*>(d.out);
    const float* const bias = d.bias;
    const int cbase = n_tile * BN + wn * 64;
    const int chn = cbase + lpiece * 8;
    // output pixel of staging row it*8 + lrow of this wave (recomputed where it is needed: sixteen 64-bit addresses kept
    // next to the accumulators and the residual tile were spilled to scratch)
    auto out_pixel = [&](int it, int& rpix) -> int {
        int m = wm * MT * 32 + it * 8 + lrow;
        asm volatile("" : "+v"(m));
        const int oy = ty0 + (m >> TWl), ox = tx0 + (m & (TW - 1));
        rpix = d.res_ups ? (img * (d.H >> 1) + (oy >> 1)) * (d.W >> 1) + (ox >> 1) : (img * d.H + oy) * d.W + ox;
        return (img * d.H + oy) * d.W + ox;
    };
    // residual tile: the first half is requested NOW (its latency overlaps the barrier and the LDS transposition), the
    // second half once the accumulators are dead -- all sixteen 16-byte pieces next to 128 accumulator registers do not
    // fit the 256-register budget of two waves per SIMD
    uint4 rv[ITS];
    auto load_res = [&](int it) {
        int rpix;
        out_pixel(it, rpix);
        rv[it] = *reinterpret_cast<const uint4*>(res + (size_t)rpix * d.Cout + chn);
    };
    if (res) {
#pragma unroll
        for (int it = 0; it < ITS / 2; ++it) load_res(it);
    }
    __syncthreads();                   // all fragment reads done: LDS becomes the epilogue's staging area
    char* const stage = lds + wave * (MT * 32) * C16_EPITCH;
    float* const stat_lds = reinterpret_cast<float*>(lds + G::NWV * (MT * 32) * C16_EPITCH);
#pragma unroll
    for (int j = 0; j < NT; ++j) {
        f32x4 bias4[4];
#pragma unroll
        for (int rg = 0; rg < 4; ++rg) {
            const int ch = cbase + j * 32 + 8 * rg + 4 * kh;
            bias4[rg] = bias ? *reinterpret_cast<const f32x4*>(bias + ch) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const f32x4 b4 = bias4[rg];
                char* dst = stage + (i * 32 + (lane & 31)) * C16_EPITCH + (j * 32 + 8 * rg + 4 * kh) * 2;
                half4 h = {(_Float16)(acc[i][j][4 * rg] + b4.x), (_Float16)(acc[i][j][4 * rg + 1] + b4.y),
                           (_Float16)(acc[i][j][4 * rg + 2] + b4.z), (_Float16)(acc[i][j][4 * rg + 3] + b4.w)};
                *reinterpret_cast<half4*>(dst) = h;
            }
    }
    if (res) {
#pragma unroll
        for (int it = ITS / 2; it < ITS; ++it) load_res(it);
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    float cs[8], cq[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) cs[e] = cq[e] = 0.f;
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
        half8 v = *reinterpret_cast<const half8*>(stage + (it * 8 + lrow) * C16_EPITCH + lpiece * 16);
        if (res) {
            const half8 r8 = __builtin_bit_cast(half8, rv[it]);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = (_Float16)((float)v[e] + (float)r8[e]);
        }
        int rpix_unused;
        const int pix = out_pixel(it, rpix_unused);
        *reinterpret_cast<half8*>(out + (size_t)pix * d.Cout + chn) = v;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float f = (float)v[e];
            cs[e] += f;
            cq[e] += f * f;
        }
    }
    if (d.stats_out) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            cs[e] += __shfl_xor(cs[e], 8);  cq[e] += __shfl_xor(cq[e], 8);
            cs[e] += __shfl_xor(cs[e], 16); cq[e] += __shfl_xor(cq[e], 16);
            cs[e] += __shfl_xor(cs[e], 32); cq[e] += __shfl_xor(cq[e], 32);
        }
        if (lane < 8) {
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const int c = wn * 64 + lane * 8 + e;
                stat_lds[(wm * BN + c) * 2 + 0] = cs[e];
                stat_lds[(wm * BN + c) * 2 + 1] = cq[e];
            }
        }
        __syncthreads();
        if (tid < BN) {
            const float a = stat_lds[tid * 2] + stat_lds[(BN + tid) * 2];
            const float q = stat_lds[tid * 2 + 1] + stat_lds[(BN + tid) * 2 + 1];
            *reinterpret_cast<float2*>(d.stats_out + ((size_t)m_tile * d.Cout + n_tile * BN + tid) * 2) = float2{a, q};
        }
    }
}

// =====================================================================================
// split-K reduction: out(fp16) = round( sum_s ws[s] + bias + res ), + GroupNorm partials of the rounded values.
// grid (B * tpi, ceil(Cout / 1024)); one workgroup per (image, pixel tile, 1024-channel slab); a thread owns one
// float4 channel column and walks the tile's pixels (fixed order, no atomics).
// =====================================================================================
// one float4 channel column of slab `k` at element offset `o` (fp32 slabs, or the fp16 slabs of `slab16` launches)
template <bool SLAB16>
__device__ __forceinline__ f32x4 slab_load4(const float* __restrict__ ws, size_t o) {
    if constexpr (SLAB16) {
        const half4 h = *reinterpret_cast<const half4*>(reinterpret_cast<const _Float16*>(ws) + o);
        return f32x4{(float)h.x, (float)h.y, (float)h.z, (float)h.w};
    } else {
        return *reinterpret_cast<const f32x4*>(ws + o);
    }
}

template <bool SLAB16>
static __global__ __launch_bounds__(256) void conv16_splitk_reduce_kernel(const Conv16Args p, int tpi) {
    __shared__ f32x4 red[2][256];
    const ddnm_conv16_desc& d = p.d;
    const int cb = blockIdx.y * 256;                  // 256 channels per workgroup: 64 float4 columns x 4 pixel rows
    const int cw = min(256, d.Cout - cb);
    const int c4n = cw >> 2;
    const int rows = 256 / c4n, active = rows * c4n;
    const int hw = d.H * d.W, P = hw / tpi;
    const int b = blockIdx.x / tpi, t = blockIdx.x - b * tpi;
    const size_t slab = (size_t)p.M * d.Cout;
    const int tid = threadIdx.x;
    const _Float16* res = reinterpret_cast<const _Float16*>(d.res);
    _Float16* out = reinterpret_cast<_Float16*>(d.out);
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, ss = {0.f, 0.f, 0.f, 0.f};
    if (tid < active) {
        const int c4 = tid % c4n, prow = tid / c4n, n = cb + c4 * 4;
        f32x4 add = {0.f, 0.f, 0.f, 0.f};
        if (d.bias) add = *reinterpret_cast<const f32x4*>(d.bias + n);
        for (int pp = prow; pp < P; pp += rows) {
            const int p2 = t * P + pp;
            const size_t o = ((size_t)b * hw + p2) * d.Cout + n;
            // all slices' loads are independent: issue them back to back (the slab sum is the latency chain here)
            f32x4 v = slab_load4<SLAB16>(d.workspace, o);
#pragma unroll 4
            for (int k = 1; k < p.ksplit; ++k) v = v + slab_load4<SLAB16>(d.workspace, o + k * slab);
            v = v + add;
            if (res) {
                size_t ro = o;
                if (d.res_ups) {
                    const int oy = p2 / d.W, ox = p2 - oy * d.W;
                    ro = (((size_t)b * (d.H >> 1) + (oy >> 1)) * (d.W >> 1) + (ox >> 1)) * d.Cout + n;
                }
                const half4 r4 = *reinterpret_cast<const half4*>(res + ro);
                v = v + f32x4{(float)r4.x, (float)r4.y, (float)r4.z, (float)r4.w};
            }
            const half4 h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
            *reinterpret_cast<half4*>(out + o) = h;
            const f32x4 f = {(float)h.x, (float)h.y, (float)h.z, (float)h.w};
            s += f;
            ss += f * f;
        }
    }
    if (!d.stats_out) return;
    red[0][tid] = s;
    red[1][tid] = ss;
    __syncthreads();
    if (tid < c4n) {
        for (int r = 1; r < rows; ++r) { s += red[0][r * c4n + tid]; ss += red[1][r * c4n + tid]; }
        float* o = d.stats_out + ((size_t)blockIdx.x * d.Cout + cb + tid * 4) * 2;
        reinterpret_cast<f32x4*>(o)[0] = f32x4{s.x, ss.x, s.y, ss.y};
        reinterpret_cast<f32x4*>(o)[1] = f32x4{s.z, ss.z, s.w, ss.w};
    }
}

// =====================================================================================
// split-K reduction that also FINALIZES the consumer's GroupNorm (ddnm_conv16_desc::fin_*): one workgroup per
// (image, slab of CS channels = whole groups) walks ALL pixels of the image, so it holds the complete per-channel
// sums of the rounded output and can turn them into the affine the next convolution applies -- no
// gn_finalize_tiles launch between a split-K convolution and its consumer (low-resolution levels: <= 1024 pixels).
// Thread = one float4 channel column x one pixel row of 256 / (CS/4); fp32 partials per thread, fp64 combination in a
// fixed order (no atomics: repeated launches are bit-identical).
// =====================================================================================
constexpr int C16_FIN_THREADS = 1024;
template <bool SLAB16>
static __global__ __launch_bounds__(C16_FIN_THREADS) void conv16_splitk_reduce_fin_kernel(const Conv16Args p, int CS) {
    __shared__ f32x4 red[2][C16_FIN_THREADS];
    __shared__ double chan[2][64];
    __shared__ float mr[2][16];
    const ddnm_conv16_desc& d = p.d;
    const int b = blockIdx.x, cb = blockIdx.y * CS;
    const int c4n = CS >> 2, rows = C16_FIN_THREADS / c4n;
    const int hw = d.H * d.W;
    const size_t slab = (size_t)p.M * d.Cout;
    const int tid = threadIdx.x;
    const int c4 = tid % c4n, prow = tid / c4n, n = cb + c4 * 4;
    const float* __restrict__ wsp = d.workspace;
    const _Float16* __restrict__ res = reinterpret_cast<const _Float16*>(d.res);
    _Float16* __restrict__ out = reinterpret_cast<_Float16*>(d.out);
    f32x4 s = {0.f, 0.f, 0.f, 0.f}, ss = {0.f, 0.f, 0.f, 0.f};
    f32x4 add = {0.f, 0.f, 0.f, 0.f};
    if (d.bias) add = *reinterpret_cast<const f32x4*>(d.bias + n);
    // the affine's own operands do not depend on the sums: request them first (threads 0 .. CS-1, one channel each)
    float gam = 0.f, bet = 0.f, f_s = 0.f, f_t = 0.f;
    if (tid < CS) {
        gam = d.fin_gamma[cb + tid];
        bet = d.fin_beta[cb + tid];
        if (d.fin_film) {
            f_s = d.fin_film[(size_t)b * d.fin_film_stride + cb + tid];
            f_t = d.fin_film[(size_t)b * d.fin_film_stride + d.Cout + cb + tid];
        }
    }
    // The pass is a latency chain unless many loads fly at once: 4 pixels per trip, every slab / residual load of the
    // trip issued before the first use (the loop as written per pixel ran at one HBM round trip per iteration).
    constexpr int PB = 4;
    for (int p0 = prow; p0 < hw; p0 += rows * PB) {
        f32x4 v[PB];
        half4 r4[PB];
        size_t o[PB];
        bool ok[PB];
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            const int pp = p0 + u * rows;
            ok[u] = pp < hw;
            o[u] = ((size_t)b * hw + (ok[u] ? pp : 0)) * d.Cout + n;
            v[u] = slab_load4<SLAB16>(wsp, o[u]);
        }
        for (int k = 1; k < p.ksplit; ++k) {
#pragma unroll
            for (int u = 0; u < PB; ++u) v[u] = v[u] + slab_load4<SLAB16>(wsp, o[u] + k * slab);
        }
        if (res) {
#pragma unroll
            for (int u = 0; u < PB; ++u) {
                size_t ro = o[u];
                if (d.res_ups) {
                    const int pp = ok[u] ? p0 + u * rows : 0;
                    const int oy = pp / d.W, ox = pp - oy * d.W;
                    ro = (((size_t)b * (d.H >> 1) + (oy >> 1)) * (d.W >> 1) + (ox >> 1)) * d.Cout + n;
                }
                r4[u] = *reinterpret_cast<const half4*>(res + ro);
            }
        }
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            if (!ok[u]) continue;
            f32x4 t = v[u] + add;
            if (res) t = t + f32x4{(float)r4[u].x, (float)r4[u].y, (float)r4[u].z, (float)r4[u].w};
            const half4 h = {(_Float16)t.x, (_Float16)t.y, (_Float16)t.z, (_Float16)t.w};
            *reinterpret_cast<half4*>(out + o[u]) = h;
            const f32x4 f = {(float)h.x, (float)h.y, (float)h.z, (float)h.w};
            s += f;
            ss += f * f;
        }
    }
    red[0][tid] = s;
    red[1][tid] = ss;
    __syncthreads();
    // per-channel totals over the pixel rows, fixed order: 8 threads per channel take every 8th row (fp64), then the
    // 8 partial sums are added in order (a single thread per channel walked up to 256 LDS rows: 4 us of the kernel)
    __shared__ double part[2][64][8];
    if (tid < CS * 8) {
        const int ch = tid >> 3, seg = tid & 7;
        const int col = ch >> 2, e = ch & 3;
        double a = 0.0, q = 0.0;
        for (int r = seg; r < rows; r += 8) {
            a += (double)red[0][r * c4n + col][e];
            q += (double)red[1][r * c4n + col][e];
        }
        part[0][ch][seg] = a;
        part[1][ch][seg] = q;
    }
    __syncthreads();
    if (tid < CS) {
        double a = 0.0, q = 0.0;
#pragma unroll
        for (int k = 0; k < 8; ++k) { a += part[0][tid][k]; q += part[1][tid][k]; }
        chan[0][tid] = a;
        chan[1][tid] = q;
        if (d.stats_out)                             // one tile per image: other consumers (skip concats) finalize from it
            *reinterpret_cast<float2*>(d.stats_out + ((size_t)b * d.Cout + cb + tid) * 2) = float2{(float)a, (float)q};
    }
    __syncthreads();
    const int cpg = d.Cout / d.fin_groups, ng = CS / cpg;
    if (tid < ng) {
        double a = 0.0, q = 0.0;
        for (int j = 0; j < cpg; ++j) { a += chan[0][tid * cpg + j]; q += chan[1][tid * cpg + j]; }     // cpg <= 64
        const double cnt = (double)hw * (double)cpg;
        const double mean = a / cnt;
        double var = q / cnt - mean * mean;
        var = var > 0.0 ? var : 0.0;
        mr[0][tid] = (float)mean;
        mr[1][tid] = (float)(1.0 / sqrt(var + (double)d.fin_eps));
    }
    __syncthreads();
    if (tid < CS) {
        const int g = tid / cpg;
        float sc = mr[1][g] * gam;
        float sh = bet - mr[0][g] * sc;
        if (d.fin_film) {                            // FiLM: GN(x)*(1+s)+t  (guided_diffusion/unet.py:248-251)
            const float s1 = 1.0f + f_s;
            sc = sc * s1;
            sh = sh * s1 + f_t;
        }
        d.fin_scale[(size_t)b * d.Cout + cb + tid] = sc;
        d.fin_shift[(size_t)b * d.Cout + cb + tid] = sh;
    }
}

// ---------------------------------------------------------------------------------------------
struct Plan16 {
    int taps, MT, TW, TW_log2, tiles_x, tiles_per_img, m_tiles, n_tiles, ksplit, stats_tiles, small;
    int fin_cs;          // > 0: the split-K reduction finalizes the consumer's GroupNorm, slabs of fin_cs channels
    int n128;            // 1: conv16_n128_kernel (256 pixels x 128 channels, 4 waves, two workgroups per CU)
};

// channel slab of the finalizing reduction (0: this launch cannot / need not finalize): whole groups, 16 .. 64 channels,
// images of at most 1024 pixels (one workgroup walks all of them)
static int fin_slab16(const ddnm_conv16_desc* d, int ksplit) {
    if (ksplit <= 1 || !d->fin_gamma || !d->fin_beta || !d->fin_scale || !d->fin_shift) return 0;
    if (d->fin_groups <= 0 || d->Cout % d->fin_groups || d->out_nchw_f32) return 0;
    const int cpg = d->Cout / d->fin_groups;
    if (cpg % 4 || cpg > 64 || d->H * d->W > 1024) return 0;
    int cs = cpg;
    while (cs < 16) cs *= 2;
    if (d->Cout % cs || cs / cpg > 16) return 0;
    return cs;
}

// pixel tiles per image of the split-K reduction: enough (image, tile, 256-channel slab) workgroups to cover the chip
// twice, at least 4 pixels per tile
static int splitk_tiles16(const ddnm_conv16_desc* d) {
    const int hw = d->H * d->W;
    const int cy = (d->Cout + 255) / 256;
    int tpi = 512 / (d->B * cy);
    if (tpi > hw / 4) tpi = hw / 4;
    if (tpi > 64) tpi = 64;
    if (tpi < 1) tpi = 1;
    while (tpi > 1 && hw % tpi) --tpi;
    return tpi;
}

static bool plan16(const ddnm_conv16_desc* d, Plan16* pl) {
    if (d->Cin <= 0 || d->Cin % C16_KC || d->B <= 0 || d->H <= 0 || d->W <= 0 || d->Cout <= 0) return false;
    if (d->ksize != 1 && d->ksize != 3) return false;
    pl->small = 0;
    pl->fin_cs = 0;
    pl->n128 = 0;
    if (d->out_nchw_f32) {
        // fp32 NCHW output with <= 32 channels: 256-pixel x 32-channel tiles, no residual / shortcut / statistics
        if (d->ksize != 3 || d->Cout > 32 || d->ups || d->res || d->skip0 || d->stats_out) return false;
        if (d->W % 32 || d->H % 8) return false;
        pl->small = 1;
        pl->taps = 9; pl->MT = 1; pl->TW = 32; pl->TW_log2 = 5; pl->tiles_x = d->W / 32;
        pl->tiles_per_img = d->H * d->W / 256;
        pl->m_tiles = d->B * pl->tiles_per_img;
        pl->n_tiles = 1; pl->ksplit = 1; pl->stats_tiles = 0;
        return true;
    }
    if (d->Cout % 64) return false;
    const int hw = d->H * d->W;
    const long M = (long)d->B * hw;
    pl->taps = d->ksize == 3 ? 9 : 1;
    pl->n_tiles = (d->Cout + C16_BN - 1) / C16_BN;
    long tiles_of[5] = {0, 0, 0, 0, 0};
    for (int mt = 1; mt <= 4; mt += (mt == 1 ? 1 : 2)) {
        if (mt == 1 && pl->taps != 1) continue;          // 64-pixel tiles: flat GEMM form only
        const int bm = 64 * mt;
        if (pl->taps == 9) {
            int tw = 32;
            while (tw >= 16 && (d->W % tw || d->H % (bm / tw))) tw >>= 1;
            if (tw < 16) continue;
            tiles_of[mt] = (M / bm) * pl->n_tiles;
        } else {
            if (d->ups || d->skip0) return false;
            tiles_of[mt] = ((M + bm - 1) / bm) * pl->n_tiles;
        }
    }
    // the 256-pixel tile unless it cannot give most of the 256 CUs a workgroup and the 128-pixel tile can
    int best_mt = tiles_of[4] > 0 ? 4 : (tiles_of[2] > 0 ? 2 : 0);
    if (tiles_of[4] < 200 && tiles_of[2] > tiles_of[4]) best_mt = 2;
    // 1x1 convolutions / im2col'ed GEMMs whose 128-pixel tiling leaves half of the CUs without a workgroup (qkv and
    // proj_out of the 8^2 .. 32^2 attention blocks: 8 .. 96 tiles, each a chain of <= 16 steps): 64-pixel tiles double the
    // workgroups of the one round and halve its length; split-K launches get half as many fp32 slabs
    if (pl->taps == 1 && best_mt == 2 && tiles_of[2] <= 128 && tiles_of[1] > tiles_of[2]) best_mt = 1;
    if (best_mt == 0) return false;
    pl->MT = best_mt;
    const int bm = 64 * best_mt;
    if (pl->taps == 9) {
        int tw = 32;
        while (tw >= 16 && (d->W % tw || d->H % (bm / tw))) tw >>= 1;
        pl->TW = tw;
        pl->TW_log2 = tw == 32 ? 5 : 4;
        pl->tiles_x = d->W / tw;
        pl->tiles_per_img = hw / bm;
        pl->m_tiles = (int)(M / bm);
    } else {
        pl->TW = 32; pl->TW_log2 = 5; pl->tiles_x = 0; pl->tiles_per_img = 0;
        pl->m_tiles = (int)((M + bm - 1) / bm);
    }
    const long tiles = (long)pl->m_tiles * pl->n_tiles;
    const int nchunks = d->Cin / C16_KC;
    int ks = 1;
    if (tiles < 160) {
        // ONE round of workgroups: tiles x slices <= 256 CUs (a 320-workgroup plan ran two rounds: 2x the time)
        ks = (int)(256 / tiles);
        if (ks > nchunks / 2) ks = nchunks / 2 > 0 ? nchunks / 2 : 1;      // keep >= 2 chunks per slice
        if (ks > 32) ks = 32;
        // a 1x1 convolution with K <= 1024 is at most 16 steps: slicing it costs more (fp32 slabs + the reduction
        // launch) than the idle CUs do (measured 1024 -> 1024 @16^2: 25.7 -> 17.7 us, 512 -> 512 @32^2: 32.8 -> 12.5 us)
        if (pl->taps == 1 && nchunks <= 16 && hw % bm == 0) ks = 1;
        if (ks < 1) ks = 1;
    }
    pl->ksplit = ks;
    // Cout = 128 on 256-pixel tiles, enough of them for two workgroups per CU to matter, no fused shortcut
    // (for Cout >= 256 the same kernel with Cout / 128 channel tiles measured SLOWER than the 256-channel tile: every pixel
    // tile's halo is fetched and activated once per channel tile; tools/experiments/HISTORY.md)
    pl->n128 = (pl->taps == 9 && best_mt == 4 && ks == 1 && !d->skip0 && pl->m_tiles >= 128 && d->Cout == 128) ? 1 : 0;
    pl->fin_cs = fin_slab16(d, ks);
    if (pl->fin_cs > 0) pl->stats_tiles = 1;
    else if (ks > 1) pl->stats_tiles = splitk_tiles16(d);
    else pl->stats_tiles = (pl->taps == 9 || hw % bm == 0) ? hw / bm : 0;
    return true;
}

extern "C" int ddnm_conv16_supported(const ddnm_conv16_desc* d) {
    Plan16 pl;
    return d && plan16(d, &pl) ? 1 : 0;
}

extern "C" int ddnm_conv16_fuses_fin(const ddnm_conv16_desc* d) {
    Plan16 pl;
    return d && plan16(d, &pl) && pl.fin_cs > 0 ? 1 : 0;
}

extern "C" int64_t ddnm_conv16_workspace_floats(const ddnm_conv16_desc* d) {
    Plan16 pl;
    if (!d || !plan16(d, &pl)) return DDNM_E_SHAPE;
    return pl.ksplit > 1 ? (int64_t)pl.ksplit * d->B * d->H * d->W * d->Cout : 0;
}

extern "C" int ddnm_conv16_stats_tiles(const ddnm_conv16_desc* d) {
    Plan16 pl;
    if (!d || !plan16(d, &pl)) return DDNM_E_SHAPE;
    return pl.stats_tiles;
}

extern "C" int ddnm_conv16(const ddnm_conv16_desc* d, void* stream) {
    if (!d || !d->src || !d->weight || !d->out) return DDNM_E_BADARG;
    Plan16 pl;
    if (!plan16(d, &pl)) return DDNM_E_SHAPE;
    if (d->ups && ((d->H | d->W) & 1)) return DDNM_E_SHAPE;
    if (d->res_ups && ((d->H | d->W) & 1)) return DDNM_E_SHAPE;
    if (d->skip0) {
        if (pl.taps != 9 || d->ups || !d->skip_weight || d->SC0 <= 0 || d->SC0 % C16_KC || d->SC1 % C16_KC ||
            (d->SC1 > 0 && !d->skip1))
            return DDNM_E_SHAPE;
    }
    {   // The kernel addresses every tensor through 32-bit byte offsets of a raw buffer descriptor and fetches zero
        // padding through the out-of-range offset 0x80000000: every operand must stay below 2 GiB (split the batch on
        // the host beyond that) -- otherwise the padding offset would land inside the buffer and read real data.
        const int64_t lim = (int64_t)1 << 31;
        const int64_t src_pix = (int64_t)d->B * (d->ups ? d->H / 2 : d->H) * (d->ups ? d->W / 2 : d->W);
        const int64_t out_pix = (int64_t)d->B * d->H * d->W;
        const int64_t c0 = d->src1 ? d->C0 : d->Cin, c1 = d->Cin - c0;
        const int64_t cmax = c0 > c1 ? c0 : c1;
        const int64_t scmax = d->SC0 > d->SC1 ? d->SC0 : d->SC1;
        if (src_pix * cmax * 2 >= lim || out_pix * d->Cout * (d->out_nchw_f32 ? 4 : 2) >= lim ||
            out_pix * scmax * 2 >= lim ||
            (int64_t)pl.n_tiles * (pl.small ? 32 : C16_BN) * pl.taps * d->Cin * 2 >= lim ||
            (int64_t)pl.n_tiles * C16_BN * (d->SC0 + d->SC1) * 2 >= lim)
            return DDNM_E_SHAPE;
    }
    if (d->stats_out && pl.stats_tiles <= 0) return DDNM_E_SHAPE;
    if ((d->gn_scale == nullptr) != (d->gn_shift == nullptr)) return DDNM_E_BADARG;
    if (d->gn_scale && pl.taps != 9) return DDNM_E_SHAPE;          // 1x1 launches take an already activated operand
    if (d->src1 && (d->C0 <= 0 || d->C0 >= d->Cin || d->C0 % C16_KC)) return DDNM_E_SHAPE;
    if (pl.ksplit > 1) {
        const int64_t need = (int64_t)pl.ksplit * d->B * d->H * d->W * d->Cout;
        if (!d->workspace || d->workspace_floats < need) return DDNM_E_BADARG;
        if (d->Cout % 4) return DDNM_E_SHAPE;
    }
    Conv16Args p;
    p.d = *d;
    p.TW = pl.TW; p.TW_log2 = pl.TW_log2; p.tiles_x = pl.tiles_x; p.tiles_per_img = pl.tiles_per_img;
    p.m_tiles = pl.m_tiles; p.n_tiles = pl.n_tiles; p.ksplit = pl.ksplit;
    p.M = d->B * d->H * d->W;
    p.Hs = d->ups ? d->H / 2 : d->H;
    p.Ws = d->ups ? d->W / 2 : d->W;
    // weight bytes > activation bytes, and few enough pixel tiles per weight stream that one XCD's CUs do not all pull
    // the same weight lines at the same moment (32 sharers measured SLOWER than replicating the weights over XCDs)
    p.wmajor = ((long)d->Cout * pl.taps > (long)p.M && pl.m_tiles <= 8) ? 1 : 0;
    // fp16 slabs for up to 8 slices only: every partial sum is rounded to fp16 once, so the added error grows like
    // sqrt(ksplit) * 2^-12 of a partial's magnitude; deeper splits keep fp32 slabs
    p.slab16 = pl.ksplit <= 8 ? 1 : 0;        // fp16 split-K slabs up to 8 slices, fp32 beyond (the workspace is sized for fp32 either way)
    hipStream_t s = (hipStream_t)stream;
    const dim3 grid(pl.m_tiles * pl.n_tiles * pl.ksplit);
    if (pl.small) {
        DDNM_LAUNCH((conv16_kernel<9, 1, 1>), grid, dim3(512), 0, s, p);
    } else if (pl.n128) {
        p.n_tiles = d->Cout / 128;
        DDNM_LAUNCH(conv16_n128_kernel, dim3(pl.m_tiles * p.n_tiles), dim3(256), 0, s, p);
    } else if (pl.taps == 9) {
        if (pl.MT == 4) { DDNM_LAUNCH((conv16_kernel<9, 4, 4>), grid, dim3(512), 0, s, p); }
        else { DDNM_LAUNCH((conv16_kernel<9, 2, 4>), grid, dim3(512), 0, s, p); }
    } else {
        if (pl.MT == 4) { DDNM_LAUNCH((conv16_kernel<1, 4, 4>), grid, dim3(512), 0, s, p); }
        else if (pl.MT == 2) { DDNM_LAUNCH((conv16_kernel<1, 2, 4>), grid, dim3(512), 0, s, p); }
        else { DDNM_LAUNCH((conv16_kernel<1, 1, 4>), grid, dim3(512), 0, s, p); }
    }
    if (pl.fin_cs > 0) {
        const dim3 g(d->B, d->Cout / pl.fin_cs);
        if (p.slab16) { DDNM_LAUNCH(conv16_splitk_reduce_fin_kernel<true>, g, dim3(C16_FIN_THREADS), 0, s, p, pl.fin_cs); }
        else { DDNM_LAUNCH(conv16_splitk_reduce_fin_kernel<false>, g, dim3(C16_FIN_THREADS), 0, s, p, pl.fin_cs); }
    } else if (pl.ksplit > 1) {
        const int tpi = pl.stats_tiles;
        const dim3 g(d->B * tpi, (d->Cout + 255) / 256);
        if (p.slab16) { DDNM_LAUNCH(conv16_splitk_reduce_kernel<true>, g, dim3(256), 0, s, p, tpi); }
        else { DDNM_LAUNCH(conv16_splitk_reduce_kernel<false>, g, dim3(256), 0, s, p, tpi); }
    }
    return 0;
}
