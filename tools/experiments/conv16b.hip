// Two-workgroups-per-CU form of the fp16-activation 3x3 convolution (ddnm_conv16, csrc/conv16.hip), gfx950.
//
// Why it exists.  conv16_kernel<9,4,4> owns a whole CU (150 KB of LDS, 8 waves): its main loop runs at the rate of the
// best plain-HIP GEMM structure on this part, but its epilogue -- residual tile in, fp16 tile out, 256 KB per workgroup
// -- is an HBM burst that all 256 CUs issue at the same moment (equal tiles, lock-step rounds), and nothing on the CU
// computes meanwhile: 17 % of every 256^2 launch.  Here a workgroup is HALF a CU:
//
//   * 4 waves (2 x 2), wave tile 128 pixels x 64 channels = 4 x 2 MFMA tiles of v_mfma_f32_32x32x16_f16 -- the same
//     per-wave work and the same 6 fragment reads per 8 MFMAs as the 8-wave kernel;
//   * block tile 256 pixels x 128 output channels, K chunk = 32 input channels (64-byte LDS rows), so halo (2 x 22 KB),
//     three weight tiles (3 x 8 KB) and the epilogue staging (74 KB) fit 80 KB: TWO workgroups per CU, one wave of
//     each per SIMD (256 registers per wave as before);
//   * the second workgroup of a CU starts half a tile late (once per launch, see `dephase`), so that one workgroup's
//     epilogue runs under the other's MFMA loop and the chip's epilogue traffic is spread over time.
//
// Everything else is the 8-wave kernel's design: both operands by LDS-DMA (`buffer_load ... lds`) into lane-linear
// images whose bank conflicts are removed by an XOR of the 16-byte piece index -- here with (row >> 2) & 3, four pieces
// per 64-byte row -- on the per-lane SOURCE address and on the fragment read address; zero padding through
// out-of-range buffer offsets; GroupNorm(+FiLM) affine + swish applied in LDS by the lane that fetched a piece; concat,
// nearest x2, fused 1x1 shortcut, bias, residual, GroupNorm partials of the rounded output.
// Replaces (with conv16.hip): guided_diffusion/unet.py:196-222,283-308,472-476 on the `use_fp16` torso.
#include "conv_common.h"
#include <cstdlib>

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void lds_ptr_t;

struct Conv16bArgs {
    ddnm_conv16_desc d;
    int TW, TW_log2, tiles_x, tiles_per_img;   // 2-D patch TH x TW of one image (256 pixels)
    int m_tiles, n_tiles, M;
    int Hs, Ws;
    unsigned* slot_counter;                    // [2048] arrival counters per physical CU (never reset: parity is what counts)
    int dephase_sleeps;                        // 64-cycle sleep quanta the second workgroup of a CU waits in the first round
    int first_round;                           // workgroups with blockIdx.x < first_round may be delayed
    int probe_wcontig;                         // timing probe (wrong results): read the weight tiles as contiguous 8 KB blocks
};

constexpr int B16_KC = 32;            // channels per chunk (= 64 bytes per LDS row)
constexpr int B16_ROWB = 64;
constexpr int B16_BM = 256, B16_BN = 128;
constexpr int B16_MT = 4, B16_NT = 2;
constexpr int B16_HROWS = 352;        // (8+2) x 34 = 340 or (16+2) x 18 = 324 halo rows, rounded up to 16-row DMA pieces
constexpr int B16_HGROUPS = B16_HROWS / 16;                 // 22
constexpr int B16_HG_PER_WAVE = (B16_HGROUPS + 3) / 4;      // 6
constexpr int B16_HBYTES = B16_HROWS * B16_ROWB;            // 22528
constexpr int B16_WBYTES = B16_BN * B16_ROWB;               // 8192
constexpr int B16_NWB = 3, B16_NHB = 2;
constexpr int B16_LDS_TILES = B16_NHB * B16_HBYTES + B16_NWB * B16_WBYTES;      // 69632
constexpr int B16_LDS_MAIN = B16_LDS_TILES + 256;           // + GroupNorm scale | shift of one 32-channel chunk
constexpr int B16_EPITCH = 144;       // epilogue staging pitch (bytes) per pixel of a wave's 64-channel slice
constexpr int B16_LDS_EPI = 4 * 128 * B16_EPITCH + 2 * B16_BN * 2 * 4;          // 73728 + 2048
constexpr int B16_LDS_BYTES = B16_LDS_MAIN > B16_LDS_EPI ? B16_LDS_MAIN : B16_LDS_EPI;
static_assert(B16_LDS_BYTES <= 81920, "two workgroups per CU");

__device__ __forceinline__ void b16_load(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff, char* lds_wave_base) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lds_ptr_t*)lds_wave_base, 16, voff, soff, 0, 0);
}
__device__ __forceinline__ __amdgpu_buffer_rsrc_t b16_rsrc(const void* p, unsigned bytes) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, bytes, 0x00020000);
}
constexpr unsigned B16_OOB = 0x80000000u;      // every tensor here is < 2 GB (checked by ddnm_conv16)

__global__ __launch_bounds__(256, 2) void conv16b_kernel(const Conv16bArgs p) {
    constexpr int MT = B16_MT, NT = B16_NT, BM = B16_BM, BN = B16_BN;
    __shared__ __attribute__((aligned(1024))) char lds[B16_LDS_BYTES];    // ONE shared object (keeps the DMA pipeline)
    char* const Hb = lds;
    char* const Wb = lds + B16_NHB * B16_HBYTES;

    const ddnm_conv16_desc& d = p.d;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int kh = lane >> 5;

    // ---- de-phasing: the second workgroup to arrive on a CU in the first round waits half a tile, once
    if (p.dephase_sleeps > 0 && (int)blockIdx.x < p.first_round) {
        int* flag = reinterpret_cast<int*>(lds);
        if (tid == 0) {
            unsigned hw_id, xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw_id));
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            // HW_ID: cu_id [11:8], sh_id [12], se_id [15:13]; XCC_ID [3:0]
            const unsigned cu = ((xcc & 15u) << 8) | ((hw_id >> 8) & 0xffu);
            const unsigned prev = atomicAdd(p.slot_counter + (cu & 2047u), 1u);
            *flag = (int)(prev & 1u);
        }
        __syncthreads();
        const int late = *flag;
        __syncthreads();
        if (late)
            for (int i = 0; i < p.dephase_sleeps; ++i) __builtin_amdgcn_s_sleep(127);
    }

    // Workgroup -> (pixel tile, channel tile): the two channel halves of one pixel tile are neighbours (same XCD: the
    // halo is fetched from HBM once), consecutive pixel tiles share an XCD as well.
    const int q = xcd_swizzle(blockIdx.x, gridDim.x);
    const int n_tile = q % p.n_tiles, m_tile = q / p.n_tiles;
    const int Cin = d.Cin;

    // ---- tile geometry
    const int TW = p.TW, TWl = p.TW_log2, HWd = TW + 2;
    const int img = m_tile / p.tiles_per_img;
    const int t_in = m_tile - img * p.tiles_per_img;
    const int ty_ = t_in / p.tiles_x, tx_ = t_in - ty_ * p.tiles_x;
    const int ty0 = ty_ * (BM >> TWl), tx0 = tx_ << TWl;
    const int NP = ((BM >> TWl) + 2) * HWd;

    // ---- LDS-DMA source mapping.  One instruction moves 16 rows x 64 B; lane -> (row = 16*g + lane/4, piece lane%4),
    // and the piece it FETCHES is piece ^ swizzle(row), swizzle(row) = (row >> 2) & 3 = (lane >> 4) & 3 for every g.
    const int lrow = lane >> 2, lpiece = lane & 3;
    const int lp8 = (lpiece ^ ((lane >> 4) & 3)) * 8;           // logical piece (in halfs) this lane fetches
    int hoff[B16_HG_PER_WAVE];           // source pixel index or -1 -> out-of-range offset (zero)
#pragma unroll
    for (int gi = 0; gi < B16_HG_PER_WAVE; ++gi) {
        const int row = (wave + 4 * gi) * 16 + lrow;
        int off = -1;
        if (row < NP) {
            const int hy = row / HWd, hx = row - hy * HWd;
            const int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;
            if ((unsigned)iy < (unsigned)d.H && (unsigned)ix < (unsigned)d.W) {
                const int sy = d.ups ? (iy >> 1) : iy, sx = d.ups ? (ix >> 1) : ix;
                off = (img * p.Hs + sy) * p.Ws + sx;
            }
        }
        hoff[gi] = off;
    }
    // weight rows: 8 pieces of 16 rows, wave w takes pieces w and w + 4
    const unsigned wrow0 = (unsigned)(n_tile * BN + wave * 16 + lrow);
    const unsigned src_pix = (unsigned)d.B * p.Hs * p.Ws, out_pix = (unsigned)p.M;
    const int C0 = d.src1 ? d.C0 : Cin, C1 = Cin - C0;
    const __amdgpu_buffer_rsrc_t r_src = b16_rsrc(d.src, src_pix * C0 * 2u);
    const __amdgpu_buffer_rsrc_t r_src1 = b16_rsrc(d.src1 ? d.src1 : d.src, src_pix * C1 * 2u);
    const unsigned wrows_total = (unsigned)((d.Cout + 255) / 256) * 256u;       // packed with Cout padded to 256 rows
    const __amdgpu_buffer_rsrc_t r_w = b16_rsrc(d.weight, wrows_total * 9u * Cin * 2u);

    auto issue_halo = [&](__amdgpu_buffer_rsrc_t rsrc, int cstride, int coff, int hb) {
        char* dst = Hb + hb * B16_HBYTES + wave * 1024;
#pragma unroll
        for (int gi = 0; gi < B16_HG_PER_WAVE; ++gi) {
            if (wave + 4 * gi < B16_HGROUPS) {
                const unsigned vo = hoff[gi] >= 0 ? ((unsigned)hoff[gi] * (unsigned)cstride + lp8) * 2u : B16_OOB;
                b16_load(rsrc, vo, (unsigned)coff * 2u, dst + gi * 4096);
            }
        }
    };
    // rows n = wrow0 + 64*j of the [Cout][rowlen] fp16 matrix behind `rsrc`, 32 channels at element offset `delta`
    auto issue_w = [&](__amdgpu_buffer_rsrc_t rsrc, unsigned rowlen, unsigned delta, int wb) {
        char* dst = Wb + wb * B16_WBYTES + wave * 1024;
        if (p.probe_wcontig == 1) { delta = (delta % 4096u) * 128u; rowlen = 32u; }
        const unsigned vo = (((p.probe_wcontig == 1 ? (unsigned)(wave * 16 + lrow) : wrow0)) * rowlen + lp8) * 2u;
#pragma unroll
        for (int j = 0; j < 2; ++j) b16_load(rsrc, vo, (delta + 64u * j * rowlen) * 2u, dst + j * 4096);
    };
    auto issue_main_halo = [&](int c, int hb) {
        const int cb = c * B16_KC;
        if (cb < C0) issue_halo(r_src, C0, cb, hb);
        else issue_halo(r_src1, C1, cb - C0, hb);
    };
    // ---- fused GroupNorm (+FiLM) affine + swish of the operand, applied IN LDS by the lane that fetched the piece
    const bool fuse_gn = d.gn_scale != nullptr;
    char* const Gb = lds + B16_LDS_TILES;
    f32x4 gsc0, gsc1, gsh0, gsh1;
    auto issue_gn = [&](int c) {
        if (wave == 0 && lane < 8) {
            const unsigned nb = (unsigned)d.B * Cin * 4u, vo = ((unsigned)(img * Cin + c * B16_KC) * 4u) + lane * 16u;
            b16_load(b16_rsrc(d.gn_scale, nb), vo, 0u, Gb);
            b16_load(b16_rsrc(d.gn_shift, nb), vo, 0u, Gb + 128);
        }
    };
    auto read_gn = [&]() {
        gsc0 = *reinterpret_cast<const f32x4*>(Gb + lp8 * 4); gsc1 = *reinterpret_cast<const f32x4*>(Gb + lp8 * 4 + 16);
        gsh0 = *reinterpret_cast<const f32x4*>(Gb + 128 + lp8 * 4); gsh1 = *reinterpret_cast<const f32x4*>(Gb + 128 + lp8 * 4 + 16);
    };
    auto act_group = [&](int gi, int hb) {
        if (wave + 4 * gi < B16_HGROUPS && hoff[gi] >= 0) {
            char* pl = Hb + hb * B16_HBYTES + (wave + 4 * gi) * 1024 + lane * 16;
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
        wa[j] = n * B16_ROWB + ((kh ^ ((n >> 2) & 3)) << 4);
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

    auto mfma_step = [&](int toff, int hb, int wb, auto&& after_first_kstep) {
        int pb[MT], wo[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const int qq = q0[i] + toff;
            pb[i] = qq * B16_ROWB + ((kh ^ ((qq >> 2) & 3)) << 4) + hb * B16_HBYTES;
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) wo[j] = wa[j] + B16_NHB * B16_HBYTES + wb * B16_WBYTES;
        half8 a[2][NT], b[2][MT];
#pragma unroll
        for (int j = 0; j < NT; ++j) a[0][j] = *reinterpret_cast<const half8*>(lds + wo[j]);
#pragma unroll
        for (int i = 0; i < MT; ++i) b[0][i] = *reinterpret_cast<const half8*>(lds + pb[i]);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            const int cur = ks & 1, nxt = cur ^ 1;
            if (ks == 0) {
#pragma unroll
                for (int j = 0; j < NT; ++j) a[nxt][j] = *reinterpret_cast<const half8*>(lds + (wo[j] ^ 32));
#pragma unroll
                for (int i = 0; i < MT; ++i) b[nxt][i] = *reinterpret_cast<const half8*>(lds + (pb[i] ^ 32));
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[cur][j], b[cur][i], acc[i][j], 0, 0, 0);
            if (ks == 0) {
                // one LDS read of the next k-step behind each of the first MFMAs
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

    // ---- K loop over (chunk, tap) steps, then the fused 1x1 shortcut's chunks.  Weight tiles are requested two steps
    // ahead into the buffer the step's barrier has just freed; `s_waitcnt vmcnt(2)` leaves the younger tile in flight.
    constexpr int LA = B16_NWB - 1;
    const int nchunks = Cin / B16_KC;
    const int SC = d.SC0 + d.SC1, nsk = d.skip0 ? SC / B16_KC : 0;
    const unsigned wrow = (unsigned)(9 * Cin);
    int hb = 0, wb = 0;
    bool pend = false;
    auto wait_tiles = [&]() {
        if (pend) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    };
    issue_main_halo(0, 0);
    if (fuse_gn) issue_gn(0);
    issue_w(r_w, wrow, 0u, 0);
    issue_w(r_w, wrow, (unsigned)Cin, 1);
    pend = true;
    if (fuse_gn) {
        wait_tiles();                           // the first halo (this wave's pieces) and the parameters have landed
        __builtin_amdgcn_s_barrier();
        read_gn();
#pragma unroll
        for (int gi = 0; gi < B16_HG_PER_WAVE; ++gi) act_group(gi, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
#pragma unroll 1
    for (int c = 0; c < nchunks; ++c) {
        const bool more = c + 1 < nchunks;
        int toff = 0, kx = 0;
#pragma unroll 1
        for (int tap = 0; tap < 9; ++tap) {
            wait_tiles();
            __builtin_amdgcn_s_barrier();           // this step's tiles have landed (every wave's); the oldest buffers are free
            auto issue_next = [&]() {
                pend = false;
                if (tap == 0 && more) {             // next chunk's halo (older than this group's weight tile: landed by tap 1)
                    issue_main_halo(c + 1, hb ^ 1);
                    if (fuse_gn) issue_gn(c + 1);
                }
                int t2 = tap + LA, c2 = c;
                if (t2 >= 9) { t2 -= 9; ++c2; }
                if (c2 < nchunks) {
                    issue_w(r_w, wrow, (unsigned)(t2 * Cin + c2 * B16_KC), wb + LA >= B16_NWB ? wb + LA - B16_NWB : wb + LA);
                    pend = true;
                }
            };
            mfma_step(toff, hb, wb, issue_next);
            // the next chunk's halo and parameters landed before this step's barrier (tap >= 1): one piece per tap
            if (fuse_gn && more && tap >= 1 && tap <= B16_HG_PER_WAVE) {
                if (tap == 1) read_gn();
                act_group(tap - 1, hb ^ 1);
            }
            wb = wb + 1 == B16_NWB ? 0 : wb + 1;
            if (++kx == 3) { kx = 0; toff += HWd - 2; } else { ++toff; }
        }
        hb ^= 1;
    }
    if (nsk > 0) {
        // fused 1x1 shortcut (skip_connection of a ResBlock, unet.py:222,256): extra K chunks over the block's RAW
        // input, read at the centre tap of its halo
        const __amdgpu_buffer_rsrc_t r_skw = b16_rsrc(d.skip_weight, wrows_total * SC * 2u);
        const __amdgpu_buffer_rsrc_t r_sk0 = b16_rsrc(d.skip0, out_pix * d.SC0 * 2u);
        const __amdgpu_buffer_rsrc_t r_sk1 = b16_rsrc(d.skip1 ? d.skip1 : d.skip0, out_pix * d.SC1 * 2u);
        auto issue_skip = [&](int ch, int hbuf, int wbuf) {
            const int cb = ch * B16_KC;
            if (cb < d.SC0) issue_halo(r_sk0, d.SC0, cb, hbuf);
            else issue_halo(r_sk1, d.SC1, cb - d.SC0, hbuf);
            issue_w(r_skw, (unsigned)SC, (unsigned)cb, wbuf);
        };
        __syncthreads();
        issue_skip(0, hb, wb);
#pragma unroll 1
        for (int ch = 0; ch < nsk; ++ch) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            const int wbn = wb + 1 == B16_NWB ? 0 : wb + 1;
            if (ch + 1 < nsk) issue_skip(ch + 1, hb ^ 1, wbn);
            mfma_step(HWd + 1, hb, wb, [] {});
            wb = wbn;
            hb ^= 1;
        }
    }
    // residual tile of the epilogue: requested before the barrier and the LDS transposition
    constexpr int ITS = MT * 32 / 8;                 // lane -> (pixel = it*8 + lane/8, 8 channels = piece lane%8)
    const int er = lane >> 3, ep = lane & 7;
    const _Float16* const res = reinterpret_cast<const _Float16*>(d.res);
    _Float16* const out = reinterpret_cast<_Float16*>(d.out);
    const int cbase = n_tile * BN + wn * 64;
    const bool wave_on = cbase < d.Cout;            // Cout % 64 == 0
    const int chn = cbase + ep * 8;
    int opix[ITS];
    uint4 rv[ITS];
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
        const int m = wm * MT * 32 + it * 8 + er;
        const int oy = ty0 + (m >> TWl), ox = tx0 + (m & (TW - 1));
        int pix = (img * d.H + oy) * d.W + ox;
        const int rpix = d.res_ups ? (img * (d.H >> 1) + (oy >> 1)) * (d.W >> 1) + (ox >> 1) : pix;
        if (!wave_on) pix = -1;
        opix[it] = pix;
        rv[it] = uint4{0u, 0u, 0u, 0u};
        if (res && pix >= 0) rv[it] = *reinterpret_cast<const uint4*>(res + (size_t)rpix * d.Cout + chn);
    }
    __syncthreads();                   // all fragment reads done: LDS becomes the epilogue's staging area
    if (p.probe_wcontig == 2) {        // timing probe: no epilogue (accumulators kept live)
        float t = 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) t += acc[i][j][r];
        if (t == 12345.678f) out[tid] = (_Float16)t;
        return;
    }

    // D layout (32x32 MFMA, A = weights): lane -> pixel = lane & 31 of M tile i, channels
    // wn*64 + j*32 + 8*(r>>2) + 4*(lane>>5) + (r&3).
    char* const stage = lds + wave * (MT * 32) * B16_EPITCH;
    float* const stat_lds = reinterpret_cast<float*>(lds + 4 * (MT * 32) * B16_EPITCH);
    if (wave_on) {
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            f32x4 bias4[4];
#pragma unroll
            for (int rg = 0; rg < 4; ++rg) {
                const int ch = cbase + j * 32 + 8 * rg + 4 * kh;
                bias4[rg] = d.bias ? *reinterpret_cast<const f32x4*>(d.bias + ch) : f32x4{0.f, 0.f, 0.f, 0.f};
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int rg = 0; rg < 4; ++rg) {
                    const f32x4 b4 = bias4[rg];
                    char* dst = stage + (i * 32 + (lane & 31)) * B16_EPITCH + (j * 32 + 8 * rg + 4 * kh) * 2;
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
        half8 v = *reinterpret_cast<const half8*>(stage + (it * 8 + er) * B16_EPITCH + ep * 16);
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
    if (d.stats_out) {
        // GroupNorm partials of the tensor just written (of the ROUNDED values the consumer will read): reduce over the
        // 8 pixel rows of a wave (lanes with equal lane%8), then over the two pixel-halves (wm)
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
                stat_lds[(wm * BN + c) * 2 + 0] = cs[e];
                stat_lds[(wm * BN + c) * 2 + 1] = cq[e];
            }
        }
        __syncthreads();
        if (tid < BN) {
            const int n = n_tile * BN + tid;
            if (n < d.Cout) {
                const float a = stat_lds[tid * 2] + stat_lds[(BN + tid) * 2];
                const float qq = stat_lds[tid * 2 + 1] + stat_lds[(BN + tid) * 2 + 1];
                *reinterpret_cast<float2*>(d.stats_out + ((size_t)m_tile * d.Cout + n) * 2) = float2{a, qq};
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// host side (called by ddnm_conv16 in conv16.hip for the launches its plan hands over)
// ---------------------------------------------------------------------------------------------
static unsigned* g_slot_counter = nullptr;

// 1: this launch can take the two-workgroups-per-CU kernel (3x3, 256-pixel tiles, no split-K, Cout % 128 == 0)
int conv16b_eligible(const ddnm_conv16_desc* d, int tw, int tiles_256) {
    if (d->ksize != 3 || d->out_nchw_f32 || d->Cout % 128 || d->Cin % B16_KC) return 0;
    if (d->src1 && d->C0 % B16_KC) return 0;
    if (d->skip0 && (d->SC0 % B16_KC || d->SC1 % B16_KC)) return 0;
    if (tw != 32 && tw != 16) return 0;
    return tiles_256 > 0 ? 1 : 0;
}

int conv16b_launch(const ddnm_conv16_desc* d, int tw, int tw_log2, int tiles_x, int tiles_per_img, int m_tiles,
                   int dephase, hipStream_t s) {
    if (!g_slot_counter) {
        if (hipMalloc(reinterpret_cast<void**>(&g_slot_counter), 2048 * sizeof(unsigned)) != hipSuccess) return DDNM_E_BADARG;
        if (hipMemset(g_slot_counter, 0, 2048 * sizeof(unsigned)) != hipSuccess) return DDNM_E_BADARG;
    }
    Conv16bArgs p;
    p.d = *d;
    p.TW = tw; p.TW_log2 = tw_log2; p.tiles_x = tiles_x; p.tiles_per_img = tiles_per_img;
    p.m_tiles = m_tiles; p.n_tiles = d->Cout / B16_BN;
    p.M = d->B * d->H * d->W;
    p.Hs = d->ups ? d->H / 2 : d->H;
    p.Ws = d->ups ? d->W / 2 : d->W;
    p.slot_counter = g_slot_counter;
    const int grid = p.m_tiles * p.n_tiles;
    // half a tile of the steady state: a workgroup shares its SIMDs with one other, so a step of 16 MFMAs per wave
    // (32 cycles each) takes ~1024 cycles; the tile has 9 * Cin/32 (+ shortcut) steps; s_sleep 127 ~ 8128 cycles
    const long steps = 9L * (d->Cin / B16_KC) + (d->skip0 ? (d->SC0 + d->SC1) / B16_KC : 0);
    p.dephase_sleeps = (dephase && grid > 512) ? (int)((steps * 1024 / 2 + 8127) / 8128) : 0;
    p.first_round = 512;
    {
        static int pc = -1;
        if (pc < 0) { const char* e = getenv("DDNM_P16B_WCONTIG"); pc = e ? atoi(e) : 0; }
        p.probe_wcontig = pc;
    }
    DDNM_LAUNCH(conv16b_kernel, dim3(grid), dim3(256), 0, s, p);
    return 0;
}
