// Ceiling ladder for the two dominant convolution kernels (development tool, round 6; results: profiles/r06_mfma_ceiling.md).
//
// Bottom-up: a loop of v_mfma_f32_32x32x16_f16 on RANDOM operands, to which the ingredients of the real kernels' main
// loops are added ONE AT A TIME, in the real kernels' geometry (LDS images, fragment addresses, request counts), with
// no tile prologue / epilogue: 256 persistent workgroups (one per CU) run one long K loop.
//
//   rung 1  registers only: the wave's MFMA stream on 8 random fragments held in registers
//   rung 2  + the fragment ds_read_b128 stream of every k-step from a static (random) LDS image
//   rung 3  + one s_barrier per tap (+ lgkmcnt(0)), as the tap loop has it
//   rung 4  + the weight tile of tap + 2 by LDS-DMA (buffer_load ... lds, L2-resident weights), counted vmcnt
//   rung 5  + the next chunk's halo: global fp32 loads (HBM), GroupNorm affine + swish, hi | lo split, ds_write
//           into the other halo buffer (split geometry only) = the product kernel's main loop without tiles
//   rung 6  rung 5 in the product's launch shape: 2048 workgroups x 4 chunks (Cin = 128) -- tile prologue (first halo,
//           first two weight tiles) and wave-count ramp included, still no epilogue
//
// Geometries:  S = conv3x3_halo_f16_kernel<4,2,2,2,..,SPLIT> (8 waves, 64 x 64 wave tile, per k-step 8 reads + 12 MFMAs,
//              2 k-steps per tap, weight tile 16 KB);  C = conv16_kernel<9,4,4> (8 waves, 128 x 64 wave tile, per k-step
//              6 reads + 8 MFMAs, 4 k-steps per tap, weight tile 32 KB, two buffers);  W = waves per workgroup (4: one per
//              SIMD, 8: two per SIMD).
// Per rung: wall time over the timed launches (HIP events), MFMA work in TFLOP/s, effective shader clock
// (s_memtime / s_memrealtime inside the kernel) and MFMA-busy = 32 cycles x MFMAs per SIMD / shader cycles.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/mfma_ceiling.hip -o tools/_build/mfma_ceiling && tools/_build/mfma_ceiling
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

struct Args {
    const float* src;          // [B][256][256][Cin] fp32 NHWC (rung >= 5)
    const char* weight;        // packed weight rows (rung >= 4)
    const float* gn;           // [Cin] scale, [Cin] shift
    const half8* frag;         // random fragments (rung 1) / LDS fill
    float* out;                // one float per thread (keeps the accumulators alive)
    unsigned long long* clk;   // per workgroup: shader cycles, 100 MHz ticks
    int chunks;                // K chunks per workgroup
    int Cin;
    int tiles_x;               // tiles per image row (rung >= 5 addressing)
};

__device__ __forceinline__ float silu_f(float v) { return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v)); }

// GEO 0 = split (S), 1 = conv16-like (C)
// PIPE = 1 (split geometry): explicit two-set fragment pipeline -- the 8 reads of k-step n + 1 are issued under the 12 MFMAs of
// k-step n (the second k-step of a tap runs behind the NEXT tap's barrier), placed by sched_group_barrier
template <int GEO, int RUNG, int NWAVES, int PIPE = 0>
__global__ __launch_bounds__(NWAVES * 64, NWAVES / 4) void ladder_kernel(const Args p) {
    constexpr bool S = GEO == 0;
    constexpr int MT = S ? 2 : 4, NT = 2;
    constexpr int WN = S ? 2 : 4, WM = NWAVES / WN;          // S: 4 x 2 (8 waves) or 2 x 2; C: 2 x 4 or 1 x 4
    constexpr int BN = WN * NT * 32;                          // 128 / 256
    constexpr int KSTEPS = S ? 2 : 4;                         // k-steps of 16 per tap and chunk
    constexpr int LDH = 72;                                   // halo row pitch in halfs (144 B)
    constexpr int MAXH = 340, HWd = 34;
    constexpr int WTILE = BN * 128;
    constexpr int NWB = S ? 3 : 2;
    constexpr int NTHREADS = NWAVES * 64;
    constexpr int BR = BN / (NTHREADS / 8);                   // LDS-DMA instructions per wave and weight tile
    __shared__ __attribute__((aligned(1024))) char lds_all[NWB * WTILE + 2 * MAXH * LDH * 2];
    char* const Bs = lds_all;
    _Float16* const Hs = reinterpret_cast<_Float16*>(lds_all + NWB * WTILE);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WN, wn = wave % WN;

    // random LDS image
    for (int i = tid; i < (int)(sizeof(lds_all) / 16); i += NTHREADS)
        reinterpret_cast<half8*>(lds_all)[i] = p.frag[(i * 7 + blockIdx.x) & 4095];
    __syncthreads();

    int a_off[MT];
#pragma unroll
    for (int i = 0; i < MT; ++i) {
        const int m = ((wm * MT + i) * 32 + (lane & 31)) & 255;
        const int ty = m >> 5, tx = m & 31;
        a_off[i] = (ty * HWd + tx) * LDH + (lane >> 5) * 8;
    }
    const int b_frag = ((wn * NT * 32 + (lane & 31)) * 128) + ((((lane >> 5) ^ (((lane & 31) >> 1) & 7))) << 4);

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // rung 1 operands
    half8 rah[MT], ral[MT], rbh[NT], rbl[NT];
#pragma unroll
    for (int i = 0; i < MT; ++i) { rah[i] = p.frag[(tid * 4 + i) & 4095]; ral[i] = p.frag[(tid * 4 + i + 1777) & 4095]; }
#pragma unroll
    for (int j = 0; j < NT; ++j) { rbh[j] = p.frag[(tid * 4 + j + 911) & 4095]; rbl[j] = p.frag[(tid * 4 + j + 2999) & 4095]; }

    // ---- weight stream (rung >= 4): the product's request pattern
    const int lrow = lane >> 3, lpiece = lane & 7;
    const int wswz = (((wave & 1) << 2) | (lrow >> 1));
    const unsigned w_rowlen = 9u * (unsigned)p.Cin * (S ? 4u : 2u);
    const __amdgpu_buffer_rsrc_t r_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(p.weight), 0, (unsigned)BN * w_rowlen, 0x00020000);
    const unsigned w_voff = (unsigned)(wave * 8 + lrow) * w_rowlen + (unsigned)((lpiece ^ wswz) * 16);
    auto issue_w = [&](int chunk, int tap, int buf) {
        char* dst = Bs + buf * WTILE + wave * 1024;
        const unsigned so = ((unsigned)tap * p.Cin + (unsigned)chunk * (S ? 32 : 64)) * (S ? 4u : 2u);
#pragma unroll
        for (int j = 0; j < BR; ++j)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(r_w, (__attribute__((address_space(3))) void*)(dst + j * (NTHREADS / 64) * 1024), 16,
                                                     w_voff, so + (unsigned)j * (NTHREADS / 8) * w_rowlen, 0, 0);
    };

    // ---- halo stream (rung >= 5, split geometry): thread -> (16-byte column hc of 8, rows prow + 64 i)
    constexpr int HCOLS = 8, HROWS_PER_PASS = NTHREADS / HCOLS, HR = (MAXH + HROWS_PER_PASS - 1) / HROWS_PER_PASS, HSPLIT = (HR + 1) / 2;
    const int hc = tid % HCOLS, prow = tid / HCOLS;
    int hoff[HR];
    uint4 h_st[HR];
    f32x4 gsc = {1.f, 1.f, 1.f, 1.f}, gsh = {0.f, 0.f, 0.f, 0.f};
    const int tiles_per_img = p.tiles_x * 32;                 // 256 / 8 tile rows
    const __amdgpu_buffer_rsrc_t r_s = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.src), 0, 0x7fffffffu, 0x00020000);
    auto set_tile = [&](int tile) {
        const int img = tile / tiles_per_img, t = tile - img * tiles_per_img;
        const int ty0 = (t / p.tiles_x) * 8, tx0 = (t % p.tiles_x) * 32;
#pragma unroll
        for (int i = 0; i < HR; ++i) {
            const int row = prow + HROWS_PER_PASS * i;
            const int hy = row / HWd, hx = row - hy * HWd;
            const int iy = ty0 - 1 + hy, ix = tx0 - 1 + hx;
            const bool ok = row < MAXH && (unsigned)iy < 256u && (unsigned)ix < 256u;
            hoff[i] = ok ? (img * 256 + iy) * 256 + ix : -1;
        }
    };
    auto prefetch_halo_part = [&](int chunk, int i0, int i1) {
        const int cb = chunk * 32;
#pragma unroll
        for (int i = 0; i < HR; ++i) {
            if (i < i0 || i >= i1) continue;
            const unsigned vo = hoff[i] >= 0 ? ((unsigned)hoff[i] * p.Cin + hc * 4) * 4 : 0x80000000u;
            const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r_s, vo, cb * 4, 0);
            h_st[i] = uint4{v.x, v.y, v.z, v.w};
        }
        if (i0 == 0) {
            gsc = *reinterpret_cast<const f32x4*>(p.gn + cb + hc * 4);
            gsh = *reinterpret_cast<const f32x4*>(p.gn + p.Cin + cb + hc * 4);
        }
    };
    auto stage_halo_part = [&](int hbuf, int i0, int i1) {
#pragma unroll
        for (int i = 0; i < HR; ++i) {
            if (i < i0 || i >= i1) continue;
            const int row = prow + HROWS_PER_PASS * i;
            if (row < MAXH) {
                _Float16* dst = &Hs[hbuf * MAXH * LDH + row * LDH + hc * 4];
                f32x4 v = __builtin_bit_cast(f32x4, h_st[i]);
                if (hoff[i] >= 0) {
                    v = v * gsc + gsh;
                    v.x = silu_f(v.x); v.y = silu_f(v.y); v.z = silu_f(v.z); v.w = silu_f(v.w);
                }
                const half4 h = {(_Float16)v.x, (_Float16)v.y, (_Float16)v.z, (_Float16)v.w};
                const half4 l = {(_Float16)(v.x - (float)h.x), (_Float16)(v.y - (float)h.y), (_Float16)(v.z - (float)h.z), (_Float16)(v.w - (float)h.w)};
                *reinterpret_cast<half4*>(dst) = h;
                *reinterpret_cast<half4*>(dst + 32) = l;
            }
        }
    };

    auto mfma_tap = [&](int tap, int buf, int hbuf) {
        const int ky = tap / 3, kx = tap - 3 * ky;
        const int tap_off = (ky * HWd + kx) * LDH + hbuf * MAXH * LDH;
        const char* bf = Bs + buf * WTILE;
#pragma unroll
        for (int ks = 0; ks < KSTEPS; ++ks) {
            half8 ah[MT], al[MT], bh[NT], bl[NT];
            if constexpr (RUNG >= 2) {
#pragma unroll
                for (int i = 0; i < MT; ++i) {
                    ah[i] = *reinterpret_cast<const half8*>(Hs + a_off[i] + tap_off + ks * 16);
                    if constexpr (S) al[i] = *reinterpret_cast<const half8*>(Hs + a_off[i] + tap_off + ks * 16 + 32);
                }
#pragma unroll
                for (int j = 0; j < NT; ++j) {
                    bh[j] = *reinterpret_cast<const half8*>(bf + ((b_frag ^ (ks << 5)) + j * 32 * 128));
                    if constexpr (S) bl[j] = *reinterpret_cast<const half8*>(bf + ((b_frag ^ ((ks + 2) << 5)) + j * 32 * 128));
                }
            } else {
#pragma unroll
                for (int i = 0; i < MT; ++i) { ah[i] = rah[i]; al[i] = ral[i]; }
#pragma unroll
                for (int j = 0; j < NT; ++j) { bh[j] = rbh[j]; bl[j] = rbl[j]; }
            }
            if constexpr (S) {
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], acc[i][j], 0, 0, 0);
            }
#pragma unroll
            for (int i = 0; i < MT; ++i)
#pragma unroll
                for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
        }
    };

    // ---- PIPE: two fragment sets
    half8 fah[2][MT], fal[2][MT], fbh[2][NT], fbl[2][NT];
#pragma unroll
    for (int q = 0; q < 2; ++q) {
#pragma unroll
        for (int i = 0; i < MT; ++i) { fah[q][i] = rah[i]; fal[q][i] = ral[i]; }
#pragma unroll
        for (int j = 0; j < NT; ++j) { fbh[q][j] = rbh[j]; fbl[q][j] = rbl[j]; }
    }
    auto read_set = [&](auto SET, int tap, int ks, int buf, int hbuf) {
        constexpr int q = decltype(SET)::value;
        const int ky = tap / 3, kx = tap - 3 * ky;
        const int tap_off = (ky * HWd + kx) * LDH + hbuf * MAXH * LDH;
        const char* bf = Bs + buf * WTILE;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            fah[q][i] = *reinterpret_cast<const half8*>(Hs + a_off[i] + tap_off + ks * 16);
            fal[q][i] = *reinterpret_cast<const half8*>(Hs + a_off[i] + tap_off + ks * 16 + 32);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            fbh[q][j] = *reinterpret_cast<const half8*>(bf + ((b_frag ^ (ks << 5)) + j * 32 * 128));
            fbl[q][j] = *reinterpret_cast<const half8*>(bf + ((b_frag ^ ((ks + 2) << 5)) + j * 32 * 128));
        }
    };
    auto mfma_set = [&](auto SET) {
        constexpr int q = decltype(SET)::value;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fal[q][i], fbh[q][j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[q][i], fbl[q][j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fah[q][i], fbh[q][j], acc[i][j], 0, 0, 0);
    };
    auto interleave = [&]() {            // 8 x (1 LDS read, 1 MFMA), then the remaining 4 MFMAs
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
    };
    auto pipe_tap = [&](int tap, int buf, int hbuf) {
        read_set(std::integral_constant<int, 0>{}, tap, 0, buf, hbuf);
        mfma_set(std::integral_constant<int, 1>{});       // the previous tap's second k-step
        interleave();
        __builtin_amdgcn_sched_barrier(0);
        read_set(std::integral_constant<int, 1>{}, tap, 1, buf, hbuf);
        mfma_set(std::integral_constant<int, 0>{});
        interleave();
        __builtin_amdgcn_sched_barrier(0);
    };

    if (S && wave >= NWAVES / 2 && NWAVES == 8) __builtin_amdgcn_s_setprio(1);      // the product's static priority
    constexpr bool TILED = RUNG >= 6;
    const unsigned long long t0 = __builtin_readcyclecounter(), r0 = wall_clock64();
    const int tiles = TILED ? 8 : 1;                         // rung 6: this workgroup walks 8 "tiles" of p.chunks each
    for (int tile = 0; tile < tiles; ++tile) {
        if constexpr (RUNG >= 5) set_tile((blockIdx.x + tile * gridDim.x) % (8 * tiles_per_img));
        if constexpr (RUNG >= 5) { prefetch_halo_part(0, 0, HR); }
        if constexpr (RUNG >= 4) { issue_w(0, 0, 0); if (NWB == 3) issue_w(0, 1, 1); }
        if constexpr (RUNG >= 5) stage_halo_part(0, 0, HR);
        int hb = 0, wb2 = 0;
        for (int chunk = 0; chunk < p.chunks; ++chunk) {
            const int ch = chunk % (p.Cin / (S ? 32 : 64));
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int cur = tap % NWB;
                if constexpr (RUNG >= 4) {
                    if constexpr (NWB == 3) {
                        if (RUNG >= 5 && (tap == 1 || tap == 2)) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(BR + HSPLIT + 2) : "memory");
                        else if (RUNG >= 5 && (tap == 4 || tap == 5)) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(BR + HR - HSPLIT) : "memory");
                        else asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(BR) : "memory");
                    } else {
                        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
                    }
                } else if constexpr (RUNG >= 3) {
                    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                }
                if constexpr (RUNG >= 3) __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                if constexpr (RUNG >= 4) {
                    const int q = tap + (NWB - 1);                  // tile of tap + 2 (S) / tap + 1 (C)
                    issue_w(q >= 9 ? (ch + 1) % (p.Cin / (S ? 32 : 64)) : ch, q % 9, NWB == 3 ? (cur >= 1 ? cur - 1 : 2) : wb2 ^ 1);
                    asm volatile("" ::: "memory");
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (RUNG >= 5 && S) {
                    const int nch = (ch + 1) % (p.Cin / 32);
                    if (tap == 0) prefetch_halo_part(nch, 0, HSPLIT);
                    if (tap == 3) { stage_halo_part(hb ^ 1, 0, HSPLIT); prefetch_halo_part(nch, HSPLIT, HR); }
                    if (tap == 6) stage_halo_part(hb ^ 1, HSPLIT, HR);
                    __builtin_amdgcn_sched_barrier(0);
                }
                if constexpr (PIPE == 1) pipe_tap(tap, cur, RUNG >= 5 ? hb : 0);
                else mfma_tap(tap, NWB == 3 ? cur : wb2, RUNG >= 5 ? hb : 0);
                wb2 ^= 1;
            }
            if constexpr (RUNG >= 5) hb ^= 1;
        }
        if constexpr (RUNG >= 4) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
    }
    const unsigned long long t1 = __builtin_readcyclecounter(), r1 = wall_clock64();
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) s += acc[i][j][r];
    p.out[blockIdx.x * NTHREADS + tid] = s;
    if (tid == 0) { p.clk[2 * blockIdx.x] = t1 - t0; p.clk[2 * blockIdx.x + 1] = r1 - r0; }
}

struct Bufs {
    float* src; char* weight; float* gn; half8* frag; float* out; unsigned long long* clk;
};

template <int GEO, int RUNG, int NWAVES, int PIPE = 0>
static void run(const Bufs& b, const char* what) {
    constexpr bool S = GEO == 0;
    constexpr int MT = S ? 2 : 4, NT = 2, KSTEPS = S ? 2 : 4;
    const int grid = 256;
    Args a;
    a.src = b.src; a.weight = b.weight; a.gn = b.gn; a.frag = b.frag; a.out = b.out; a.clk = b.clk;
    a.Cin = 128; a.tiles_x = 8;
    a.chunks = RUNG >= 6 ? 4 : (NWAVES == 8 ? 48 : 96);
    const double mfma_per_wave = (double)(RUNG >= 6 ? 8 : 1) * a.chunks * 9 * KSTEPS * MT * NT * (S ? 3 : 1);
    const double flop = (double)grid * NWAVES * mfma_per_wave * 2.0 * 32 * 32 * 16;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    const int warm = 40, reps = 40;                            // ~50 ms of warm-up: the clock settles to the power budget
    for (int i = 0; i < warm; ++i) hipLaunchKernelGGL((ladder_kernel<GEO, RUNG, NWAVES, PIPE>), dim3(grid), dim3(NWAVES * 64), 0, 0, a);
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((ladder_kernel<GEO, RUNG, NWAVES, PIPE>), dim3(grid), dim3(NWAVES * 64), 0, 0, a);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    if (hipGetLastError() != hipSuccess) { printf("launch failed: %s\n", what); exit(1); }
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> clk(2 * grid);
    hipMemcpy(clk.data(), b.clk, sizeof(unsigned long long) * 2 * grid, hipMemcpyDeviceToHost);
    double cyc = 0, real = 0;
    for (int i = 0; i < grid; ++i) { cyc += (double)clk[2 * i]; real += (double)clk[2 * i + 1]; }
    const double ghz = cyc / real * 0.1;                       // s_memrealtime ticks at 100 MHz
    const double busy = mfma_per_wave * (NWAVES / 4) * 32.0 / (cyc / grid);
    const double tf = flop * reps / (ms * 1e-3) / 1e12;
    printf("| %c | %d | %d | %-58s | %8.1f | %7.1f | %5.3f | %5.2f | %5.3f |\n", S ? 'S' : 'C', RUNG, NWAVES / 4, what, ms * 1e3 / reps, tf, tf / 2500.0, ghz, busy);
    fflush(stdout);
}

int main() {
    Bufs b;
    const size_t src_floats = (size_t)8 * 256 * 256 * 128;
    hipMalloc(&b.src, src_floats * 4);
    hipMalloc(&b.weight, 256 * 9 * 128 * 4);
    hipMalloc(&b.gn, 2 * 128 * 4);
    hipMalloc(&b.frag, 4096 * 16);
    hipMalloc(&b.out, 256 * 512 * 4);
    hipMalloc(&b.clk, 256 * 16);
    {   // random data: N(0,1)-like activations, fp16 fragments / weights with random mantissas in [0.5, 2) and both signs
        std::vector<float> h(src_floats);
        unsigned s = 12345u;
        auto rnd = [&]() { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return s; };
        for (auto& v : h) { float u = 0.f; for (int k = 0; k < 4; ++k) u += (float)(rnd() & 0xffff) / 65536.f; v = (u - 2.f) * 1.7f; }
        hipMemcpy(b.src, h.data(), src_floats * 4, hipMemcpyHostToDevice);
        std::vector<unsigned short> w(256 * 9 * 128 * 2);
        for (auto& v : w) v = (unsigned short)(0x3800u | (rnd() & 0x87ffu));
        hipMemcpy(b.weight, w.data(), w.size() * 2, hipMemcpyHostToDevice);
        std::vector<unsigned short> f(4096 * 8);
        for (auto& v : f) v = (unsigned short)(0x3800u | (rnd() & 0x87ffu));
        hipMemcpy(b.frag, f.data(), f.size() * 2, hipMemcpyHostToDevice);
        std::vector<float> g(256);
        for (int i = 0; i < 128; ++i) { g[i] = 0.8f + 0.4f * (float)(rnd() & 255) / 255.f; g[128 + i] = 0.2f * ((float)(rnd() & 255) / 255.f - 0.5f); }
        hipMemcpy(b.gn, g.data(), 1024, hipMemcpyHostToDevice);
    }
    printf("| geo | rung | waves/SIMD | what | us / launch | MFMA TFLOP/s | of 2500 | GHz | MFMA-busy |\n|---|---|---|---|---:|---:|---:|---:|---:|\n");
    run<0, 1, 4>(b, "registers only");
    run<0, 1, 8>(b, "registers only");
    run<0, 2, 4>(b, "+ fragment ds_read_b128 (8 per 12 MFMAs)");
    run<0, 2, 8>(b, "+ fragment ds_read_b128 (8 per 12 MFMAs)");
    run<0, 3, 8>(b, "+ s_barrier per tap (24 MFMAs per wave)");
    run<0, 4, 8>(b, "+ LDS-DMA weight tile per tap, 3 deep, counted vmcnt");
    run<0, 5, 8>(b, "+ halo: HBM loads, GroupNorm + swish, split, ds_write");
    run<0, 6, 8>(b, "the same as 8 tiles x 4 chunks per workgroup (launch shape)");
    run<0, 2, 8, 1>(b, "PIPELINED rung 2: reads of k-step n+1 under the MFMAs of n");
    run<0, 3, 8, 1>(b, "PIPELINED rung 3 (+ barrier)");
    run<0, 4, 8, 1>(b, "PIPELINED rung 4 (+ LDS-DMA weights)");
    run<0, 5, 8, 1>(b, "PIPELINED rung 5 (+ halo staging)");
    run<0, 6, 8, 1>(b, "PIPELINED rung 6 (launch shape)");
    run<1, 1, 4>(b, "registers only");
    run<1, 1, 8>(b, "registers only");
    run<1, 2, 4>(b, "+ fragment ds_read_b128 (6 per 8 MFMAs)");
    run<1, 2, 8>(b, "+ fragment ds_read_b128 (6 per 8 MFMAs)");
    run<1, 3, 8>(b, "+ s_barrier per tap (32 MFMAs per wave)");
    run<1, 4, 8>(b, "+ LDS-DMA weight tile (32 KB) per tap, 2 buffers, vmcnt(0)");
    return 0;
}
