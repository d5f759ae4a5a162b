// Batched GEMM on v_mfma_f32_32x32x2_f32 and row softmax, gfx950.
//   C[i] = alpha * A[i] * op(B[i]) + beta * D[i]
// Same LDS tile geometry as the convolution (k-contiguous rows, pitch 36 floats, one
// ds_read_b128 per 4 MFMAs).  Used for the attention products (QK^T, PV) and for the
// separable projection of the bicubic SR operator (Ae X Ae^T, Pe R Pe^T).
// Replaces torch.bmm / einsum (guided_diffusion/models.py:171-185, unet.py:344-354) and
// torch.matmul in functions/svd_operators.py:853-859.
#include "common.h"

template <int MT, int NT>   // 2x2 waves, MT x NT tiles of 32x32 per wave
__global__ __launch_bounds__(256) void bgemm_f32_kernel(const ddnm_gemm_desc d, int m_tiles, int n_tiles) {
    constexpr int BM = 2 * MT * 32, BN = 2 * NT * 32;
    constexpr int AR = BM / 32, BR = BN / 32;
    __shared__ __attribute__((aligned(16))) float As[BM * LDT];
    __shared__ __attribute__((aligned(16))) float Bs[BN * LDT];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int per_batch = m_tiles * n_tiles;
    const int bi = blockIdx.x / per_batch, t = blockIdx.x - bi * per_batch;
    const int m_tile = t / n_tiles, n_tile = t - m_tile * n_tiles;
    const int bo = bi / d.inner, bn = bi - bo * d.inner;
    // transa: A stored [K][M] (lda = row pitch of that storage)
    const float* A = d.A + bo * d.sAo + bn * d.sAi + (d.transa ? (size_t)(m_tile * BM) : (size_t)(m_tile * BM) * d.lda);
    const float* Bm = d.Bm + bo * d.sBo + bn * d.sBi;
    const int c4 = tid & 7, row0 = tid >> 3;

    f32x16 acc[MT][NT];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int frag_off = (lane & 31) * LDT + (lane >> 5) * 4;
    const float* a_frag = As + (wm * MT * 32) * LDT + frag_off;
    const float* b_frag = Bs + (wn * NT * 32) * LDT + frag_off;

    for (int k0 = 0; k0 < d.K; k0 += KC) {
        f32x4 a_st[AR], b_st[BR];
        if (!d.transa) {
#pragma unroll
            for (int i = 0; i < AR; ++i)
                a_st[i] = *reinterpret_cast<const f32x4*>(A + (size_t)(row0 + 32 * i) * d.lda + k0 + c4 * 4);
        } else {          // 32 k-rows x BM/4 float4 along m
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                const int idx = tid + 256 * i;
                const int kr = idx / (BM / 4), m4 = idx - kr * (BM / 4);
                a_st[i] = *reinterpret_cast<const f32x4*>(A + (size_t)(k0 + kr) * d.lda + m4 * 4);
            }
        }
        if (d.transb) {   // B stored [N][K]: same pattern as A
#pragma unroll
            for (int i = 0; i < BR; ++i)
                b_st[i] = *reinterpret_cast<const f32x4*>(Bm + (size_t)(n_tile * BN + row0 + 32 * i) * d.ldb + k0 + c4 * 4);
        } else {          // B stored [K][N]: 32 k-rows x BN/4 float4 = 8*BN float4, BR per thread
#pragma unroll
            for (int i = 0; i < BR; ++i) {
                const int idx = tid + 256 * i;
                const int kr = idx / (BN / 4), n4 = idx - kr * (BN / 4);
                b_st[i] = *reinterpret_cast<const f32x4*>(Bm + (size_t)(k0 + kr) * d.ldb + n_tile * BN + n4 * 4);
            }
        }
        __syncthreads();
        if (!d.transa) {
#pragma unroll
            for (int i = 0; i < AR; ++i) *reinterpret_cast<f32x4*>(&As[(row0 + 32 * i) * LDT + c4 * 4]) = a_st[i];
        } else {
#pragma unroll
            for (int i = 0; i < AR; ++i) {
                const int idx = tid + 256 * i;
                const int kr = idx / (BM / 4), m4 = idx - kr * (BM / 4);
                As[(m4 * 4 + 0) * LDT + kr] = a_st[i].x;
                As[(m4 * 4 + 1) * LDT + kr] = a_st[i].y;
                As[(m4 * 4 + 2) * LDT + kr] = a_st[i].z;
                As[(m4 * 4 + 3) * LDT + kr] = a_st[i].w;
            }
        }
        if (d.transb) {
#pragma unroll
            for (int i = 0; i < BR; ++i) *reinterpret_cast<f32x4*>(&Bs[(row0 + 32 * i) * LDT + c4 * 4]) = b_st[i];
        } else {          // transpose while staging: Bs is [n][k]
#pragma unroll
            for (int i = 0; i < BR; ++i) {
                const int idx = tid + 256 * i;
                const int kr = idx / (BN / 4), n4 = idx - kr * (BN / 4);
                Bs[(n4 * 4 + 0) * LDT + kr] = b_st[i].x;
                Bs[(n4 * 4 + 1) * LDT + kr] = b_st[i].y;
                Bs[(n4 * 4 + 2) * LDT + kr] = b_st[i].z;
                Bs[(n4 * 4 + 3) * LDT + kr] = b_st[i].w;
            }
        }
        __syncthreads();
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

    float* C = d.C + bo * d.sCo + bn * d.sCi;
    const float* D = d.D ? d.D + bo * d.sDo + bn * d.sDi : nullptr;
    const int ncol = lane & 31, rsel = 4 * (lane >> 5);
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j) {
            const int n = n_tile * BN + (wn * NT + j) * 32 + ncol;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m_tile * BM + (wm * MT + i) * 32 + (r & 3) + 8 * (r >> 2) + rsel;
                float v = d.alpha * acc[i][j][r];
                if (D) v += d.beta * D[(size_t)m * d.ldd + n];
                C[(size_t)m * d.ldc + n] = v;
            }
        }
}


// Fallback for shapes the MFMA tiling does not cover (tiny test sizes): one thread per output.
__global__ __launch_bounds__(256) void bgemm_naive_kernel(const ddnm_gemm_desc d) {
    const int64_t per = (int64_t)d.M * d.N;
    const int64_t total = per * d.batch;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t bi = i / per, r = i - bi * per;
        const int m = (int)(r / d.N), n = (int)(r - (int64_t)m * d.N);
        const int64_t bo = bi / d.inner, bn = bi - bo * d.inner;
        const float* A = d.A + bo * d.sAo + bn * d.sAi + (d.transa ? (size_t)m : (size_t)m * d.lda);
        const size_t ak = d.transa ? (size_t)d.lda : 1;
        const float* Bm = d.Bm + bo * d.sBo + bn * d.sBi;
        float acc = 0.f;
        if (d.transb) {
            const float* Br = Bm + (size_t)n * d.ldb;
            for (int k = 0; k < d.K; ++k) acc += A[k * ak] * Br[k];
        } else {
            for (int k = 0; k < d.K; ++k) acc += A[k * ak] * Bm[(size_t)k * d.ldb + n];
        }
        float v = d.alpha * acc;
        if (d.D) v += d.beta * d.D[bo * d.sDo + bn * d.sDi + (size_t)m * d.ldd + n];
        d.C[bo * d.sCo + bn * d.sCi + (size_t)m * d.ldc + n] = v;
    }
}

extern "C" int ddnm_bgemm_f32(const ddnm_gemm_desc* d, void* stream) {
    if (!d || !d->A || !d->Bm || !d->C) return DDNM_E_BADARG;
    if (d->M <= 0 || d->N <= 0 || d->K <= 0 || d->batch <= 0 || d->inner <= 0) return DDNM_E_BADARG;
    if (d->batch % d->inner) return DDNM_E_BADARG;
    hipStream_t s = (hipStream_t)stream;
    const bool aligned = ((d->lda | d->ldb) & 3) == 0 && ((d->sAo | d->sAi | d->sBo | d->sBi) & 3) == 0 &&
                         (((uintptr_t)d->A | (uintptr_t)d->Bm) & 15) == 0;
    if (d->M % 64 || d->N % 64 || d->K % KC || !aligned) {
        const int64_t total = (int64_t)d->M * d->N * d->batch;
        const unsigned grid = (unsigned)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
        DDNM_LAUNCH(bgemm_naive_kernel, dim3(grid), dim3(256), 0, s, *d);
        return 0;
    }
    const bool big = (d->M % 128 == 0) && (d->N % 128 == 0) &&
                     ((long)d->batch * (d->M / 128) * (d->N / 128) >= 256);
    if (big) {
        const int mt = d->M / 128, nt = d->N / 128;
        DDNM_LAUNCH((bgemm_f32_kernel<2, 2>), dim3(d->batch * mt * nt), dim3(256), 0, s, *d, mt, nt);
    } else {
        const int mt = d->M / 64, nt = d->N / 64;
        DDNM_LAUNCH((bgemm_f32_kernel<1, 1>), dim3(d->batch * mt * nt), dim3(256), 0, s, *d, mt, nt);
    }
    return 0;
}

// ---- row softmax: one wave per row, row length n <= 64*32
__global__ __launch_bounds__(256) void softmax_rows_kernel(float* x, int64_t rows, int n, int ld, float scale) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    float* p = x + row * ld;
    constexpr int MAXE = 32;
    float v[MAXE];
    float mx = -INFINITY;
#pragma unroll
    for (int i = 0; i < MAXE; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < n ? p[c] * scale : -INFINITY;
        mx = fmaxf(mx, v[i]);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
    float sum = 0.f;
#pragma unroll
    for (int i = 0; i < MAXE; ++i) {
        const int c = lane + 64 * i;
        v[i] = c < n ? expf(v[i] - mx) : 0.f;
        sum += v[i];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
    const float inv = 1.0f / sum;
#pragma unroll
    for (int i = 0; i < MAXE; ++i) {
        const int c = lane + 64 * i;
        if (c < n) p[c] = v[i] * inv;
    }
}

extern "C" int ddnm_softmax_rows_f32(float* x, int64_t rows, int32_t n, int32_t ld, float scale, void* stream) {
    if (!x || rows <= 0 || n <= 0) return DDNM_E_BADARG;
    if (n > 64 * 32) return DDNM_E_SHAPE;
    DDNM_LAUNCH(softmax_rows_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, (hipStream_t)stream, x,
                       rows, n, ld, scale);
    return 0;
}
