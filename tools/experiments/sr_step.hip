// The headline workload's sampler step in TWO launches (round 5; it was six: x0 kernel, four batched GEMMs, combine).
//
// sr_bicubic (BASELINE configs[1]): A = Ae (x) Ae acting on every (b, c) plane as Y = Ae X Ae^T with Ae [M x D] = U S V^T of the
// 1-D strided bicubic matrix and A^+ = Pe (x) Pe, Pe [D x M] (functions/svd_operators.py:851-931; host SVD in
// ddnm_amd/functions/svd_operators.py::SRConv).  One reverse step (functions/svd_ddnm.py:57-65):
//   X0  = (x_t - e_t sqrt(1 - abar_t)) / sqrt(abar_t)
//   R   = Ae X0 Ae^T - Y                                   [M x M]
//   X0h = X0 - Pe R Pe^T
//   x_{t-1} = sqrt(abar') X0h + c1 noise + c2 e_t
// Kernel A, grid (planes, D / 64): X0 for a 64-column block (written out: time travel and the caller need it), the block's
//   T1 = Ae X0[:, blk] and its contribution T1 Ae[:, blk]^T to R -- four partial [M x M] matrices per plane, summed in a
//   FIXED order by kernel B (no atomics: repeated steps are bit-identical).
// Kernel B, grid (planes, D / 64): R, T2 = Pe[rows] R, P = T2 Pe^T for a 64-row block, and the DDIM update of those rows
//   with the noise read from a tensor or drawn in-kernel (philox.h).
// fp32 FMA on the vector ALU: 21 MFLOP per plane, 0.5 GFLOP per step at B = 8 -- a launch-latency problem, not an arithmetic one.
// D = 256, M = 64 (256 x 256 images, 4x): other sizes keep the GEMM route.
#include "common.h"
#include "philox.h"

constexpr int SR_D = 256, SR_M = 64, SR_BLK = 64;

// the elementwise parts round like the other step kernels (ddnm_step.hip: mul / add separately, like the ATen kernels the
// reference runs); only the four small matrix products below may contract into FMAs
__device__ __forceinline__ f32x4 sr_x0_of(f32x4 xt, f32x4 et, const ddnm_step_scalars& s) {
#pragma clang fp contract(off)
    return (xt - et * s.sqrt_1m_at) / s.sqrt_at;
}
__device__ __forceinline__ f32x4 sr_update_of(f32x4 x0, f32x4 proj, f32x4 nz, f32x4 et, const ddnm_step_scalars& s) {
#pragma clang fp contract(off)
    const f32x4 x0h = x0 - proj * s.lambda;
    return (x0h * s.sqrt_at_next + nz * s.c1) + et * s.c2;
}

struct SrNoise {
    const float* p;
    PhiloxKey key;
    unsigned iter, img_base;
};

__global__ __launch_bounds__(256) void sr_step_a_kernel(const float* __restrict__ xt, const float* __restrict__ et,
                                                        int64_t et_bstride, const float* __restrict__ AeT_g,
                                                        float* __restrict__ x0, float* __restrict__ ws, int C,
                                                        ddnm_step_scalars s) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* AeT = smem;                          // [D][M]   AeT[k][i] = Ae[i][k]
    float* Xb = smem + SR_D * SR_M;             // [D][BLK] X0[k][j0 + j];  later T1T [BLK][M]: T1T[j][i]
    const int plane = blockIdx.x, jb = blockIdx.y, j0 = jb * SR_BLK;
    const int b = plane / C, c = plane - b * C;
    const int tid = threadIdx.x;
    // Ae^T [D][M] (transposed once on the host, in the operator's constructor): straight copy, no bank conflicts
    for (int e = tid; e < SR_M * SR_D / 4; e += 256)
        *reinterpret_cast<f32x4*>(AeT + e * 4) = *reinterpret_cast<const f32x4*>(AeT_g + e * 4);
    // X0 block: rows k = 0..255, 64 columns -> 16 float4 per row
    const float* xp = xt + (size_t)plane * SR_D * SR_D;
    const float* ep = et + (size_t)b * et_bstride + (size_t)c * SR_D * SR_D;
    float* x0p = x0 + (size_t)plane * SR_D * SR_D;
    for (int e = tid; e < SR_D * (SR_BLK / 4); e += 256) {
        const int k = e / (SR_BLK / 4), q = e - k * (SR_BLK / 4);
        const size_t off = (size_t)k * SR_D + j0 + q * 4;
        const f32x4 v = sr_x0_of(*reinterpret_cast<const f32x4*>(xp + off), *reinterpret_cast<const f32x4*>(ep + off), s);
        *reinterpret_cast<f32x4*>(x0p + off) = v;
        *reinterpret_cast<f32x4*>(Xb + k * SR_BLK + q * 4) = v;
    }
    __syncthreads();
    // T1[i][j] = sum_k Ae[i][k] X0[k][j0 + j]; thread -> 4 x 4 tile (rows 4 ti, cols 4 tj)
    const int ti = tid >> 4, tj = tid & 15;
    f32x4 acc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int k = 0; k < SR_D; ++k) {
        const f32x4 a = *reinterpret_cast<const f32x4*>(AeT + k * SR_M + ti * 4);
        const f32x4 x = *reinterpret_cast<const f32x4*>(Xb + k * SR_BLK + tj * 4);
        acc[0] += x * a.x; acc[1] += x * a.y; acc[2] += x * a.z; acc[3] += x * a.w;
    }
    __syncthreads();                             // everybody is done with the X0 block: it becomes T1T[j][i]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        Xb[(tj * 4 + 0) * SR_M + ti * 4 + r] = acc[r].x;
        Xb[(tj * 4 + 1) * SR_M + ti * 4 + r] = acc[r].y;
        Xb[(tj * 4 + 2) * SR_M + ti * 4 + r] = acc[r].z;
        Xb[(tj * 4 + 3) * SR_M + ti * 4 + r] = acc[r].w;
    }
    __syncthreads();
    // Rpart[i][m] = sum_j T1[i][j] Ae[m][j0 + j] = sum_j T1T[j][i] AeT[j0 + j][m]; thread -> rows 4 ti, cols 4 tj
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int j = 0; j < SR_BLK; ++j) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(Xb + j * SR_M + ti * 4);
        const f32x4 a = *reinterpret_cast<const f32x4*>(AeT + (j0 + j) * SR_M + tj * 4);
        acc[0] += a * t.x; acc[1] += a * t.y; acc[2] += a * t.z; acc[3] += a * t.w;
    }
    float* wp = ws + ((size_t)plane * (SR_D / SR_BLK) + jb) * SR_M * SR_M;
#pragma unroll
    for (int r = 0; r < 4; ++r) *reinterpret_cast<f32x4*>(wp + (ti * 4 + r) * SR_M + tj * 4) = acc[r];
}

__global__ __launch_bounds__(256) void sr_step_b_kernel(const float* __restrict__ x0, const float* __restrict__ et,
                                                        int64_t et_bstride, SrNoise noise, const float* __restrict__ y,
                                                        const float* __restrict__ PeT_g, const float* __restrict__ ws,
                                                        float* __restrict__ xt_next, int C, ddnm_step_scalars s) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* PeT = smem;                           // [M][D]    PeT[m][c] = Pe[c][m]  (its columns r0 .. r0 + 63 are Pe[rows]^T)
    float* R = PeT + SR_M * SR_D;                // [M][M]
    float* T2T = R + SR_M * SR_M;                // [M][BLK]  T2T[m][r]
    const int plane = blockIdx.x, rb = blockIdx.y, r0 = rb * SR_BLK;
    const int b = plane / C, c = plane - b * C;
    const int tid = threadIdx.x;
    for (int e = tid; e < SR_D * SR_M / 4; e += 256)
        *reinterpret_cast<f32x4*>(PeT + e * 4) = *reinterpret_cast<const f32x4*>(PeT_g + e * 4);
    const float* wp = ws + (size_t)plane * (SR_D / SR_BLK) * SR_M * SR_M;
    const float* yp = y + (size_t)plane * SR_M * SR_M;
    for (int e = tid; e < SR_M * SR_M / 4; e += 256) {
        f32x4 v = *reinterpret_cast<const f32x4*>(wp + e * 4);
#pragma unroll
        for (int q = 1; q < SR_D / SR_BLK; ++q) v += *reinterpret_cast<const f32x4*>(wp + q * SR_M * SR_M + e * 4);   // fixed order
        *reinterpret_cast<f32x4*>(R + e * 4) = v - *reinterpret_cast<const f32x4*>(yp + e * 4);
    }
    __syncthreads();
    // T2[r][m] = sum_i Pe[r0 + r][i] R[i][m]; thread -> rows 4 ti, cols 4 tj
    const int ti = tid >> 4, tj = tid & 15;
    f32x4 acc[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) acc[r] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int i = 0; i < SR_M; ++i) {
        const f32x4 p = *reinterpret_cast<const f32x4*>(PeT + i * SR_D + r0 + ti * 4);       // Pe[r0 + 4 ti ..][i]
        const f32x4 rr = *reinterpret_cast<const f32x4*>(R + i * SR_M + tj * 4);
        acc[0] += rr * p.x; acc[1] += rr * p.y; acc[2] += rr * p.z; acc[3] += rr * p.w;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        T2T[(tj * 4 + 0) * SR_BLK + ti * 4 + r] = acc[r].x;
        T2T[(tj * 4 + 1) * SR_BLK + ti * 4 + r] = acc[r].y;
        T2T[(tj * 4 + 2) * SR_BLK + ti * 4 + r] = acc[r].z;
        T2T[(tj * 4 + 3) * SR_BLK + ti * 4 + r] = acc[r].w;
    }
    __syncthreads();
    // P[r][cc] = sum_m T2[r][m] Pe[cc][m]; thread -> rows 4 ti, columns 16 tj .. 16 tj + 15 (four float4)
    f32x4 pa[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) pa[r][q] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int m = 0; m < SR_M; ++m) {
        const f32x4 t = *reinterpret_cast<const f32x4*>(T2T + m * SR_BLK + ti * 4);
        f32x4 pv[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) pv[q] = *reinterpret_cast<const f32x4*>(PeT + m * SR_D + tj * 16 + q * 4);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            pa[0][q] += pv[q] * t.x; pa[1][q] += pv[q] * t.y; pa[2][q] += pv[q] * t.z; pa[3][q] += pv[q] * t.w;
        }
    }
    const float* x0p = x0 + (size_t)plane * SR_D * SR_D;
    const float* ep = et + (size_t)b * et_bstride + (size_t)c * SR_D * SR_D;
    float* op = xt_next + (size_t)plane * SR_D * SR_D;
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const size_t off = (size_t)(r0 + ti * 4 + r) * SR_D + tj * 16 + q * 4;
            f32x4 nz;
            if (noise.p) {
                nz = *reinterpret_cast<const f32x4*>(noise.p + (size_t)plane * SR_D * SR_D + off);
            } else {
                const size_t in_img = (size_t)c * SR_D * SR_D + off;          // element offset inside image b
                nz = philox_normal4(noise.key, (unsigned)(in_img >> 2), noise.iter, noise.img_base + (unsigned)b);
            }
            *reinterpret_cast<f32x4*>(op + off) = sr_update_of(*reinterpret_cast<const f32x4*>(x0p + off), pa[r][q], nz,
                                                               *reinterpret_cast<const f32x4*>(ep + off), s);
        }
}

extern "C" int64_t ddnm_step_srconv_workspace_floats(int32_t B, int32_t C, int32_t D, int32_t M) {
    if (B <= 0 || C <= 0 || D != SR_D || M != SR_M) return DDNM_E_SHAPE;
    return (int64_t)B * C * (SR_D / SR_BLK) * SR_M * SR_M;
}

extern "C" int ddnm_step_srconv_f32(const float* xt, const float* et, int64_t et_bstride, const float* noise, const float* y,
                                    const float* AeT, const float* PeT, float* workspace, float* x0, float* xt_next, int32_t B,
                                    int32_t C, int32_t D, int32_t M, const ddnm_step_scalars* s, void* stream) {
    if (!xt || !et || !y || !AeT || !PeT || !workspace || !x0 || !xt_next || !s || B <= 0 || C <= 0) return DDNM_E_BADARG;
    if (!noise && !s->rng_on) return DDNM_E_BADARG;
    if (D != SR_D || M != SR_M || (et_bstride & 3)) return DDNM_E_SHAPE;
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(B * C, SR_D / SR_BLK);
    const size_t lds_a = (size_t)(SR_D * SR_M + SR_D * SR_BLK) * sizeof(float);                       // 128 KB
    const size_t lds_b = (size_t)(SR_M * SR_D + SR_M * SR_M + SR_M * SR_BLK) * sizeof(float);          // 96 KB
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sr_step_a_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_a);
        (void)hipFuncSetAttribute(reinterpret_cast<const void*>(sr_step_b_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_b);
        attr_set = true;
    }
    DDNM_LAUNCH(sr_step_a_kernel, grid, dim3(256), lds_a, st, xt, et, et_bstride, AeT, x0, workspace, C, *s);
    const SrNoise nz{noise, PhiloxKey{s->rng_seed_lo, s->rng_seed_hi}, s->rng_iter, s->rng_image_base};
    DDNM_LAUNCH(sr_step_b_kernel, grid, dim3(256), lds_b, st, x0, et, et_bstride, nz, y, PeT, workspace, xt_next, C, *s);
    return 0;
}
