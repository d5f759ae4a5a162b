// Fused backward of the 64-channel-head self-attention (ddnm_attn16_d64, attn16.hip) for the classifier-guidance
// gradient: the reference differentiates QKVAttentionLegacy (guided_diffusion/unet.py:328-354) with torch autograd
// (guided_diffusion/diffusion.py:183-189), materialising the [T][T] probabilities and their gradient per head.  Here
// nothing of size [T][T] touches HBM: the probabilities are recomputed from the forward's per-query log-sum-exp,
//
//   S = q k^T / 8,  P = exp2(S * log2 e - lse),  dP = dO v^T,  D_q = sum_d dO[q,d] O[q,d]  (= rowsum(dP . P)),
//   dS = P . (dP - D),  dq = dS k / 8,  dk = dS^T q / 8,  dv = P^T dO,
//
// in two kernels that mirror the forward's tiling (a lane owns ONE column entity, 64 row entities stream through LDS):
//   * attn16_bwd_dq_kernel:  column = query (fragments of q and dO in registers), rows = keys (K, V tiles in LDS);
//                            also emits D_q for the second kernel;
//   * attn16_bwd_dkv_kernel: column = key (fragments of k and v in registers), rows = queries (Q, dO tiles in LDS).
// MFMA layouts exactly as in attn16.hip: the first products give S^T / dP^T (rows = streamed entity, col = owned
// entity) in the accumulator layout; rounded to fp16 they ARE the B operand of the second products, whose A operand is
// the transposed tile stored in LDS in the k-order that layout dictates.
//
//   qkv, dqkv fp16 [B][T][3C]  channel = head*192 + {q: 0..63, k: 64..127, v: 128..191}
//   o, dO     fp16 [B][T][C]   channel = head*64 + d
//   lse, dsum fp32 [B][C/64][T]
#include "common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

constexpr int AB_KT = 64;            // streamed rows per tile

// row-major tile image: row r (128 bytes = 64 halfs), 16-byte piece index XOR-swizzled with (r >> 1) & 7
__device__ __forceinline__ void stage_rows(char* S, int r, int pc, uint4 v) {
    *reinterpret_cast<uint4*>(S + r * 128 + ((pc ^ ((r >> 1) & 7)) << 4)) = v;
}
// transposed tile image: row = d (0..63), column position of streamed row r permuted inside its group of 16 to
// [0-3, 8-11, 4-7, 12-15] (the 8 rows one lane-half contributes to an MFMA k-step form one 16-byte piece)
__device__ __forceinline__ void stage_transposed(char* St, int r, int pc, uint4 v) {
    const int r16 = r & 15;
    const int pos = (r & ~15) | (r16 & 3) | ((r16 & 8) >> 1) | ((r16 & 4) << 1);
    const half8 h = __builtin_bit_cast(half8, v);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int d = pc * 8 + e;
        *reinterpret_cast<_Float16*>(St + d * 128 + (((pos >> 3) ^ ((d >> 1) & 7)) << 4) + (pos & 7) * 2) = h[e];
    }
}
__device__ __forceinline__ half8 frag_rows(const char* S, int r, int piece) {
    return *reinterpret_cast<const half8*>(S + r * 128 + ((piece ^ ((r >> 1) & 7)) << 4));
}

// ------------------------------------------------------------------------------------------------
// dq (and D).  One workgroup = NW waves = 32*NW queries of one (batch, head).
// ------------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(NW * 64) void attn16_bwd_dq_kernel(const _Float16* __restrict__ qkv, const _Float16* __restrict__ o,
                                                                const _Float16* __restrict__ dO, const float* __restrict__ lse,
                                                                float* __restrict__ dsum, _Float16* __restrict__ dqkv, int T,
                                                                int C, float scale_log2, float scale) {
    constexpr int NT = NW * 64;
    constexpr int PPT = AB_KT * 8 / NT;               // 16-byte pieces per thread per 64 x 64 tile
    __shared__ __attribute__((aligned(16))) char Ks[AB_KT * 128];
    __shared__ __attribute__((aligned(16))) char Vs[AB_KT * 128];
    __shared__ __attribute__((aligned(16))) char Kt[64 * 128];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kh = lane >> 5, ql = lane & 31;
    const int head = blockIdx.y, b = blockIdx.z, nh = gridDim.y;
    const int q0 = blockIdx.x * (32 * NW) + wave * 32;
    const size_t row3 = (size_t)3 * C;
    const _Float16* base = qkv + (size_t)b * T * row3 + (size_t)head * 192;
    const _Float16* obase = o + (size_t)b * T * C + (size_t)head * 64;
    const _Float16* dobase = dO + (size_t)b * T * C + (size_t)head * 64;

    // fragments of this lane's query (B operands, cols = queries): d = 16*ks + 8*kh .. +7
    half8 qf[4], dof[4];
    float dq_sum = 0.f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        qf[ks] = *reinterpret_cast<const half8*>(base + (size_t)(q0 + ql) * row3 + ks * 16 + kh * 8);
        dof[ks] = *reinterpret_cast<const half8*>(dobase + (size_t)(q0 + ql) * C + ks * 16 + kh * 8);
        const half8 of = *reinterpret_cast<const half8*>(obase + (size_t)(q0 + ql) * C + ks * 16 + kh * 8);
#pragma unroll
        for (int e = 0; e < 8; ++e) dq_sum += (float)dof[ks][e] * (float)of[e];
    }
    dq_sum += __shfl_xor(dq_sum, 32);                 // D_q = sum over all 64 d
    const size_t srow = ((size_t)b * nh + head) * T + q0 + ql;
    if (kh == 0) dsum[srow] = dq_sum;
    const float lse_q = lse[srow];

    f32x16 acc[2];                 // dq^T: rows d = dt*32 + (r&3) + 8*(r>>2) + 4*kh, col = query
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[dt][r] = 0.f;

    const int nkt = T / AB_KT;
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();                              // previous tile's fragment reads are done
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int pi = tid + i * NT;
            const int key = pi >> 3, pc = pi & 7;
            const _Float16* r = base + (size_t)(kt * AB_KT + key) * row3 + pc * 8;
            const uint4 kr = *reinterpret_cast<const uint4*>(r + 64);
            const uint4 vr = *reinterpret_cast<const uint4*>(r + 128);
            stage_rows(Ks, key, pc, kr);
            stage_rows(Vs, key, pc, vr);
            stage_transposed(Kt, key, pc, kr);
        }
        __syncthreads();
        half8 dsf[2][2];                              // [key tile j][k-step t]: dS^T rounded to fp16 = B operand
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
            const int key = j * 32 + ql;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag_rows(Ks, key, ks * 2 + kh), qf[ks], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag_rows(Vs, key, ks * 2 + kh), dof[ks], dp, 0, 0, 0);
            }
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int r = t * 8 + e;
                    const float p = __builtin_amdgcn_exp2f(s[r] * scale_log2 - lse_q);
                    dsf[j][t][e] = (_Float16)(p * (dp[r] - dq_sum));
                }
        }
        // dq^T += K^T dS^T
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const int d = dt * 32 + ql;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int t = 0; t < 2; ++t)
                    acc[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag_rows(Kt, d, j * 4 + t * 2 + kh), dsf[j][t], acc[dt], 0, 0, 0);
        }
    }
    _Float16* op = dqkv + ((size_t)b * T + q0 + ql) * row3 + (size_t)head * 192;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            half4 h = {(_Float16)(acc[dt][4 * g] * scale), (_Float16)(acc[dt][4 * g + 1] * scale),
                       (_Float16)(acc[dt][4 * g + 2] * scale), (_Float16)(acc[dt][4 * g + 3] * scale)};
            *reinterpret_cast<half4*>(op + dt * 32 + 8 * g + 4 * kh) = h;
        }
}

// ------------------------------------------------------------------------------------------------
// dk, dv.  One workgroup = NW waves = 32*NW keys of one (batch, head); query tiles stream through LDS.
// ------------------------------------------------------------------------------------------------
template <int NW>
__global__ __launch_bounds__(NW * 64) void attn16_bwd_dkv_kernel(const _Float16* __restrict__ qkv, const _Float16* __restrict__ dO,
                                                                 const float* __restrict__ lse, const float* __restrict__ dsum,
                                                                 _Float16* __restrict__ dqkv, int T, int C, float scale_log2,
                                                                 float scale) {
    constexpr int NT = NW * 64;
    constexpr int PPT = AB_KT * 8 / NT;
    __shared__ __attribute__((aligned(16))) char Qs[AB_KT * 128];
    __shared__ __attribute__((aligned(16))) char Gs[AB_KT * 128];      // dO tile, rows = queries
    __shared__ __attribute__((aligned(16))) char Qt[64 * 128];
    __shared__ __attribute__((aligned(16))) char Gt[64 * 128];
    __shared__ __attribute__((aligned(16))) float Ls[AB_KT];
    __shared__ __attribute__((aligned(16))) float Ds[AB_KT];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kh = lane >> 5, ql = lane & 31;
    const int head = blockIdx.y, b = blockIdx.z, nh = gridDim.y;
    const int k0 = blockIdx.x * (32 * NW) + wave * 32;
    const size_t row3 = (size_t)3 * C;
    const _Float16* base = qkv + (size_t)b * T * row3 + (size_t)head * 192;
    const _Float16* dobase = dO + (size_t)b * T * C + (size_t)head * 64;
    const float* lrow = lse + ((size_t)b * nh + head) * T;
    const float* drow = dsum + ((size_t)b * nh + head) * T;

    half8 kf[4], vf[4];            // fragments of this lane's key (B operands, cols = keys)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        kf[ks] = *reinterpret_cast<const half8*>(base + (size_t)(k0 + ql) * row3 + 64 + ks * 16 + kh * 8);
        vf[ks] = *reinterpret_cast<const half8*>(base + (size_t)(k0 + ql) * row3 + 128 + ks * 16 + kh * 8);
    }
    f32x16 ak[2], av[2];           // dk^T, dv^T: rows d, col = key
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) { ak[dt][r] = 0.f; av[dt][r] = 0.f; }

    const int nqt = T / AB_KT;
    for (int qt = 0; qt < nqt; ++qt) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < PPT; ++i) {
            const int pi = tid + i * NT;
            const int qr = pi >> 3, pc = pi & 7;
            const uint4 qv = *reinterpret_cast<const uint4*>(base + (size_t)(qt * AB_KT + qr) * row3 + pc * 8);
            const uint4 gv = *reinterpret_cast<const uint4*>(dobase + (size_t)(qt * AB_KT + qr) * C + pc * 8);
            stage_rows(Qs, qr, pc, qv);
            stage_rows(Gs, qr, pc, gv);
            stage_transposed(Qt, qr, pc, qv);
            stage_transposed(Gt, qr, pc, gv);
        }
        if (tid < AB_KT) {
            Ls[tid] = lrow[qt * AB_KT + tid];
            Ds[tid] = drow[qt * AB_KT + tid];
        }
        __syncthreads();
        half8 pf[2][2], dsf[2][2];                    // [query tile i][k-step t]
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            f32x16 s, dp;
#pragma unroll
            for (int r = 0; r < 16; ++r) { s[r] = 0.f; dp[r] = 0.f; }
            const int qr = i * 32 + ql;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                s = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag_rows(Qs, qr, ks * 2 + kh), kf[ks], s, 0, 0, 0);
                dp = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag_rows(Gs, qr, ks * 2 + kh), vf[ks], dp, 0, 0, 0);
            }
            // rows of the accumulator = queries i*32 + (r&3) + 8*(r>>2) + 4*kh: four consecutive ones per r>>2
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const f32x4 l4 = *reinterpret_cast<const f32x4*>(&Ls[i * 32 + 8 * g + 4 * kh]);
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(&Ds[i * 32 + 8 * g + 4 * kh]);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int r = 4 * g + e;
                    const float p = __builtin_amdgcn_exp2f(s[r] * scale_log2 - l4[e]);
                    pf[i][r >> 3][r & 7] = (_Float16)p;
                    dsf[i][r >> 3][r & 7] = (_Float16)(p * (dp[r] - d4[e]));
                }
            }
        }
        // dv^T += dO^T P,  dk^T += Q^T dS
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const int d = dt * 32 + ql;
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    av[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag_rows(Gt, d, i * 4 + t * 2 + kh), pf[i][t], av[dt], 0, 0, 0);
                    ak[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(frag_rows(Qt, d, i * 4 + t * 2 + kh), dsf[i][t], ak[dt], 0, 0, 0);
                }
        }
    }
    _Float16* op = dqkv + ((size_t)b * T + k0 + ql) * row3 + (size_t)head * 192;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            half4 hk = {(_Float16)(ak[dt][4 * g] * scale), (_Float16)(ak[dt][4 * g + 1] * scale),
                        (_Float16)(ak[dt][4 * g + 2] * scale), (_Float16)(ak[dt][4 * g + 3] * scale)};
            half4 hv = {(_Float16)av[dt][4 * g], (_Float16)av[dt][4 * g + 1], (_Float16)av[dt][4 * g + 2],
                        (_Float16)av[dt][4 * g + 3]};
            *reinterpret_cast<half4*>(op + 64 + dt * 32 + 8 * g + 4 * kh) = hk;
            *reinterpret_cast<half4*>(op + 128 + dt * 32 + 8 * g + 4 * kh) = hv;
        }
}

extern "C" int ddnm_attn16_d64_bwd(const void* qkv, const void* o, const void* dO, const float* lse, float* dsum,
                                   void* dqkv, int32_t B, int32_t T, int32_t C, void* stream) {
    if (!qkv || !o || !dO || !lse || !dsum || !dqkv || B <= 0 || T <= 0 || C <= 0) return DDNM_E_BADARG;
    if (C % 64 || T % 64) return DDNM_E_SHAPE;
    const int nh = C / 64;
    const float scale = 0.125f, scale_log2 = 0.125f * 1.4426950408889634f;
    hipStream_t s = (hipStream_t)stream;
    const _Float16* q = reinterpret_cast<const _Float16*>(qkv);
    const _Float16* oo = reinterpret_cast<const _Float16*>(o);
    const _Float16* g = reinterpret_cast<const _Float16*>(dO);
    _Float16* dq = reinterpret_cast<_Float16*>(dqkv);
    if (T % 128 == 0) {
        DDNM_LAUNCH((attn16_bwd_dq_kernel<4>), dim3(T / 128, nh, B), dim3(256), 0, s, q, oo, g, lse, dsum, dq, T, C, scale_log2, scale);
        DDNM_LAUNCH((attn16_bwd_dkv_kernel<4>), dim3(T / 128, nh, B), dim3(256), 0, s, q, g, lse, dsum, dq, T, C, scale_log2, scale);
    } else {
        DDNM_LAUNCH((attn16_bwd_dq_kernel<2>), dim3(T / 64, nh, B), dim3(128), 0, s, q, oo, g, lse, dsum, dq, T, C, scale_log2, scale);
        DDNM_LAUNCH((attn16_bwd_dkv_kernel<2>), dim3(T / 64, nh, B), dim3(128), 0, s, q, g, lse, dsum, dq, T, C, scale_log2, scale);
    }
    return 0;
}
