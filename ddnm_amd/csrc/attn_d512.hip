// Fused single-head self-attention of the celeba `Model`'s AttnBlock (/root/reference/guided_diffusion/models.py:137-189,
// SURVEY.md K5 `attn_fwd_d512`), gfx950:   w = softmax_j(q_i . k_j * C^-0.5),  o_i = sum_j w_ij v_j   over T = H*W tokens of
// one image, ONE head of C channels (512 in the shipped net).  Until round 5 this ran as bgemm -> softmax_rows -> bgemm with
// the [T][T] scores in HBM (3 launches per block).  Here one workgroup owns 32 queries of one image:
//
//   phase 1  S[32][T] = Q K^T (wave w takes key tiles w, w + 4, ..): both operands straight from global memory in MFMA
//            fragment order (a lane holds 8 consecutive channels of its query / key row = 32 contiguous bytes), scores to LDS;
//   phase 2  row softmax in fp32, exactly the two-pass form of torch.softmax (max, exp, sum, divide), P kept in LDS as
//            hi | lo fp16 halves;
//   phase 3  O[32][C] = P V (wave w takes channel blocks w, w + 4, ..), V from global memory (a lane holds 8 consecutive KEYS
//            of its channel: per key a coalesced 128-byte row segment).
//
// Arithmetic = the split-fp16 form of the convolutions (csrc/conv_common.h::split_store): every fp32 operand value is carried
// as hi = rn16(v), lo = rn16(v - hi), a product is hi*hi' + hi*lo' + lo*hi' on v_mfma_f32_32x32x16_f16 with fp32 accumulation
// (operand error 2^-22).  q, k, v are RAW convolution outputs, so they are multiplied by powers of two the host derives ONCE
// per checkpoint from a static bound of |W . GN(x) + b| (ops.attn_operand_scales; the scores / outputs are multiplied by the
// inverse powers: exact); probabilities are carried as 2^14 p.
//
// Measured (one MI355X, B = 8, C = 512; tools/r06/attn_time.py): T = 64 (the 8 x 8 mid block) 26.0 -> 17.4 us against the three
// launches; T = 256 (the 16 x 16 blocks) 42.1 us for the three launches against 50.7 us here -- 64 workgroups that each pull
// ~1 MB of q / k / v from L2 in fragment order do not beat three chip-filling launches whose 2 MB score tensor never leaves
// L2.  The model therefore uses this kernel for T <= 64 and keeps the three-launch route at T = 256 (models.Model._attn).
#include "conv_common.h"

namespace {
constexpr int AQ = 32;                 // queries per workgroup
constexpr int A_TMAX = 256;            // tokens per image the LDS score tile is sized for (16 x 16 level; 8 x 8: 64)

__device__ __forceinline__ void split8(const f32x4 a, const f32x4 b, const float s, half8& hi, half8& lo) {
    const float v[8] = {a.x * s, a.y * s, a.z * s, a.w * s, b.x * s, b.y * s, b.z * s, b.w * s};
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        hi[i] = (_Float16)v[i];
        lo[i] = (_Float16)(v[i] - (float)hi[i]);
    }
}
}  // namespace

// NB = 32-channel output blocks per wave (C = 128 NB): a wave multiplies all of them per key step, so the probabilities are
// read from LDS once and 8 NB value loads are in flight together; phase 1 likewise keeps two key tiles per wave going on one
// set of query fragments.
template <int NB>
__global__ __launch_bounds__(256) void attn_d512_kernel(const float* __restrict__ qkv, float* __restrict__ out, int T,
                                                        float s_qk, float s_v, float sm_scale) {
    constexpr int C = 128 * NB;
    __shared__ __attribute__((aligned(16))) float S[AQ * (A_TMAX + 4)];                 // scores, row pitch T + 4
    __shared__ __attribute__((aligned(16))) _Float16 P[2 * AQ * (A_TMAX + 8)];          // probabilities: hi rows, then lo rows
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int qblocks = T / AQ;
    const int b = blockIdx.x / qblocks, q0 = (blockIdx.x - b * qblocks) * AQ;
    constexpr int ld = 3 * C;
    const float* base = qkv + (size_t)b * T * ld;
    const int r31 = lane & 31, kh = lane >> 5;
    const int SP = T + 4, PP = T + 8;
    const int nkt = T / 32;

    // ---- phase 1: scores.  A = Q (rows = queries), B = K (columns = keys); k-step = 16 channels; key tiles wave, wave + 4
    const float* qrow = base + (size_t)(q0 + r31) * ld + kh * 8;
    for (int kt = wave; kt < nkt; kt += 8) {
        const bool two = kt + 4 < nkt;                                   // wave-uniform
        const float* k0row = base + (size_t)(kt * 32 + r31) * ld + C + kh * 8;
        const float* k1row = base + (size_t)((two ? kt + 4 : kt) * 32 + r31) * ld + C + kh * 8;
        f32x16 acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc0[r] = acc1[r] = 0.f;
#pragma unroll 4
        for (int c = 0; c < C; c += 16) {
            const f32x4 qa = *reinterpret_cast<const f32x4*>(qrow + c), qb = *reinterpret_cast<const f32x4*>(qrow + c + 4);
            const f32x4 ka = *reinterpret_cast<const f32x4*>(k0row + c), kb = *reinterpret_cast<const f32x4*>(k0row + c + 4);
            const f32x4 la = *reinterpret_cast<const f32x4*>(k1row + c), lb = *reinterpret_cast<const f32x4*>(k1row + c + 4);
            half8 qh, ql, khh, kl;
            split8(qa, qb, s_qk, qh, ql);
            split8(ka, kb, s_qk, khh, kl);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ql, khh, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(qh, kl, acc0, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(qh, khh, acc0, 0, 0, 0);
            if (two) {
                split8(la, lb, s_qk, khh, kl);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ql, khh, acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(qh, kl, acc1, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(qh, khh, acc1, 0, 0, 0);
            }
        }
        const float un = sm_scale / (s_qk * s_qk);           // undo the operand scaling (powers of two) and apply C^-0.5
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * kh;
            S[m * SP + kt * 32 + r31] = acc0[r] * un;
            if (two) S[m * SP + (kt + 4) * 32 + r31] = acc1[r] * un;
        }
    }
    __syncthreads();
    // ---- phase 2: softmax over the keys; 8 threads per query row
    {
        const int row = tid >> 3, sub = tid & 7;
        float* s = S + row * SP;
        float mx = -3.0e38f;
        for (int j = sub; j < T; j += 8) mx = fmaxf(mx, s[j]);
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) mx = fmaxf(mx, __shfl_xor(mx, o));
        float sum = 0.f;
        for (int j = sub; j < T; j += 8) {
            const float e = expf(s[j] - mx);
            s[j] = e;
            sum += e;
        }
#pragma unroll
        for (int o = 1; o < 8; o <<= 1) sum += __shfl_xor(sum, o);
        const float inv = 1.0f / sum;
        _Float16* ph = P + row * PP;
        _Float16* pl = P + (AQ + row) * PP;
        for (int j = sub; j < T; j += 8) {
            const float p = s[j] * inv * 16384.0f;      // 2^14: the lo half of p ~ 1 / T stays a NORMAL fp16 number
            const _Float16 h = (_Float16)p;
            ph[j] = h;
            pl[j] = (_Float16)(p - (float)h);
        }
    }
    __syncthreads();
    // ---- phase 3: O = P V.  A = P (rows = queries, k = keys) from LDS, B = V (k = keys, columns = channels) from global;
    // channel blocks wave + 4 i, i < NB, together
    const float* vbase = base + 2 * C + wave * 32 + r31;
    const float inv_sv = 1.0f / (s_v * 16384.0f);
    f32x16 acc[NB];
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll 2
    for (int k0 = 0; k0 < T; k0 += 16) {
        const half8 ph = *reinterpret_cast<const half8*>(P + r31 * PP + k0 + kh * 8);
        const half8 pl = *reinterpret_cast<const half8*>(P + (AQ + r31) * PP + k0 + kh * 8);
        float v[NB][8];
#pragma unroll
        for (int i = 0; i < NB; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) v[i][j] = vbase[(size_t)(k0 + kh * 8 + j) * ld + i * 128];
#pragma unroll
        for (int i = 0; i < NB; ++i) {
            half8 vh, vl;
            split8(f32x4{v[i][0], v[i][1], v[i][2], v[i][3]}, f32x4{v[i][4], v[i][5], v[i][6], v[i][7]}, s_v, vh, vl);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(pl, vh, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, vl, acc[i], 0, 0, 0);
            acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ph, vh, acc[i], 0, 0, 0);
        }
    }
#pragma unroll
    for (int i = 0; i < NB; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * kh;
            out[((size_t)b * T + q0 + m) * C + wave * 32 + i * 128 + r31] = acc[i][r] * inv_sv;
        }
}

// qkv = [B][T][3C] fp32 (q | k | v per token: the fused q / k / v 1x1 convolution's output), out = [B][T][C] fp32.
// s_qk / s_v = powers of two that bring max|q|, max|k| resp. max|v| below 2^15 (host: static bound per checkpoint).
extern "C" int ddnm_attn_fused_f32(const float* qkv, float* out, int32_t B, int32_t T, int32_t C, float s_qk, float s_v,
                                   float softmax_scale, void* stream) {
    if (!qkv || !out || B <= 0 || !(s_qk > 0.f) || !(s_v > 0.f)) return DDNM_E_BADARG;
    if (T <= 0 || T % AQ || T > A_TMAX || C <= 0 || C % 128 || C > 512) return DDNM_E_SHAPE;
    const dim3 grid(B * (T / AQ));
    hipStream_t s = (hipStream_t)stream;
    switch (C / 128) {
        case 1: DDNM_LAUNCH(attn_d512_kernel<1>, grid, dim3(256), 0, s, qkv, out, T, s_qk, s_v, softmax_scale); break;
        case 2: DDNM_LAUNCH(attn_d512_kernel<2>, grid, dim3(256), 0, s, qkv, out, T, s_qk, s_v, softmax_scale); break;
        case 3: DDNM_LAUNCH(attn_d512_kernel<3>, grid, dim3(256), 0, s, qkv, out, T, s_qk, s_v, softmax_scale); break;
        default: DDNM_LAUNCH(attn_d512_kernel<4>, grid, dim3(256), 0, s, qkv, out, T, s_qk, s_v, softmax_scale);
    }
    return 0;
}

extern "C" int ddnm_attn_fused_supported(int32_t T, int32_t C) { return (T > 0 && T % AQ == 0 && T <= A_TMAX && C > 0 && C % 128 == 0 && C <= 512) ? 1 : 0; }
