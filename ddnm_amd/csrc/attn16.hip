// Fused self-attention of the ADM UNet's AttentionBlock (guided_diffusion/unet.py:259-354, QKVAttentionLegacy
// :328-354) on fp16 MFMA, gfx950: softmax(q k^T / sqrt(d)) v per head with d = 64, no score tensor in HBM.
//
//   qkv  fp16 [B][T][3C]  channel = head*192 + {q: 0..63, k: 64..127, v: 128..191}   (legacy head-major order)
//   out  fp16 [B][T][C]   channel = head*64 + d
//
// One workgroup = NW waves = 32*NW queries of one (batch, head); key/value tiles of 64 keys stream through LDS.
// Everything is computed TRANSPOSED so that a lane owns one query:
//   S^T = K Q^T    A operand = K tile rows (keys) from LDS (XOR-swizzled 128-byte rows, ds_read_b128),
//                  B operand = the wave's Q fragments, loaded once from HBM into registers;
//                  D layout: lane -> query = lane & 31, 16 of the tile's 32 keys (the other 16 in lane ^ 32)
//   online softmax in fp32 registers: per-lane max / sum over its 32 keys + ONE cross-half exchange per tile;
//                  P is rounded to fp16 (`.type(weight.dtype)`, unet.py:352) and IS the B operand of the second
//                  product as it lies in the accumulator registers (the contraction order over keys is free, so the
//                  V^T image in LDS is permuted to the order the accumulator layout dictates -- no shuffles)
//   O^T = V^T P^T  A operand = V^T rows (d) from LDS: the V tile is transposed while it is written to LDS.
// The next tile's K / V rows are fetched into registers before the current tile's MFMAs (one tile of prefetch).
#include "common.h"

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef _Float16 half4 __attribute__((ext_vector_type(4)));

constexpr int AT_D = 64, AT_KT = 64;           // head dim, keys per tile

template <int NW>
__global__ __launch_bounds__(NW * 64) void attn16_d64_kernel(const _Float16* __restrict__ qkv, _Float16* __restrict__ out,
                                                             float* __restrict__ lse, int T, int C, float scale_log2) {
    constexpr int NT = NW * 64;
    constexpr int PIECES = AT_KT * 8;                 // 16-byte pieces per 64 x 64 fp16 tile
    constexpr int PPT = PIECES / NT;                  // pieces per thread per tile (2 for 4 waves, 4 for 2)
    __shared__ __attribute__((aligned(16))) char Ks[AT_KT * 128];
    __shared__ __attribute__((aligned(16))) char Vt[AT_D * 128];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kh = lane >> 5, ql = lane & 31;
    const int head = blockIdx.y, b = blockIdx.z;
    const int q0 = blockIdx.x * (32 * NW) + wave * 32;
    const size_t row3 = (size_t)3 * C;
    const _Float16* base = qkv + (size_t)b * T * row3 + (size_t)head * 192;

    // Q fragments (B operand, cols = queries): k-slots = d, 8 consecutive at 16*ks + 8*kh
    half8 qf[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks)
        qf[ks] = *reinterpret_cast<const half8*>(base + (size_t)(q0 + ql) * row3 + ks * 16 + kh * 8);

    // tile loader mapping: piece index pi -> (key = pi / 8, 16-byte piece = pi % 8)
    // prefetch registers as NAMED values: as arrays they were demoted to scratch memory, which put a wait for the
    // global loads + a scratch round trip in front of every tile (no prefetch at all)
    uint4 k0, k1, k2, k3, v0, v1, v2, v3;
    auto fetch1 = [&](int kt, int i, uint4& kr, uint4& vr) {
        const int pi = tid + i * NT;
        const int key = pi >> 3, pc = pi & 7;
        const _Float16* r = base + (size_t)(kt * AT_KT + key) * row3 + pc * 8;
        kr = *reinterpret_cast<const uint4*>(r + 64);
        vr = *reinterpret_cast<const uint4*>(r + 128);
    };
    auto fetch = [&](int kt) {
        fetch1(kt, 0, k0, v0);
        fetch1(kt, 1, k1, v1);
        if constexpr (PPT > 2) {
            fetch1(kt, 2, k2, v2);
            fetch1(kt, 3, k3, v3);
        }
    };
    auto stage1 = [&](int i, const uint4& kr, const uint4& vr) {
        const int pi = tid + i * NT;
        const int key = pi >> 3, pc = pi & 7;
        *reinterpret_cast<uint4*>(Ks + key * 128 + ((pc ^ ((key >> 1) & 7)) << 4)) = kr;
        // V^T: row = d, position of `key` inside its 16-key group permuted to [0-3, 8-11, 4-7, 12-15] so that the 8
        // keys one lane-half contributes to an MFMA k-step are one contiguous 16-byte piece
        const int k16 = key & 15;
        const int pos = (key & ~15) | (k16 & 3) | ((k16 & 8) >> 1) | ((k16 & 4) << 1);
        const half8 v = __builtin_bit_cast(half8, vr);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int d = pc * 8 + e;
            *reinterpret_cast<_Float16*>(Vt + d * 128 + (((pos >> 3) ^ ((d >> 1) & 7)) << 4) + (pos & 7) * 2) = v[e];
        }
    };
    auto stage = [&]() {
        stage1(0, k0, v0);
        stage1(1, k1, v1);
        if constexpr (PPT > 2) {
            stage1(2, k2, v2);
            stage1(3, k3, v3);
        }
    };

    f32x16 o[2];                  // O^T: rows d = dt*32 + (r&3) + 8*(r>>2) + 4*kh, col = query
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int r = 0; r < 16; ++r) o[dt][r] = 0.f;
    float m = -1e30f, l = 0.f;

    const int nkt = T / AT_KT;
    fetch(0);
    for (int kt = 0; kt < nkt; ++kt) {
        __syncthreads();                          // previous tile's fragment reads are done
        stage();
        __syncthreads();
        if (kt + 1 < nkt) fetch(kt + 1);

        // ---- S^T = K Q^T for 2 x 32 keys
        f32x16 s[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[j][r] = 0.f;
            const int key = j * 32 + ql;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const half8 kf = *reinterpret_cast<const half8*>(Ks + key * 128 + (((ks * 2 + kh) ^ ((key >> 1) & 7)) << 4));
                s[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], s[j], 0, 0, 0);
            }
        }
        // ---- online softmax (base-2 domain), per query = per lane (+ its partner lane ^ 32)
        float mx = s[0][0];
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[j][r]);
        mx = fmaxf(mx, __shfl_xor(mx, 32));
        const float m_new = fmaxf(m, mx * scale_log2);
        const float alpha = __builtin_amdgcn_exp2f(m - m_new);
        float psum = 0.f;
        half8 pf[2][2];                           // [key tile j][k-step t]: 8 fp16 probabilities = the B operand
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    // raw v_exp_f32 (libm's exp2f wraps it in 5 more instructions for results below 2^-126, which round to 0 in
                    // fp16 anyway): a third of this kernel's vector-ALU work
                    const _Float16 ph = (_Float16)__builtin_amdgcn_exp2f(s[j][t * 8 + e] * scale_log2 - m_new);
                    pf[j][t][e] = ph;
                    psum += (float)ph;            // the sum of what is actually multiplied into V (like the reference)
                }
        l = l * alpha + psum;
        m = m_new;
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
            for (int r = 0; r < 16; ++r) o[dt][r] *= alpha;
        // ---- O^T += V^T P^T
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const int d = dt * 32 + ql;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int piece = j * 4 + t * 2 + kh;
                    const half8 vf = *reinterpret_cast<const half8*>(Vt + d * 128 + ((piece ^ ((d >> 1) & 7)) << 4));
                    o[dt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[j][t], o[dt], 0, 0, 0);
                }
        }
    }
    l += __shfl_xor(l, 32);
    const float inv = 1.0f / l;
    // base-2 log-sum-exp of the query's scaled logits: the backward pass (attn16_bwd.hip) recomputes the probabilities
    // as exp2(s * scale_log2 - lse) instead of reading a [T][T] tensor
    if (lse && kh == 0) lse[((size_t)b * gridDim.y + head) * T + q0 + ql] = m + __log2f(l);
    _Float16* op = out + ((size_t)b * T + q0 + ql) * C + head * 64;
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            half4 h = {(_Float16)(o[dt][4 * g] * inv), (_Float16)(o[dt][4 * g + 1] * inv),
                       (_Float16)(o[dt][4 * g + 2] * inv), (_Float16)(o[dt][4 * g + 3] * inv)};
            *reinterpret_cast<half4*>(op + dt * 32 + 8 * g + 4 * kh) = h;
        }
}

extern "C" int ddnm_attn16_d64_lse(const void* qkv, void* out, float* lse, int32_t B, int32_t T, int32_t C, void* stream) {
    if (!qkv || !out || B <= 0 || T <= 0 || C <= 0) return DDNM_E_BADARG;
    if (C % 64 || T % 64) return DDNM_E_SHAPE;
    const int nh = C / 64;
    // logits = (q * s) . (k * s) with s = 64^-1/4 (unet.py:348-350)  ->  q.k / 8, evaluated in the exp2 domain
    const float scale_log2 = 0.125f * 1.4426950408889634f;
    hipStream_t s = (hipStream_t)stream;
    if (T % 128 == 0) {
        DDNM_LAUNCH((attn16_d64_kernel<4>), dim3(T / 128, nh, B), dim3(256), 0, s, reinterpret_cast<const _Float16*>(qkv),
                    reinterpret_cast<_Float16*>(out), lse, T, C, scale_log2);
    } else {
        DDNM_LAUNCH((attn16_d64_kernel<2>), dim3(T / 64, nh, B), dim3(128), 0, s, reinterpret_cast<const _Float16*>(qkv),
                    reinterpret_cast<_Float16*>(out), lse, T, C, scale_log2);
    }
    return 0;
}

extern "C" int ddnm_attn16_d64(const void* qkv, void* out, int32_t B, int32_t T, int32_t C, void* stream) {
    return ddnm_attn16_d64_lse(qkv, out, nullptr, B, T, C, stream);
}
