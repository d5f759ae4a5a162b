// Counter-based Gaussian noise for the sampler step (gfx950): Philox4x32-10 (Salmon et al., SC'11; the generator behind
// torch.randn on GPUs) + Box-Muller.  The reference draws `torch.randn_like(x)` once per loop iteration
// (functions/svd_ddnm.py:65,74); here the draw can happen INSIDE the step kernels -- no noise tensor is written and read
// back -- with the counter (element / 4, iteration, GLOBAL image index, 0) and the key (seed): an image's noise does not
// depend on the batch it sits in or on the rank that restores it, so sharded runs reproduce the unsharded one.
#pragma once
#include "common.h"

struct PhiloxKey { unsigned k0, k1; };

__device__ __forceinline__ void philox_round(unsigned (&c)[4], unsigned k0, unsigned k1) {
    const unsigned long long p0 = 0xD2511F53ull * c[0], p1 = 0xCD9E8D57ull * c[2];
    const unsigned h0 = (unsigned)(p0 >> 32), l0 = (unsigned)p0, h1 = (unsigned)(p1 >> 32), l1 = (unsigned)p1;
    c[0] = h1 ^ c[1] ^ k0;
    c[1] = l1;
    c[2] = h0 ^ c[3] ^ k1;
    c[3] = l0;
}

// Philox4x32-10: 4 x 32 random bits from a 128-bit counter and a 64-bit key
__device__ __forceinline__ void philox4x32_10(unsigned (&c)[4], unsigned k0, unsigned k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        philox_round(c, k0, k1);
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

// four independent N(0, 1) values for counter (c0, c1, c2, 0): uniforms (u + 0.5) * 2^-32 in (0, 1), two Box-Muller pairs
__device__ __forceinline__ f32x4 philox_normal4(PhiloxKey key, unsigned c0, unsigned c1, unsigned c2) {
    unsigned c[4] = {c0, c1, c2, 0u};
    philox4x32_10(c, key.k0, key.k1);
    const float s = 2.3283064365386963e-10f;         // 2^-32
    const float u0 = ((float)c[0] + 0.5f) * s, u1 = ((float)c[1] + 0.5f) * s;
    const float u2 = ((float)c[2] + 0.5f) * s, u3 = ((float)c[3] + 0.5f) * s;
    // (float)c rounds to 2^32 for the largest counters: clamp below 1 so that log() stays finite and negative
    const float a0 = fminf(u0, 0.99999994f), a2 = fminf(u2, 0.99999994f);
    const float r0 = sqrtf(-2.0f * logf(a0)), r1 = sqrtf(-2.0f * logf(a2));
    float s0, c0f, s1, c1f;
    sincosf(6.283185307179586f * u1, &s0, &c0f);
    sincosf(6.283185307179586f * u3, &s1, &c1f);
    return f32x4{r0 * c0f, r0 * s0, r1 * c1f, r1 * s1};
}
