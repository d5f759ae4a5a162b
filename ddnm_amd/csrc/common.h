// Shared device/host helpers for libddnm_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/ddnm_hip.h"

#define DDNM_LAUNCH_CHECK()                       \
    do {                                          \
        hipError_t e__ = hipGetLastError();       \
        if (e__ != hipSuccess) return (int)e__;   \
    } while (0)

// launch + error check that ignores stale errors left by unrelated runtime calls of the host process
#define DDNM_LAUNCH(...)                          \
    do {                                          \
        (void)hipGetLastError();                  \
        hipLaunchKernelGGL(__VA_ARGS__);          \
        hipError_t e__ = hipGetLastError();       \
        if (e__ != hipSuccess) return (int)e__;   \
    } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// MI355X: 8 XCDs, block b is dispatched to XCD b % 8 (performance only).  Remap so each
// XCD walks a contiguous range of tiles and neighbouring tiles share one L2.  Bijective for any n.
__device__ __forceinline__ int xcd_swizzle(int bid, int nwg) {
    const int xcd = bid & 7;
    const int q = nwg >> 3, r = nwg & 7;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + (bid >> 3);
}

// swish with the hardware reciprocal (v_rcp_f32, 1 ulp) instead of an IEEE division sequence
__device__ __forceinline__ float silu_f(float v) {
    return v * __builtin_amdgcn_rcpf(1.0f + __expf(-v));
}

// 32x32 MFMA tile step over 8 k-values held as two float4 (lanes 0-31: k0..k0+3, lanes 32-63: k0+4..k0+7)
__device__ __forceinline__ f32x16 mfma_k8(const f32x4 a, const f32x4 b, f32x16 c) {
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, c, 0, 0, 0);
    return c;
}

// LDS tile geometry shared by conv and gemm: rows of KC=32 floats padded to 36 (conflict-free ds_read_b128)
constexpr int KC = 32;
constexpr int LDT = KC + 4;
