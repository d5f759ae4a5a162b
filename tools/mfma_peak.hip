// Practical MFMA issue rate of this MI355X (development tool): register-only loops of v_mfma_f32_32x32x2_f32 and
// v_mfma_f32_32x32x16_f16, 4 independent accumulator tiles per wave, 1 / 2 / 3 waves per SIMD.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o gpurun_out/mfma_peak && gpurun_out/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 half8 __attribute__((ext_vector_type(8)));

// RANDOM = operands change every iteration (xorshift bits reinterpreted as finite floats / halfs): the data-dependent
// switching power of a real GEMM, instead of constant operands
template <int F16, int RANDOM>
__global__ void mfma_loop(float* out, int iters) {
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t)
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const float a = (float)threadIdx.x * 1e-3f, b = (float)blockIdx.x * 1e-3f;
    half8 ah, bh;
    for (int r = 0; r < 8; ++r) { ah[r] = (_Float16)a; bh[r] = (_Float16)b; }
    unsigned rs = 0x9E3779B9u * (threadIdx.x + 1) + blockIdx.x;
    float ar = a, br = b;
    for (int i = 0; i < iters; ++i) {
        if (RANDOM) {                        // new operand bits every iteration (values stay in [1, 2) / [-2, -1))
            rs ^= rs << 13; rs ^= rs >> 17; rs ^= rs << 5;
            ar = __uint_as_float(0x3F800000u | (rs & 0x807FFFFFu));
            br = __uint_as_float(0x3F800000u | ((rs * 2654435761u) & 0x807FFFFFu));
            typedef unsigned short us8 __attribute__((ext_vector_type(8)));
            us8 ua, ub;
            for (int r = 0; r < 8; ++r) {
                ua[r] = (unsigned short)(0x3C00u | ((rs >> r) & 0x83FFu));
                ub[r] = (unsigned short)(0x3C00u | ((rs >> (r + 8)) & 0x83FFu));
            }
            ah = __builtin_bit_cast(half8, ua);
            bh = __builtin_bit_cast(half8, ub);
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                if (F16) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc[t], 0, 0, 0);
                else acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(RANDOM ? ar : a, RANDOM ? br : b, acc[t], 0, 0, 0);
            }
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t)
        for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int F16, int RANDOM>
static void run(const char* name, int waves_per_simd, double flop_per_mfma) {
    const int cus = 256, iters = 60000;
    const int blocks = cus * waves_per_simd, threads = 256;       // 4 waves per block = 1 per SIMD
    float* out;
    hipMalloc(&out, sizeof(float) * blocks * threads);
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((mfma_loop<F16, RANDOM>), dim3(blocks), dim3(threads), 0, 0, out, 100);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((mfma_loop<F16, RANDOM>), dim3(blocks), dim3(threads), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfmas = (double)blocks * 4 * iters * 32;
    printf("%-34s %d wave(s)/SIMD: %8.2f ms  %8.1f TFLOP/s\n", name, waves_per_simd, ms, mfmas * flop_per_mfma / (ms * 1e-3) / 1e12);
    hipFree(out);
}

int main() {
    for (int w = 1; w <= 2; ++w) run<0, 0>("v_mfma_f32_32x32x2_f32 const", w, 2.0 * 32 * 32 * 2);
    for (int w = 1; w <= 2; ++w) run<0, 1>("v_mfma_f32_32x32x2_f32 random", w, 2.0 * 32 * 32 * 2);
    for (int w = 1; w <= 2; ++w) run<1, 0>("v_mfma_f32_32x32x16_f16 const", w, 2.0 * 32 * 32 * 16);
    for (int w = 1; w <= 2; ++w) run<1, 1>("v_mfma_f32_32x32x16_f16 random", w, 2.0 * 32 * 32 * 16);
    return 0;
}
