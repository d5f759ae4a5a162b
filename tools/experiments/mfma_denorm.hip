// Does v_mfma_f32_32x32x16_f16 honour fp16 subnormal inputs on gfx950?  (development probe, not product code)
// out[0] = sum_k a_k * b_k with a = 2^-20 (fp16 subnormal), b = 1  -> 16 * 2^-20 if honoured, 0 if flushed
// out[1] = same through the VALU conversion path: (float)(_Float16)(2^-20f)
#include <hip/hip_runtime.h>
typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
__global__ void k(float* out, float tiny) {
    half8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)tiny; b[i] = (_Float16)1.0f; }
    f32x16 c = {};
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)(_Float16)tiny; }
}
extern "C" int mfma_denorm_probe(float* out, float tiny, void* stream) {
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, (hipStream_t)stream, out, tiny);
    return (int)hipGetLastError();
}
