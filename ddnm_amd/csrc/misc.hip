// Small kernels of the embedding path and layout plumbing, gfx950.
#include "common.h"

// y[b][n] = bias[n] + sum_k W[n][k] * act(x[b][k]); one wave per output, coalesced over k.
// Replaces nn.Linear on the timestep embedding (guided_diffusion/models.py:216-222,306-308)
// and every ResnetBlock.temb_proj (models.py:92,121) -- the latter concatenated into ONE
// launch per step since they all read the same swish(temb).
__global__ __launch_bounds__(256) void linear_kernel(const float* __restrict__ x, const float* __restrict__ W,
                                                     const float* __restrict__ bias, float* __restrict__ y, int K,
                                                     int N, int silu_in) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6), b = blockIdx.y;
    if (n >= N) return;
    const float* xr = x + (size_t)b * K;
    const float* wr = W + (size_t)n * K;
    float acc = 0.f;
    for (int k = lane; k < K; k += 64) {
        float v = xr[k];
        if (silu_in) v = silu_f(v);
        acc += wr[k] * v;
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
    if (lane == 0) y[(size_t)b * N + n] = acc + (bias ? bias[n] : 0.f);
}

extern "C" int ddnm_linear_f32(const float* x, const float* W, const float* bias, float* y, int32_t B, int32_t K,
                               int32_t N, int32_t silu_in, void* stream) {
    if (!x || !W || !y || B <= 0 || K <= 0 || N <= 0) return DDNM_E_BADARG;
    DDNM_LAUNCH(linear_kernel, dim3((N + 3) / 4, B), dim3(256), 0, (hipStream_t)stream, x, W, bias, y, K, N,
                       silu_in);
    return 0;
}

__global__ void temb_kernel(const float* __restrict__ t, const float* __restrict__ freq, float* __restrict__ emb,
                            int half, int order) {
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < half; i += blockDim.x) {
        const float a = t[b] * freq[i];
        const float s = sinf(a), c = cosf(a);
        float* e = emb + (size_t)b * 2 * half;
        if (order == 0) { e[i] = s; e[half + i] = c; }
        else { e[i] = c; e[half + i] = s; }
    }
}

extern "C" int ddnm_timestep_embedding_f32(const float* t, const float* freq, float* emb, int32_t B, int32_t half,
                                           int32_t order, void* stream) {
    if (!t || !freq || !emb || B <= 0 || half <= 0) return DDNM_E_BADARG;
    DDNM_LAUNCH(temb_kernel, dim3(B), dim3(128), 0, (hipStream_t)stream, t, freq, emb, half, order);
    return 0;
}

// NCHW [B][C][HW] -> NHWC [B][HW][Cpad]; float4 stores, channels >= C are zero.
__global__ __launch_bounds__(256) void nchw_to_nhwc_pad_kernel(const float* __restrict__ src, float* __restrict__ dst,
                                                               int C, int HW, int Cpad, size_t total4) {
    const int q = Cpad >> 2;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total4; i += (size_t)gridDim.x * 256) {
        const int c4 = (int)(i % q);
        const size_t pix = i / q;                // b*HW + p
        const size_t b = pix / HW, p = pix - b * HW;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        const int c = c4 * 4;
        const float* s = src + (b * C) * HW + p;
        if (c + 0 < C) v.x = s[(size_t)(c + 0) * HW];
        if (c + 1 < C) v.y = s[(size_t)(c + 1) * HW];
        if (c + 2 < C) v.z = s[(size_t)(c + 2) * HW];
        if (c + 3 < C) v.w = s[(size_t)(c + 3) * HW];
        reinterpret_cast<f32x4*>(dst)[i] = v;
    }
}

extern "C" int ddnm_nchw_to_nhwc_pad_f32(const float* src, float* dst, int32_t B, int32_t C, int32_t HW, int32_t Cpad,
                                         void* stream) {
    if (!src || !dst || B <= 0 || C <= 0 || HW <= 0 || Cpad < C || (Cpad & 3)) return DDNM_E_BADARG;
    const size_t total4 = (size_t)B * HW * (Cpad / 4);
    const unsigned grid = (unsigned)((total4 + 255) / 256 < 4096 ? (total4 + 255) / 256 : 4096);
    DDNM_LAUNCH(nchw_to_nhwc_pad_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, src, dst, C, HW, Cpad,
                       total4);
    return 0;
}

extern "C" int ddnm_version(void) { return 1; }

extern "C" const char* ddnm_error_string(int code) {
    if (code == 0) return "success";
    if (code == DDNM_E_BADARG) return "ddnm: bad argument (null pointer, non-positive size or misalignment)";
    if (code == DDNM_E_SHAPE) return "ddnm: shape not supported by this kernel family";
    if (code > 0) return hipGetErrorString((hipError_t)code);
    return "ddnm: unknown error";
}
