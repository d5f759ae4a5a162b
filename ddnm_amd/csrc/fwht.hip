// Walsh-Hadamard compressive-sensing operator kernels, gfx950.
//
// The reference transform (functions/svd_operators.py:212-222) is a natural-order FWHT over
// the n*n pixels of one channel: log2(n*n) butterfly stages, each cloning the whole array
// (~48 full-tensor passes per call).  Over the row-major index the low log2(n) stages act
// inside a row and the high ones across rows, so it factors as H_n (x) H_n:
//   rows pass : each 64-lane wave owns rows of n floats; 2 stages in-lane (float4), the rest
//               through wavefront shuffles (ds_bpermute-free __shfl_xor on 64 lanes);
//   cols pass : a [n][32]-column strip lives in LDS (pitch 33), log2(n) stages in place;
//               the masked variant applies  H_col (mask .* H_col .)  while the strip is resident.
// The projection of a DDNM step,  A^+ A x0 = H (W .* (H x0)),  therefore costs 3 passes over
// the image instead of ~96.
#include "common.h"

// ---- rows pass: out[p][r][:] = scale * H_n in[p][r][:]
__global__ __launch_bounds__(256) void fwht_rows_kernel(const float* __restrict__ in, float* __restrict__ out, int n,
                                                        int64_t total_rows, float scale) {
    const int lanes_per_row = n >> 2;                 // float4 per lane
    const int rows_per_wave = 64 / lanes_per_row;
    const int lane = threadIdx.x & 63;
    const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int64_t row = wave * rows_per_wave + lane / lanes_per_row;
    const int l = lane % lanes_per_row;
    const bool ok = row < total_rows;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (ok) v = *reinterpret_cast<const f32x4*>(in + row * n + l * 4);
    // h = 1
    { const float a = v.x, b = v.y, c = v.z, d = v.w; v.x = a + b; v.y = a - b; v.z = c + d; v.w = c - d; }
    // h = 2
    { const float a = v.x, b = v.y, c = v.z, d = v.w; v.x = a + c; v.y = b + d; v.z = a - c; v.w = b - d; }
    // h = 4 .. n/2 across lanes (distance h/4 inside the row's lane group)
    for (int dlt = 1; dlt < lanes_per_row; dlt <<= 1) {
        f32x4 o;
        o.x = __shfl_xor(v.x, dlt);
        o.y = __shfl_xor(v.y, dlt);
        o.z = __shfl_xor(v.z, dlt);
        o.w = __shfl_xor(v.w, dlt);
        v = (l & dlt) ? (o - v) : (v + o);
    }
    if (ok) *reinterpret_cast<f32x4*>(out + row * n + l * 4) = v * scale;
}

// ---- cols pass (optionally masked, forward+inverse in one residency)
constexpr int CS = 32;   // columns per strip
template <bool MASKED>
__global__ __launch_bounds__(256) void fwht_cols_kernel(const float* __restrict__ in, const float* __restrict__ mask,
                                                        int planes_mask, float* __restrict__ out, int n,
                                                        float scale) {
    extern __shared__ __attribute__((aligned(16))) float tile[];   // [n][CS+1]
    const int strips = n / CS;
    const int plane = blockIdx.x / strips, strip = blockIdx.x - plane * strips;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 8 row lanes
    const float* src = in + (size_t)plane * n * n + strip * CS;
    float* dst = out + (size_t)plane * n * n + strip * CS;
    for (int r = ty; r < n; r += 8) tile[r * (CS + 1) + tx] = src[(size_t)r * n + tx];
    __syncthreads();
    auto butterflies = [&]() {
        for (int h = 1; h < n; h <<= 1) {
            for (int q = ty; q < n / 2; q += 8) {            // q-th butterfly of this column
                const int i = ((q / h) * 2 * h) + (q % h);
                const float a = tile[i * (CS + 1) + tx], b = tile[(i + h) * (CS + 1) + tx];
                tile[i * (CS + 1) + tx] = a + b;
                tile[(i + h) * (CS + 1) + tx] = a - b;
            }
            __syncthreads();
        }
    };
    butterflies();
    if (MASKED) {
        const float* m = mask + (size_t)(plane % planes_mask) * n * n + strip * CS;
        for (int r = ty; r < n; r += 8) tile[r * (CS + 1) + tx] *= m[(size_t)r * n + tx];
        __syncthreads();
        butterflies();
    }
    for (int r = ty; r < n; r += 8) dst[(size_t)r * n + tx] = tile[r * (CS + 1) + tx] * scale;
}

// ---- cols pass, register-resident (round 6; n >= 64): the strip's log2(n) butterfly stages used to be log2(n) read-modify-
// write passes over an LDS image with a barrier each (16 for the masked form: 0.05 ... 0.10 of the byte floor).  Here a thread
// (column tx, row group ty of 8) holds n / 8 CONSECUTIVE rows of its column in registers: the low log2(n / 8) stages are
// register arithmetic, ONE transposition through LDS regroups the strip so that the same thread holds rows r + (n / 8) k,
// k = 0..7, and the top three stages are register arithmetic again.  The masked form multiplies by the mask in that layout
// (the mask is read once, coalesced) and runs the second transform the same way (back to consecutive rows, low stages, top
// stages): three barriers instead of sixteen, LDS traffic 6 x the strip instead of 32 x.  Same additions in the same order
// per element as the stage-by-stage kernel (the stages of a transform are NOT reordered): bit-identical results.
template <int N, bool MASKED>
__global__ __launch_bounds__(256) void fwht_cols_reg_kernel(const float* __restrict__ in, const float* __restrict__ mask,
                                                            int planes_mask, float* __restrict__ out, float scale) {
    constexpr int RPT = N / 8, Q = RPT / 8;          // rows per thread; cross-phase row slots per thread
    static_assert(Q >= 1, "n >= 64");
    __shared__ float tile[N * (CS + 1)];
    constexpr int strips = N / CS;
    const int plane = blockIdx.x / strips, strip = blockIdx.x - plane * strips;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float* src = in + (size_t)plane * N * N + strip * CS + tx;
    float* dst = out + (size_t)plane * N * N + strip * CS + tx;
    float v[RPT];
#pragma unroll
    for (int i = 0; i < RPT; ++i) v[i] = src[(size_t)(ty * RPT + i) * N];
    auto low = [&]() {                                // stages h = 1 .. RPT / 2 over the thread's consecutive rows
#pragma unroll
        for (int h = 1; h < RPT; h <<= 1)
#pragma unroll
            for (int i = 0; i < RPT; ++i)
                if (!(i & h)) { const float a = v[i], b = v[i + h]; v[i] = a + b; v[i + h] = a - b; }
    };
    auto cross = [&]() {                              // stages h = RPT, 2 RPT, 4 RPT: v[j * 8 + k] = row (ty Q + j) + RPT k
#pragma unroll
        for (int h = 1; h < 8; h <<= 1)
#pragma unroll
            for (int i = 0; i < RPT; ++i)
                if (!((i & 7) & h)) { const float a = v[i], b = v[i + h]; v[i] = a + b; v[i + h] = a - b; }
    };
    low();
#pragma unroll
    for (int i = 0; i < RPT; ++i) tile[(ty * RPT + i) * (CS + 1) + tx] = v[i];
    __syncthreads();
#pragma unroll
    for (int j = 0; j < Q; ++j)
#pragma unroll
        for (int k = 0; k < 8; ++k) v[j * 8 + k] = tile[(ty * Q + j + RPT * k) * (CS + 1) + tx];
    cross();
    if constexpr (MASKED) {
        const float* m = mask + (size_t)(plane % planes_mask) * N * N + strip * CS + tx;
#pragma unroll
        for (int j = 0; j < Q; ++j)
#pragma unroll
            for (int k = 0; k < 8; ++k) v[j * 8 + k] *= m[(size_t)(ty * Q + j + RPT * k) * N];
        // second transform: its low stages need the consecutive-row layout again.  H is applied stage by stage in ascending
        // order like the first one: low stages first -- so back through LDS, low(), and through LDS once more for the top
        // three.  (The stages commute mathematically, but not in floating point: the reference order is kept.)
#pragma unroll
        for (int j = 0; j < Q; ++j)
#pragma unroll
            for (int k = 0; k < 8; ++k) tile[(ty * Q + j + RPT * k) * (CS + 1) + tx] = v[j * 8 + k];      // own slots: no hazard
        __syncthreads();
#pragma unroll
        for (int i = 0; i < RPT; ++i) v[i] = tile[(ty * RPT + i) * (CS + 1) + tx];
        low();
#pragma unroll
        for (int i = 0; i < RPT; ++i) tile[(ty * RPT + i) * (CS + 1) + tx] = v[i];      // own slots again
        __syncthreads();
#pragma unroll
        for (int j = 0; j < Q; ++j)
#pragma unroll
            for (int k = 0; k < 8; ++k) v[j * 8 + k] = tile[(ty * Q + j + RPT * k) * (CS + 1) + tx];
        cross();
    }
#pragma unroll
    for (int j = 0; j < Q; ++j)
#pragma unroll
        for (int k = 0; k < 8; ++k) dst[(size_t)(ty * Q + j + RPT * k) * N] = v[j * 8 + k] * scale;
}

template <bool MASKED>
static int launch_cols(const float* in, const float* mask, int planes_mask, float* out, int planes, int n, float scale, hipStream_t s) {
    const dim3 grid(planes * (n / CS));
    switch (n) {
        case 256: DDNM_LAUNCH((fwht_cols_reg_kernel<256, MASKED>), grid, dim3(256), 0, s, in, mask, planes_mask, out, scale); break;
        case 128: DDNM_LAUNCH((fwht_cols_reg_kernel<128, MASKED>), grid, dim3(256), 0, s, in, mask, planes_mask, out, scale); break;
        case 64: DDNM_LAUNCH((fwht_cols_reg_kernel<64, MASKED>), grid, dim3(256), 0, s, in, mask, planes_mask, out, scale); break;
        default: DDNM_LAUNCH(fwht_cols_kernel<MASKED>, grid, dim3(256), n * (CS + 1) * sizeof(float), s, in, mask, planes_mask, out, n, scale);
    }
    return 0;
}

static bool fwht_n_ok(int n) { return n == 32 || n == 64 || n == 128 || n == 256; }

extern "C" int ddnm_fwht2d_f32(const float* in, float* out, int32_t planes, int32_t n, void* stream) {
    if (!in || !out || planes <= 0) return DDNM_E_BADARG;
    if (!fwht_n_ok(n)) return DDNM_E_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const int64_t rows = (int64_t)planes * n;
    const int rows_per_block = 4 * (64 / (n / 4));
    DDNM_LAUNCH(fwht_rows_kernel, dim3((unsigned)((rows + rows_per_block - 1) / rows_per_block)), dim3(256), 0,
                       s, in, out, n, rows, 1.0f);
    return launch_cols<false>(out, nullptr, 1, out, planes, n, 1.0f / (float)n, s);
}

extern "C" int ddnm_fwht2d_masked_f32(const float* in, const float* mask, int32_t planes_mask, float* out,
                                      int32_t planes, int32_t n, float* scratch, void* stream) {
    if (!in || !out || !mask || !scratch || planes <= 0 || planes_mask <= 0) return DDNM_E_BADARG;
    if (!fwht_n_ok(n)) return DDNM_E_SHAPE;
    hipStream_t s = (hipStream_t)stream;
    const int64_t rows = (int64_t)planes * n;
    const int rows_per_block = 4 * (64 / (n / 4));
    const dim3 grid_rows((unsigned)((rows + rows_per_block - 1) / rows_per_block));
    // H in (unnormalised rows), then cols-mask-cols with the forward 1/n, then rows with the inverse 1/n
    DDNM_LAUNCH(fwht_rows_kernel, grid_rows, dim3(256), 0, s, in, scratch, n, rows, 1.0f);
    { const int rc = launch_cols<true>(scratch, mask, planes_mask, scratch, planes, n, 1.0f / (float)n, s); if (rc) return rc; }
    DDNM_LAUNCH(fwht_rows_kernel, grid_rows, dim3(256), 0, s, scratch, out, n, rows, 1.0f / (float)n);
    return 0;
}

// y[b][k*C + c] = planes[b][c][perm[k]]   for k*C + c < n_keep   (svd_operators.py:236-237,249-251)
__global__ __launch_bounds__(256) void wh_gather_kernel(const float* __restrict__ planes, const int* __restrict__ perm,
                                                        float* __restrict__ y, int C, int64_t N, int64_t n_keep,
                                                        int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / n_keep, j = i - b * n_keep;
        const int64_t k = j / C;
        const int c = (int)(j - k * C);
        y[i] = planes[(b * C + c) * N + perm[k]];
    }
}

extern "C" int ddnm_wh_gather_f32(const float* planes, const int32_t* perm, float* y, int32_t B, int32_t C, int32_t N,
                                  int32_t n_keep, void* stream) {
    if (!planes || !perm || !y || B <= 0 || C <= 0 || N <= 0 || n_keep <= 0 || n_keep > (int64_t)C * N)
        return DDNM_E_BADARG;
    const int64_t total = (int64_t)B * n_keep;
    const unsigned grid = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    DDNM_LAUNCH(wh_gather_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, planes, perm, y, C, (int64_t)N,
                       (int64_t)n_keep, total);
    return 0;
}

// planes[b][c][perm[k]] = (k*C + c < n_keep) ? y[b][k*C + c] : 0      (covers every entry: perm is a bijection)
__global__ __launch_bounds__(256) void wh_scatter_kernel(const float* __restrict__ y, const int* __restrict__ perm,
                                                         float* __restrict__ planes, int C, int64_t N, int64_t n_keep,
                                                         int64_t total) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t b = i / (C * N), j = i - b * C * N;
        const int64_t k = j / C;
        const int c = (int)(j - k * C);
        planes[(b * C + c) * N + perm[k]] = j < n_keep ? y[b * n_keep + j] : 0.f;
    }
}

extern "C" int ddnm_wh_scatter_f32(const float* y, const int32_t* perm, float* planes, int32_t B, int32_t C, int32_t N,
                                   int32_t n_keep, void* stream) {
    if (!planes || !perm || !y || B <= 0 || C <= 0 || N <= 0 || n_keep <= 0 || n_keep > (int64_t)C * N)
        return DDNM_E_BADARG;
    const int64_t total = (int64_t)B * C * N;
    const unsigned grid = (unsigned)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    DDNM_LAUNCH(wh_scatter_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, y, perm, planes, C, (int64_t)N,
                       (int64_t)n_keep, total);
    return 0;
}
