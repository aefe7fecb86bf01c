// LayerNorm with fp32 statistics (clip/model.py:153-159): y = cast(LN_fp32(float(x)) * g + b).
// One wave per row, the row lives in registers (D <= 2048), two-pass mean/variance like
// torch's CPU kernel, 8-byte (fp16) / 16-byte (fp32) vector accesses, wave64 shuffle reductions.
// HBM-bound: algorithmic traffic = rows * D * (in + out) bytes.
#include "common.h"

namespace {

template <typename T, int MAXIT>
__global__ __launch_bounds__(256) void layernorm_rows(const T* __restrict__ x, T* __restrict__ y,
                                                      const float* __restrict__ g, const float* __restrict__ b,
                                                      int rows, int D, long in_stride) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const T* xr = x + (long)row * in_stride;
    T* yr = y + (long)row * D;
    float v[MAXIT][4];
    float sum = 0.f;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int c = (it * 64 + lane) * 4;
        if (c < D) {
            if (sizeof(T) == 2) {
                half4_t h = *(const half4_t*)((const half_t*)xr + c);
#pragma unroll
                for (int k = 0; k < 4; ++k) v[it][k] = (float)h[k];
            } else {
                float4_t f = *(const float4_t*)((const float*)xr + c);
#pragma unroll
                for (int k = 0; k < 4; ++k) v[it][k] = f[k];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) sum += v[it][k];
        } else {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[it][k] = 0.f;
        }
    }
    const float mean = wave_sum(sum) / (float)D;
    float sq = 0.f;
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int c = (it * 64 + lane) * 4;
        if (c < D) {
#pragma unroll
            for (int k = 0; k < 4; ++k) { float d = v[it][k] - mean; sq += d * d; }
        }
    }
    const float rstd = rsqrtf(wave_sum(sq) / (float)D + 1e-5f);
#pragma unroll
    for (int it = 0; it < MAXIT; ++it) {
        const int c = (it * 64 + lane) * 4;
        if (c < D) {
            float4_t gg = *(const float4_t*)(g + c);
            float4_t bb = *(const float4_t*)(b + c);
            float o[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) o[k] = (v[it][k] - mean) * rstd * gg[k] + bb[k];
            if (sizeof(T) == 2) {
                half4_t h;
#pragma unroll
                for (int k = 0; k < 4; ++k) h[k] = (half_t)o[k];
                *(half4_t*)((half_t*)yr + c) = h;
            } else {
                *(float4_t*)((float*)yr + c) = (float4_t){o[0], o[1], o[2], o[3]};
            }
        }
    }
}

// Partial row statistics of an fp16 matrix in the layout the LN-folding GEMM epilogue consumes (common.h, EPI_LN_BIAS):
// slot 0 = (sum, sum of squares) of the whole row, the other slots zero.  Used once per tower call, in front of the
// first block; every later LayerNorm gets its statistics from the EPI_BIAS_RES epilogue that produced its input.
__global__ __launch_bounds__(256) void row_stats_kernel(const half_t* __restrict__ x, float* __restrict__ stats, int rows, int D, int slots) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const half_t* xr = x + (long)row * D;
    float s = 0.f, q = 0.f;
    for (int c = lane * 8; c < D; c += 512) {
        half8_t h = *(const half8_t*)(xr + c);
#pragma unroll
        for (int k = 0; k < 8; ++k) { const float v = (float)h[k]; s += v; q += v * v; }
    }
    s = wave_sum(s);
    q = wave_sum(q);
    if (lane < slots) *(float2_t*)(stats + ((long)row * slots + lane) * 2) = lane == 0 ? (float2_t){s, q} : (float2_t){0.f, 0.f};
}

// W'[n,k] = h(gamma[k] * W[n,k]);  g[n] = sum_k W'[n,k];  b[n] = bias[n] + sum_k beta[k] * W[n,k]   (one wave per row n)
__global__ __launch_bounds__(256) void fold_ln_kernel(const half_t* __restrict__ W, const float* __restrict__ gamma,
                                                      const float* __restrict__ beta, const half_t* __restrict__ bias,
                                                      half_t* __restrict__ Wf, float* __restrict__ g, float* __restrict__ b, int N, int K) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float sg = 0.f, sb = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float w = (float)W[(long)n * K + k];
        const half_t wf = (half_t)(gamma[k] * w);
        Wf[(long)n * K + k] = wf;
        sg += (float)wf;
        sb += beta[k] * w;
    }
    sg = wave_sum(sg);
    sb = wave_sum(sb);
    if (lane == 0) { g[n] = sg; b[n] = sb + (float)bias[n]; }
}

}  // namespace

int launch_row_stats(const half_t* x, float* stats, int rows, int D, int slots, hipStream_t s) {
    if (rows <= 0) return 0;
    if ((D & 7) || slots < 1 || slots > 64) return -2;
    hipLaunchKernelGGL(row_stats_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, stats, rows, D, slots);
    return (int)hipGetLastError();
}

int launch_fold_ln(const half_t* W, const float* gamma, const float* beta, const half_t* bias, half_t* Wf, float* g, float* b,
                   int N, int K, hipStream_t s) {
    hipLaunchKernelGGL(fold_ln_kernel, dim3((N + 3) / 4), dim3(256), 0, s, W, gamma, beta, bias, Wf, g, b, N, K);
    return (int)hipGetLastError();
}

int launch_layernorm(const void* x, void* y, const float* g, const float* b, int rows, int D,
                     long in_row_stride, int is_f32, hipStream_t s) {
    if (rows <= 0) return 0;
    if ((D & 3) || D > 2048 || (in_row_stride & 3)) return -2;
    const int grid = (rows + 3) / 4;
    const bool small = D <= 1024;
    if (is_f32) {
        if (small) hipLaunchKernelGGL((layernorm_rows<float, 4>), dim3(grid), dim3(256), 0, s, (const float*)x, (float*)y, g, b, rows, D, in_row_stride);
        else hipLaunchKernelGGL((layernorm_rows<float, 8>), dim3(grid), dim3(256), 0, s, (const float*)x, (float*)y, g, b, rows, D, in_row_stride);
    } else {
        if (small) hipLaunchKernelGGL((layernorm_rows<half_t, 4>), dim3(grid), dim3(256), 0, s, (const half_t*)x, (half_t*)y, g, b, rows, D, in_row_stride);
        else hipLaunchKernelGGL((layernorm_rows<half_t, 8>), dim3(grid), dim3(256), 0, s, (const half_t*)x, (half_t*)y, g, b, rows, D, in_row_stride);
    }
    return (int)hipGetLastError();
}
