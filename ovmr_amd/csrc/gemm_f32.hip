// fp32 GEMM on the exact-f32 MFMA (v_mfma_f32_16x16x4_f32) for the OVMR visual-token generator.
//
// The aggregator (trainers/mm_classifier_one_prompt.py:138-143; blocks clip/model.py:219-252) is
// created after convert_weights() and its input is promoted to fp32 by the cat with the fp32
// cls_token (:167-168), so the reference runs it in fp32.  It is 0.08 % of the path's FLOPs; this
// kernel keeps it in exact fp32 (the f32 MFMA is bit-for-bit an fmaf chain) rather than fp16.
//
// 64x64x32 tile, 4 waves (2x2), each wave 32x32 = 2x2 tiles of 16x16.  Lane (r=l&15, g=l>>4)
// reads 4 consecutive k (one ds_read_b128) and feeds element t to MFMA step t; A and W use the
// same k permutation so the contraction is complete.  W is the MFMA A operand (C^T in registers,
// four consecutive columns per lane -> 16-byte epilogue accesses).
#include "common.h"

namespace {

constexpr int FBK = 32;

template <int EPI>
__global__ __launch_bounds__(256) void gemm_f32_t64(GemmArgs a) {
    __shared__ __attribute__((aligned(16))) float sA[2][64 * FBK];
    __shared__ __attribute__((aligned(16))) float sB[2][64 * FBK];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (a.N + 63) >> 6;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
    const int m0 = tm << 6, n0 = tn << 6;
    const float* A = (const float*)a.A;
    const float* W = (const float*)a.W;

    const int lc = tid & 7, lr = tid >> 3;          // chunk (4 floats) and row (0..31)
    const int sw = ((lc ^ (lr & 7)) << 2);
    const float* ga[2];
    const float* gb[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        int r = lr + 32 * i;
        ga[i] = A + (long)min(m0 + r, a.M - 1) * a.lda + lc * 4;
        gb[i] = W + (long)min(n0 + r, a.N - 1) * a.ldw + lc * 4;
    }
    float4_t ra[2], rb[2];
    const int nk = a.K / FBK;
#pragma unroll
    for (int i = 0; i < 2; ++i) { ra[i] = *(const float4_t*)ga[i]; rb[i] = *(const float4_t*)gb[i]; }
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        *(float4_t*)(&sA[0][(lr + 32 * i) * FBK + sw]) = ra[i];
        *(float4_t*)(&sB[0][(lr + 32 * i) * FBK + sw]) = rb[i];
    }
    __syncthreads();

    float4_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fg = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = (kt + 1 < nk);
        if (more) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ra[i] = *(const float4_t*)(ga[i] + (long)(kt + 1) * FBK);
                rb[i] = *(const float4_t*)(gb[i] + (long)(kt + 1) * FBK);
            }
        }
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            float4_t fa[2], fb[2];
            const int ch = (((kk << 2) + fg) ^ (fr & 7)) << 2;
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                fa[t] = *(const float4_t*)(&sA[cur][(wm * 32 + t * 16 + fr) * FBK + ch]);
                fb[t] = *(const float4_t*)(&sB[cur][(wn * 32 + t * 16 + fr) * FBK + ch]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(fb[j][e], fa[i][e], acc[i][j], 0, 0, 0);
        }
        if (more) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                *(float4_t*)(&sA[cur ^ 1][(lr + 32 * i) * FBK + sw]) = ra[i];
                *(float4_t*)(&sB[cur ^ 1][(lr + 32 * i) * FBK + sw]) = rb[i];
            }
        }
        __syncthreads();
    }

    float* C = (float*)a.C;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int m = m0 + wm * 32 + i * 16 + fr;
            const int n = n0 + wn * 32 + j * 16 + fg * 4;
            if (m >= a.M) continue;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                if (n + r >= a.N) continue;
                float x = acc[i][j][r];
                if (EPI != EPI_NONE) x += ((const float*)a.bias)[n + r];
                if (EPI == EPI_BIAS_QGELU) x = x * (1.0f / (1.0f + __expf(-1.702f * x)));
                if (EPI == EPI_BIAS_RES) x += ((const float*)a.res)[(long)m * a.ldres + n + r];
                C[(long)m * a.ldc + n + r] = x;
            }
        }
}

}  // namespace

int launch_gemm_f32(const GemmArgs& a, hipStream_t s) {
    if (a.M <= 0 || a.N <= 0) return 0;
    if (a.K <= 0 || (a.K % FBK) != 0 || (a.lda & 3) || (a.ldw & 3)) return -2;
    const int tiles = ((a.M + 63) / 64) * ((a.N + 63) / 64);
    switch (a.epi) {
        case EPI_NONE: hipLaunchKernelGGL(gemm_f32_t64<EPI_NONE>, dim3(tiles), dim3(256), 0, s, a); break;
        case EPI_BIAS: hipLaunchKernelGGL(gemm_f32_t64<EPI_BIAS>, dim3(tiles), dim3(256), 0, s, a); break;
        case EPI_BIAS_QGELU: hipLaunchKernelGGL(gemm_f32_t64<EPI_BIAS_QGELU>, dim3(tiles), dim3(256), 0, s, a); break;
        case EPI_BIAS_RES: hipLaunchKernelGGL(gemm_f32_t64<EPI_BIAS_RES>, dim3(tiles), dim3(256), 0, s, a); break;
        default: return -3;
    }
    return (int)hipGetLastError();
}
