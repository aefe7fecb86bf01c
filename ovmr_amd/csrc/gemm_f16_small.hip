// fp16 MFMA GEMM for LATENCY-bound shapes: C[M,N] = epi(A[M,K] * W[N,K]^T) when the whole problem is a fraction of one round
// of the 256-row / 128-row tile kernels (gemm_f16_v5.hip) -- the text tower on a rank's 100-odd prompts, the CLS-row chain of
// the last vision block, the classifier head of a small shard.  There a 128 x 256 tile grid is 10-60 workgroups on 256 CUs
// and every workgroup walks its K-tiles one global-memory latency at a time (r04a: c_proj of the text tower, M = 1500,
// N = 512, K = 2048: 24 workgroups, 41 us = 1.3 us per K-tile; the 128 x 128 register-staged kernel at M = 40: 52 us).
//
// This kernel trades operand reuse for parallelism and memory-level parallelism:
//   * 64 x 64 output tile per workgroup (M = 1500, N = 512: 192 workgroups instead of 24);
//   * the four waves of a workgroup split K four ways (wave w owns the K range [w K/4, (w+1) K/4)) and each computes the WHOLE
//     64 x 64 tile over its range -- 16 accumulator tiles of `v_mfma_f32_16x16x32_f16` per wave;
//   * operands go global -> registers directly in the MFMA fragment layout (lane (r, g): 16 bytes of row r at k-chunk g: a wave
//     reads 16 rows x 64 contiguous bytes per instruction; the operands of these shapes are L2 / Infinity-Cache resident), D
//     K-steps of 32 ahead (D x 8 loads of 16 bytes in flight per lane: the loop runs at the issue rate, not at one latency per step);
//   * the four partial tiles are summed through LDS in wave order ((p0 + p1) + p2) + p3 -- deterministic -- wave w finishing
//     rows 16 w .. 16 w + 15 of the tile with the shared epilogue (gemm_epi.h: same rounding points as the other kernels).
// Same C^T accumulator convention as gemm_f16.hip: lane l owns row l & 15 and four consecutive columns 4 (l >> 4) ..+3.
#include "common.h"
#include "gemm_epi.h"

namespace {

template <int EPI, int D>
__global__ __launch_bounds__(256) void gemm_f16_s64(GemmArgs a, int groups) {
    __shared__ __attribute__((aligned(16))) float4_t part[4 * 3 * 4 * 64];      // [row block i][source slot][j][lane]: 48 KiB

    const int tid = threadIdx.x, lane = tid & 63, w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int tiles_n = (a.N + 63) >> 6;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
    const int m0 = tm << 6, n0 = tn << 6;

    const half_t* A = (const half_t*)a.A;
    const half_t* W = (const half_t*)a.W;
    const long kq = (long)w * (a.K >> 2) + fg * 8;                  // this wave's K range, this lane's 16-byte chunk
    const half_t* pa[4];
    const half_t* pb[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        pa[t] = A + (long)min(m0 + t * 16 + fr, a.M - 1) * a.lda + kq;
        pb[t] = W + (long)min(n0 + t * 16 + fr, a.N - 1) * a.ldw + kq;
    }

    float4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

    half8_t fa[D][4], fb[D][4];
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            fa[d][t] = *(const half8_t*)(pa[t] + d * 32);
            fb[d][t] = *(const half8_t*)(pb[t] + d * 32);
        }
    for (int g = 1; g < groups; ++g) {
#pragma unroll
        for (int d = 0; d < D; ++d) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[d][j], fa[d][i], acc[i][j], 0, 0, 0);
            const int k = (g * D + d) * 32;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                fa[d][t] = *(const half8_t*)(pa[t] + k);
                fb[d][t] = *(const half8_t*)(pb[t] + k);
            }
        }
    }
#pragma unroll
    for (int d = 0; d < D; ++d)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[d][j], fa[d][i], acc[i][j], 0, 0, 0);

    // wave w keeps row block w; its other three row blocks go to the owners through LDS (slot = source wave, minus one above the owner)
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        if (i == w) continue;
        const int slot = w - (w > i ? 1 : 0);
#pragma unroll
        for (int j = 0; j < 4; ++j) part[((i * 3 + slot) * 4 + j) * 64 + lane] = acc[i][j];
    }
    __syncthreads();
    float4_t sum[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float4_t p[4];
#pragma unroll
        for (int src = 0; src < 4; ++src) {
            if (src == w) {
                // (the compiler needs a constant first index: select the own block without dynamic register indexing)
                p[src] = w == 0 ? acc[0][j] : w == 1 ? acc[1][j] : w == 2 ? acc[2][j] : acc[3][j];
            } else {
                const int slot = src - (src > w ? 1 : 0);
                p[src] = part[((w * 3 + slot) * 4 + j) * 64 + lane];
            }
        }
        sum[j] = ((p[0] + p[1]) + p[2]) + p[3];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) epilogue_store<EPI>(a, m0 + w * 16 + fr, n0 + j * 16 + fg * 4, sum[j]);
}

template <int EPI>
int launch_s64(const GemmArgs& a, hipStream_t s) {
    const int steps = a.K >> 7;                                   // K-steps of 32 per wave
    const int tiles = ((a.M + 63) >> 6) * ((a.N + 63) >> 6);
    // D = 4 (32 loads in flight per lane, 284 registers: one wave per SIMD) while the grid leaves at most one workgroup per CU anyway
    // (the shapes launch_gemm_f16 routes here); larger grids (variant 9 in the tests) keep two workgroups per CU resident (D <= 3)
    if (steps % 4 == 0 && tiles <= 256) hipLaunchKernelGGL((gemm_f16_s64<EPI, 4>), dim3(tiles), dim3(256), 0, s, a, steps / 4);
    else if (steps % 3 == 0) hipLaunchKernelGGL((gemm_f16_s64<EPI, 3>), dim3(tiles), dim3(256), 0, s, a, steps / 3);
    else if (steps % 2 == 0) hipLaunchKernelGGL((gemm_f16_s64<EPI, 2>), dim3(tiles), dim3(256), 0, s, a, steps / 2);
    else hipLaunchKernelGGL((gemm_f16_s64<EPI, 1>), dim3(tiles), dim3(256), 0, s, a, steps);
    return (int)hipGetLastError();
}

}  // namespace

// -100: shape / epilogue not taken (the caller falls back to the tile kernels).
int launch_gemm_f16_small(const GemmArgs& a, hipStream_t s) {
    if (a.K < 128 || (a.K & 127) || (a.lda & 7) || (a.ldw & 7) || ((uintptr_t)a.A & 15) || ((uintptr_t)a.W & 15) ||
        a.im2col_R || a.stats_out || a.a_blocked || a.w_blocked)
        return -100;
    switch (a.epi) {
        case EPI_NONE: return launch_s64<EPI_NONE>(a, s);
        case EPI_BIAS: return launch_s64<EPI_BIAS>(a, s);
        case EPI_BIAS_QGELU: return launch_s64<EPI_BIAS_QGELU>(a, s);
        case EPI_BIAS_RES: return launch_s64<EPI_BIAS_RES>(a, s);
        case EPI_SCALE: return launch_s64<EPI_SCALE>(a, s);
    }
    return -100;
}
