// fp16 MFMA GEMM for gfx950: C[M,N] = epi(A[M,K] * W[N,K]^T), fp32 accumulate.
//
// This one kernel family carries ~96 % of the hot path's FLOPs: patch embed (K1), QKV (K4),
// out-proj (K6), c_fc+QuickGELU (K7), c_proj (K8), the CLS/EOS projections (K9,K15) and the
// classifier logits (K18,K21) of SURVEY.md section 2.3.
//
// Variant 0 ("t128"): 128x128x64 tile, 4 waves (2x2), each wave 64x64 = 4x4 MFMA 16x16x32 f16
// tiles; operands staged global -> registers -> LDS (double buffered, XOR-swizzled 128-byte
// rows so every ds_read_b128 lane group hits 16 distinct 16-byte slots).
// The MFMA is issued with W as the A operand and A as the B operand, so the accumulator holds
// C^T: lane l owns row m = l&15 and four CONSECUTIVE columns n = 4*(l>>4)..+3, which turns the
// epilogue (bias / residual / QuickGELU / positional add) into 8-byte vector loads and stores.
#include "common.h"
#include "gemm_epi.h"

namespace {

constexpr int BK = 64;

// ------------------------------------------------------------------------------------------
// Variant 0: 128x128x64, register-staged double buffer.
template <int EPI>
__global__ __launch_bounds__(256) void gemm_f16_t128(GemmArgs a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    half_t* sA = (half_t*)smem;               // [2][128*64]
    half_t* sB = sA + 2 * 128 * BK;           // [2][128*64]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int tiles_n = (a.N + 127) >> 7;
    const int tm = blockIdx.x / tiles_n, tn = blockIdx.x - tm * tiles_n;
    const int m0 = tm << 7, n0 = tn << 7;

    const half_t* A = (const half_t*)a.A;
    const half_t* W = (const half_t*)a.W;

    // staging map: 16-byte chunk id = tid + 256*i -> row = id>>3 (0..127), chunk = id&7
    const int lc = tid & 7, lr = tid >> 3;
    const int sw = ((lc ^ (lr & 7)) << 3);
    const half_t* ga[4];
    const half_t* gb[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        int r = lr + 32 * i;
        int ma = min(m0 + r, a.M - 1), nb = min(n0 + r, a.N - 1);
        ga[i] = A + (long)ma * a.lda + lc * 8;
        gb[i] = W + (long)nb * a.ldw + lc * 8;
    }
    uint4 ra[4], rb[4];
    const int nk = a.K / BK;

#pragma unroll
    for (int i = 0; i < 4; ++i) { ra[i] = *(const uint4*)ga[i]; rb[i] = *(const uint4*)gb[i]; }
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        *(uint4*)(sA + (lr + 32 * i) * BK + sw) = ra[i];
        *(uint4*)(sB + (lr + 32 * i) * BK + sw) = rb[i];
    }
    __syncthreads();

    float4_t acc[4][4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fg = lane >> 4;
    for (int kt = 0; kt < nk; ++kt) {
        const int cur = kt & 1;
        const bool more = (kt + 1 < nk);
        if (more) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                ra[i] = *(const uint4*)(ga[i] + (long)(kt + 1) * BK);
                rb[i] = *(const uint4*)(gb[i] + (long)(kt + 1) * BK);
            }
        }
        const half_t* cA = sA + cur * 128 * BK;
        const half_t* cB = sB + cur * 128 * BK;
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            half8_t fa[4], fb[4];
            const int ch = (((ks << 2) + fg) ^ (fr & 7)) << 3;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                fa[t] = *(const half8_t*)(cA + (wm * 64 + t * 16 + fr) * BK + ch);
                fb[t] = *(const half8_t*)(cB + (wn * 64 + t * 16 + fr) * BK + ch);
            }
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        }
        if (more) {
            half_t* nA = sA + (cur ^ 1) * 128 * BK;
            half_t* nB = sB + (cur ^ 1) * 128 * BK;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                *(uint4*)(nA + (lr + 32 * i) * BK + sw) = ra[i];
                *(uint4*)(nB + (lr + 32 * i) * BK + sw) = rb[i];
            }
        }
        __syncthreads();
    }

#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            epilogue_store<EPI>(a, m0 + wm * 64 + i * 16 + fr, n0 + wn * 64 + j * 16 + fg * 4, acc[i][j]);
}

template <int EPI>
int launch_t128(const GemmArgs& a, hipStream_t s) {
    const int tiles = ((a.M + 127) / 128) * ((a.N + 127) / 128);
    const size_t lds = 2 * 2 * 128 * BK * sizeof(half_t);
    static bool attr_set[OVMR_MAX_DEVICES] = {};      // the attribute is per device (one process may drive several)
    int dev = 0;
    HIP_CHECK_RET(hipGetDevice(&dev));
    if (dev >= 0 && dev < OVMR_MAX_DEVICES && !attr_set[dev]) {
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)gemm_f16_t128<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL(gemm_f16_t128<EPI>, dim3(tiles), dim3(256), lds, s, a);
    return (int)hipGetLastError();
}

}  // namespace

int launch_gemm_f16_v5(const GemmArgs& a, int variant, hipStream_t s);  // gemm_f16_v5.hip: 256(128)x256x64 LDS-DMA tiles, fused epilogues
#ifdef OVMR_EXPERIMENTS
int launch_gemm_f16_w2(const GemmArgs& a, hipStream_t s);               // experiments/gemm_f16_w2.hip
#endif
int launch_gemm_f16_small(const GemmArgs& a, hipStream_t s);            // gemm_f16_small.hip: 64x64 tiles, K split over the waves (latency-bound shapes)

// Shapes that are at most ONE round of 64 x 64 tiles on the 256 CUs: there the K loop of a tile kernel runs at one memory latency
// per K-tile on a mostly idle machine and the split-K kernel (gemm_f16_small.hip) wins -- 8-9 us against 10-14 us at K = 512,
// 20 against 28 at K = 2048, 27 against 43 at K = 3072, 3-5x at a few dozen rows; beyond one round its missing operand reuse costs
// more than the latency it hides (profiles/r04d_small_gemm_bench.log: the rule matches the faster kernel on all 81 measured shapes
// but the two within 5 %).
bool gemm_f16_is_small(int M, int N) { return (long)((M + 63) / 64) * ((N + 63) / 64) <= 256; }

// variant 0: the 128x128 register-staged kernel above for every shape; 6: 256-row tiles with the double-buffered K loop;
// 8 (default): 6 with the ping-pong K loop, and the 64x64 split-K kernel (gemm_f16_small.hip) for latency-bound shapes
// (gemm_f16_is_small); 7: 8 without that kernel (A/B); 9: the 64x64 split-K kernel wherever it takes the shape (tests).  6 / 8 / 9 fall back to the 128x128
// kernel for shapes they do not take (M < 256, N < 128, K not a multiple of 128 ...).  Experiment builds (OVMR_EXPERIMENTS) add
// timing-only ablation variants (gemm_f16_v5.hip).
int launch_gemm_f16(const GemmArgs& a_in, int variant, hipStream_t s) {
    GemmArgs a = a_in;
    if (variant >= 100) {                   // variant = kernel + 100 * QuickGELU form (common.h quick_gelu_fast_h4)
        a.gelu_mode = variant / 100;
        variant %= 100;
    }
    if (a.M <= 0 || a.N <= 0) return 0;
    if (a.K <= 0 || (a.K % BK) != 0 || (a.lda & 7) || (a.ldw & 7)) return -2;  // caller pads K to 64
    const int v5 = (variant == 0 || variant == 9 || variant == 7 || variant == 10) ? 8 : variant;
    if (a.im2col_R) {                                                           // only the v5 kernel gathers patch rows from the image; -4: shape not supported
        const int rc = launch_gemm_f16_v5(a, v5, s);
        return rc == -100 ? -4 : rc;
    }
    if (a.epi == EPI_SCALE_ARGMAX) {                                            // only the v5 kernel; -4: shape not supported
        const int rc = launch_gemm_f16_v5(a, v5, s);
        return rc == -100 ? -4 : rc;
    }
    if (a.epi == EPI_LN_BIAS || a.epi == EPI_LN_BIAS_QGELU || a.stats_out) {   // only the v5 kernel folds LayerNorm
        const int rc = launch_gemm_f16_v5(a, v5, s);
        return rc == -100 ? -4 : rc;
    }
#ifdef OVMR_EXPERIMENTS
    if (variant == 10) {                    // experiments/gemm_f16_w2.hip: 256 x 128 tiles, two workgroups per CU (timing experiment, r04)
        const int rc = launch_gemm_f16_w2(a, s);
        if (rc != -100) return rc;
        variant = 8;
    }
#endif
    if (variant == 7) variant = 8;          // A/B: variant 8 without the split-K kernel (the caller's LayerNorm-fold rule keys on 8 as well)
    else if (variant == 9 || (variant == 8 && gemm_f16_is_small(a.M, a.N))) {
        const int rc = launch_gemm_f16_small(a, s);
        if (rc != -100) return rc;
        if (variant == 9) variant = 8;
    }
    if (variant >= 1) {
        const int rc = launch_gemm_f16_v5(a, variant, s);
        if (rc != -100) return rc;   // -100: shape not supported -> the 128x128 kernel
    }
    switch (a.epi) {
        case EPI_NONE: return launch_t128<EPI_NONE>(a, s);
        case EPI_BIAS: return launch_t128<EPI_BIAS>(a, s);
        case EPI_BIAS_QGELU: return launch_t128<EPI_BIAS_QGELU>(a, s);
        case EPI_BIAS_RES: return launch_t128<EPI_BIAS_RES>(a, s);
        case EPI_PATCH: return launch_t128<EPI_PATCH>(a, s);
        case EPI_SCALE: return launch_t128<EPI_SCALE>(a, s);
    }
    return -3;
}
