// Experiment (round 4, experiment build only; variant 10 of ovmr_debug_gemm): can TWO resident workgroups per CU hide what the
// 256 x 256 tile kernel (gemm_f16_v5.hip: one 512-thread workgroup per CU, 128 KiB of LDS) leaves exposed -- the epilogue's
// residual loads / store burst / drain and the K loop's barrier waits?  Register file: a 256 x 256 fp32 accumulator tile is half
// of a CU's registers, so two resident workgroups mean two 256 x 128 tiles (the same accumulator footprint, 1.5x the L2 -> LDS
// bytes per FLOP).  This prototype keeps everything else simple:
//   * 256 x 128 x 32 tile, 4 waves (2 x 2), each wave 128 x 64 = 8 x 4 accumulator blocks of v_mfma_f32_16x16x32_f16;
//   * three 24 KiB stages of LDS (72 KiB: two workgroups per CU), 64-byte rows (K = 32 halves): a ds_read_b128 of 16 rows x 4
//     chunks is 1 KiB contiguous -- conflict-free without a swizzle -- and the LDS-DMA destination is lane-linear as it is;
//   * LDS-DMA from inline asm (scalar base + lane offset), counted vmcnt, ONE barrier per stage of 32 MFMAs;
//   * the shared element-wise epilogue (gemm_epi.h: 8-byte accesses straight from the accumulator layout, no LDS staging).
// Correct results (tests can run it); what it is for is the timing against variant 8 on the block's launch shapes.
#ifdef OVMR_EXPERIMENTS
#include "../common.h"
#include "../gemm_epi.h"

namespace {

typedef __attribute__((address_space(3))) void* lptr_w2;

__device__ __forceinline__ void glds16_w2(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

constexpr int W2_BM = 256, W2_BN = 128, W2_BK = 32;
constexpr int W2_STAGE = (W2_BM + W2_BN) * W2_BK;          // halves per stage: A rows, then W rows (64 B each)

template <int EPI>
__global__ __launch_bounds__(256, 2) void gemm_f16_w2(GemmArgs a, int tiles_m, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) half_t smem[];           // 3 stages = 72 KiB

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int fr = lane & 15, fg = lane >> 4;

    // block id -> tile: ids congruent mod 8 run on one XCD and get a contiguous range; inside it groups of 8 N tiles (1024 columns)
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, idx = bid >> 3, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    int tm, tn;
    {
        const int G = min(8, tiles_n), full = tiles_n / G, per = tiles_m * G, g = bid / per;
        if (g < full) { const int r = bid - g * per; tm = r / G; tn = g * G + (r - tm * G); }
        else { const int Gl = tiles_n - full * G, r = bid - full * per; tm = r / Gl; tn = full * G + (r - tm * Gl); }
    }
    const int m0 = tm * W2_BM, n0 = tn * W2_BN;
    const half_t* A = (const half_t*)a.A;
    const half_t* W = (const half_t*)a.W;

    // staging: a DMA instruction moves 16 rows x 64 B; lane -> (row l >> 2, 16-byte chunk l & 3).  Wave w stages A rows
    // 64 w .. 64 w + 63 (4 instructions) and W rows 32 w .. 32 w + 31 (2 instructions) of every stage.
    const int srow = lane >> 2, sch = lane & 3;
    unsigned oa[4], ob[2];
#pragma unroll
    for (int j = 0; j < 4; ++j) oa[j] = (unsigned)(((long)min(m0 + wave * 64 + j * 16 + srow, a.M - 1) * a.lda + sch * 8) * 2);
#pragma unroll
    for (int j = 0; j < 2; ++j) ob[j] = (unsigned)(((long)min(n0 + wave * 32 + j * 16 + srow, a.N - 1) * a.ldw + sch * 8) * 2);
    const unsigned lds_base = (unsigned)(uintptr_t)(lptr_w2)smem;
    auto stage = [&](int st, int kt) {
        const char* ab = (const char*)A + (long)kt * (W2_BK * 2);
        const char* wb = (const char*)W + (long)kt * (W2_BK * 2);
        const unsigned d = lds_base + 2u * (unsigned)(st * W2_STAGE);
#pragma unroll
        for (int j = 0; j < 4; ++j) glds16_w2(ab, oa[j], __builtin_amdgcn_readfirstlane(d + 2u * (unsigned)((wave * 64 + j * 16) * W2_BK)));
#pragma unroll
        for (int j = 0; j < 2; ++j) glds16_w2(wb, ob[j], __builtin_amdgcn_readfirstlane(d + 2u * (unsigned)((W2_BM + wave * 32 + j * 16) * W2_BK)));
    };

    float4_t acc[8][4];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

    const int nk = a.K / W2_BK;
    stage(0, 0);
    if (nk > 1) stage(1, 1);
    const int a_off = (wm * 128 + fr) * W2_BK + fg * 8, b_off = (W2_BM + wn * 64 + fr) * W2_BK + fg * 8;
    for (int kt = 0; kt < nk; ++kt) {
        // this wave's six DMA instructions of stage kt have landed (those of stage kt + 1 may stay in flight) ...
        if (kt + 1 < nk) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                      // ... and everybody's; everybody is done reading stage kt - 1
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 2 < nk) stage((kt + 2) % 3, kt + 2);      // into the buffer stage kt - 1 occupied
        const half_t* sA = smem + (kt % 3) * W2_STAGE;
        half8_t fa[8], fb[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) fb[t] = *(const half8_t*)(sA + b_off + t * 16 * W2_BK);
#pragma unroll
        for (int t = 0; t < 8; ++t) fa[t] = *(const half8_t*)(sA + a_off + t * 16 * W2_BK);
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j], fa[i], acc[i][j], 0, 0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            epilogue_store<EPI>(a, m0 + wm * 128 + i * 16 + fr, n0 + wn * 64 + j * 16 + fg * 4, acc[i][j]);
}

template <int EPI>
int launch_w2(const GemmArgs& a, hipStream_t s) {
    const int tiles_m = (a.M + W2_BM - 1) / W2_BM, tiles_n = (a.N + W2_BN - 1) / W2_BN;
    const size_t lds = (size_t)3 * W2_STAGE * sizeof(half_t);
    static bool attr_set = false;
    if (!attr_set) {
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)gemm_f16_w2<EPI>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_f16_w2<EPI>), dim3(tiles_m * tiles_n), dim3(256), lds, s, a, tiles_m, tiles_n);
    return (int)hipGetLastError();
}

}  // namespace

int launch_gemm_f16_w2(const GemmArgs& a, hipStream_t s) {
    if ((a.K % W2_BK) || (a.lda & 7) || (a.ldw & 7) || ((uintptr_t)a.A & 15) || ((uintptr_t)a.W & 15) ||
        (long)a.M * a.lda * 2 >= 0x7fffffffL || (long)a.N * a.ldw * 2 >= 0x7fffffffL || a.im2col_R || a.stats_out)
        return -100;
    switch (a.epi) {
        case EPI_NONE: return launch_w2<EPI_NONE>(a, s);
        case EPI_BIAS: return launch_w2<EPI_BIAS>(a, s);
        case EPI_BIAS_QGELU: return launch_w2<EPI_BIAS_QGELU>(a, s);
        case EPI_BIAS_RES: return launch_w2<EPI_BIAS_RES>(a, s);
        case EPI_SCALE: return launch_w2<EPI_SCALE>(a, s);
    }
    return -100;
}
#endif  // OVMR_EXPERIMENTS
