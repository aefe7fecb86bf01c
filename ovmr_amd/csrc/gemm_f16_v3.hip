// fp16 MFMA GEMM, variants 2/3: LDS-DMA ring with COUNTED vmcnt (loads stay in flight across barriers).
//
// Variant 1 drains its loads (vmcnt(0)) at every K-tile, so only one K-tile is ever in flight and the
// L2/HBM latency (~2 us under load) is exposed against ~1 us of MFMA work per K-tile.  Here:
//   * K-stage = 32 (one MFMA 16x16x32 k-step), NS ring slots in LDS, prefetch distance P = NS - 1:
//     while stage s is consumed, stages s+1 .. s+P are in flight;
//   * s_waitcnt vmcnt(N) with N = (stages allowed in flight) x (glds per wave per stage) -- never 0 in the
//     steady state -- followed by a raw s_barrier (a __syncthreads() would drain the LDS-DMA queue);
//   * the slot being refilled at the top of iteration s is the one consumed in iteration s-1; every wave
//     waits lgkmcnt(0) before the barrier, so its reads of that slot have returned (WAR safe);
//   * 64-byte LDS rows, 16-byte chunk c of row r stored in slot c ^ ((-(r>>2)) & 3): every ds_read_b128
//     lane group touches 16 distinct 16-byte slots (derived in DESIGN.md), swizzle applied on the per-lane
//     SOURCE address of the LDS-DMA and on the read address.
// Configurations:
//   variant 2: 256x256 tile, 8 waves (2x4), 4 slots x 32 KiB = 128 KiB, 1 workgroup per CU;
//   variant 3: 128x256 tile, 4 waves (1x4), 3 slots x 24 KiB =  72 KiB, 2 workgroups per CU so that one
//              workgroup's prologue / epilogue (bias, QuickGELU, stores) overlaps the other's MFMA loop.
#include "common.h"
#include "gemm_epi.h"

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int EPI, int MT, int WM, int WN, int NS>
__global__ __launch_bounds__(WM * WN * 64) void gemm_f16_ring(GemmArgs a, int tiles_m, int tiles_n) {
    constexpr int NW = WM * WN, BM = WM * MT * 16, BN = WN * 64;
    constexpr int A_BYTES = BM * 64, STAGE = (BM + BN) * 64;
    constexpr int AJ = BM / 16 / NW, BJ = BN / 16 / NW, NG = AJ + BJ;   // glds per wave per stage
    constexpr int P = NS - 1;
    static_assert(AJ * 16 * NW == BM && BJ * 16 * NW == BN, "tile must split evenly over the waves");
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;

    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, idx = bid >> 3, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN;

    const half_t* A = (const half_t*)a.A;
    const half_t* W = (const half_t*)a.W;

    // staging: one glds = 16 rows x 64 B; lane -> row lane>>2, destination slot lane&3
    const int srow = lane >> 2;
    const int schunk = ((lane & 3) ^ ((-(lane >> 4)) & 3)) * 8;
    const half_t* ga[AJ];
    const half_t* gb[BJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j)
        ga[j] = A + (long)min(m0 + (wave * AJ + j) * 16 + srow, a.M - 1) * a.lda + schunk;
#pragma unroll
    for (int j = 0; j < BJ; ++j)
        gb[j] = W + (long)min(n0 + (wave * BJ + j) * 16 + srow, a.N - 1) * a.ldw + schunk;
    const int ldsA_w = wave * AJ * 1024;
    const int ldsB_w = A_BYTES + wave * BJ * 1024;

    auto stage = [&](int s) {
        char* base = smem + (s % NS) * STAGE;
#pragma unroll
        for (int j = 0; j < AJ; ++j)
            __builtin_amdgcn_global_load_lds((gptr_t)(ga[j] + (long)s * 32), (lptr_t)(base + ldsA_w + j * 1024), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < BJ; ++j)
            __builtin_amdgcn_global_load_lds((gptr_t)(gb[j] + (long)s * 32), (lptr_t)(base + ldsB_w + j * 1024), 16, 0, 0);
    };

    float4_t acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fg = lane >> 4;
    const int ch = (fg ^ ((-(fr >> 2)) & 3)) << 4;
    const int a_off = (wm * MT * 16 + fr) * 64 + ch;
    const int b_off = A_BYTES + (wn * 64 + fr) * 64 + ch;
    const int nk = a.K / 32;

#pragma unroll
    for (int s = 0; s < P; ++s)
        if (s < nk) stage(s);

    for (int s = 0; s < nk; ++s) {
        // stage s must have landed: stages s+1 .. min(s+P-1, nk-1) may stay in flight
        const int ahead = min(P - 1, nk - 1 - s);
        if (P >= 3 && ahead >= 2) wait_vmcnt<2 * NG>();
        else if (ahead == 1) wait_vmcnt<NG>();
        else wait_vmcnt<0>();
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (s + P < nk) stage(s + P);                 // refills the slot consumed in iteration s-1

        const char* cur = smem + (s % NS) * STAGE;
        half8_t fb[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) fb[t] = *(const half8_t*)(cur + b_off + t * 1024);
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            const half8_t fa = *(const half8_t*)(cur + a_off + i * 1024);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j], fa, acc[i][j], 0, 0, 0);
        }
    }

#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            epilogue_store<EPI>(a, m0 + wm * MT * 16 + i * 16 + fr, n0 + wn * 64 + j * 16 + fg * 4, acc[i][j]);
}

template <int EPI, int MT, int WM, int WN, int NS>
int launch_ring(const GemmArgs& a, hipStream_t s) {
    constexpr int BM = WM * MT * 16, BN = WN * 64;
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN - 1) / BN;
    const size_t lds = (size_t)NS * (BM + BN) * 64;
    static bool attr_set = false;
    if (!attr_set) {
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)gemm_f16_ring<EPI, MT, WM, WN, NS>,
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_f16_ring<EPI, MT, WM, WN, NS>), dim3(tiles_m * tiles_n), dim3(WM * WN * 64), lds, s, a, tiles_m, tiles_n);
    return (int)hipGetLastError();
}

template <int EPI>
int pick_ring(const GemmArgs& a, int variant, hipStream_t s) {
    if (variant == 2) return launch_ring<EPI, 8, 2, 4, 4>(a, s);     // 256x256, 8 waves, 128 KiB
    return launch_ring<EPI, 8, 1, 4, 3>(a, s);                        // 128x256, 4 waves, 72 KiB, 2 WG/CU
}

}  // namespace

int launch_gemm_f16_v3(const GemmArgs& a, int variant, hipStream_t s) {
    if (a.M < 256 || a.N < 128) return -100;
    switch (a.epi) {
        case EPI_NONE: return pick_ring<EPI_NONE>(a, variant, s);
        case EPI_BIAS: return pick_ring<EPI_BIAS>(a, variant, s);
        case EPI_BIAS_QGELU: return pick_ring<EPI_BIAS_QGELU>(a, variant, s);
        case EPI_BIAS_RES: return pick_ring<EPI_BIAS_RES>(a, variant, s);
        case EPI_PATCH: return pick_ring<EPI_PATCH>(a, variant, s);
        case EPI_SCALE: return pick_ring<EPI_SCALE>(a, variant, s);
    }
    return -3;
}
