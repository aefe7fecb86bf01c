// fp16 MFMA GEMM, variant 11: 4-slot LDS-DMA ring (K-stage 32) with REGISTER double-buffered fragments.
//
// What the measurements of variants 1-10 say: per 256x256x64 of work the loads alone need
// ~1.9 us per CU when every CU streams (~10 TB/s of L2->LDS traffic chip wide, the same rate the 256-square
// 8-phase template of the CDNA guide runs at), the MFMAs alone 1.0-1.7 us; a kernel that drains its loads at every
// K-tile pays the sum of parts of both.  This variant keeps the load queue busy all the time and takes the LDS
// read latency off the MFMA path:
//   * ring of 4 slots x 32 KiB (BM = 256; 24 KiB at BM = 128), K-stage = 32.  In iteration s the MFMAs consume
//     the fragments of stage s FROM REGISTERS, the fragments of stage s+1 are read from LDS into the other register
//     set in between those MFMAs (order pinned with sched_group_barrier), and stages s+2, s+3, s+4 are in flight;
//   * one counted s_waitcnt vmcnt(2 x glds-per-stage) + lgkmcnt(0) + raw s_barrier per stage (never vmcnt(0) in the
//     steady state).  The slot refilled in iteration s (stage s+4) is the one whose fragments were read in
//     iteration s-1 by every wave before it reached this barrier (WAR safe);
//   * 64-byte LDS rows, 16-byte chunk c of row r in slot c ^ ((-(r>>2)) & 3), swizzle on the LDS-DMA source
//     address and on the ds_read_b128 address;
//   * epilogue through LDS as in variant 6 (16-byte coalesced residual loads / stores).
#include "common.h"

#include <algorithm>
#include <type_traits>

namespace {

constexpr int BN7 = 256, NS7 = 4;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int N>
__device__ __forceinline__ void wait_vm() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

template <int EPI, int MT>
__global__ __launch_bounds__(512) void gemm_f16_v7_kernel(GemmArgs a, int tiles_m, int tiles_n) {
    constexpr int BM = MT * 32;
    constexpr int A_BYTES = BM * 64, STAGE = (BM + BN7) * 64;
    constexpr int AJ = BM / 128, BJ = 2, NG = AJ + BJ;          // LDS-DMA instructions per wave per stage (16 rows each)
    constexpr int EP = 144;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, idx = bid >> 3, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
    const int tm = bid / tiles_n, tn = bid - tm * tiles_n;
    const int m0 = tm * BM, n0 = tn * BN7;

    const char* A = (const char*)a.A;
    const char* W = (const char*)a.W;

    // staging: one LDS-DMA instruction = 16 rows x 64 B; lane -> row lane>>2, destination slot lane&3
    const int srow = lane >> 2;
    const int schunk = ((lane & 3) ^ ((-(lane >> 4)) & 3)) * 16;                 // source byte offset inside the 64-B row
    unsigned oa[AJ], ob[BJ];
#pragma unroll
    for (int j = 0; j < AJ; ++j)
        oa[j] = (unsigned)((long)min(m0 + (wave * AJ + j) * 16 + srow, a.M - 1) * a.lda * 2 + schunk);
#pragma unroll
    for (int j = 0; j < BJ; ++j)
        ob[j] = (unsigned)((long)min(n0 + (wave * BJ + j) * 16 + srow, a.N - 1) * a.ldw * 2 + schunk);
    const int ldsA_w = wave * AJ * 1024;
    const int ldsB_w = A_BYTES + wave * BJ * 1024;

    auto stage = [&](int s) {
        char* base = smem + (s & (NS7 - 1)) * STAGE;
#pragma unroll
        for (int j = 0; j < AJ; ++j)
            __builtin_amdgcn_global_load_lds((gptr_t)(A + oa[j] + (long)s * 64), (lptr_t)(base + ldsA_w + j * 1024), 16, 0, 0);
#pragma unroll
        for (int j = 0; j < BJ; ++j)
            __builtin_amdgcn_global_load_lds((gptr_t)(W + ob[j] + (long)s * 64), (lptr_t)(base + ldsB_w + j * 1024), 16, 0, 0);
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
    const int nk = a.K / 32;                                   // even (K % 64 == 0)

    half8_t fa[2][MT], fb[2][4];
    auto read_frags = [&](auto ph, int s) {
        constexpr int P = decltype(ph)::value;
        const char* cur = smem + (s & (NS7 - 1)) * STAGE;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            fa[P][i] = *(const half8_t*)(cur + a_off + i * 1024);
            if (i < 4) fb[P][i] = *(const half8_t*)(cur + b_off + i * 1024);
        }
    };
    // wait until stage `need` has landed (for this wave's loads) given that stages up to `issued` were issued
    auto wait_stage = [&](int need, int issued) {
        const int ahead = issued - need;                       // stages allowed to stay in flight
        if (ahead >= 3) wait_vm<3 * NG>();
        else if (ahead == 2) wait_vm<2 * NG>();
        else if (ahead == 1) wait_vm<NG>();
        else wait_vm<0>();
    };
    auto sync = [&]() {
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
    };

    // iteration s (register set P holds stage s): read stage s+1 into set P^1 between the MFMAs of stage s
    auto iter = [&](auto ph, int s) {
        constexpr int P = decltype(ph)::value;
        wait_stage(s + 1, min(s + 3, nk - 1));
        sync();                                                // stage s+1 visible; slot of stage s is free
        if (s + 4 < nk) stage(s + 4);
        const char* nxt = smem + ((s + 1) & (NS7 - 1)) * STAGE;
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            fa[P ^ 1][i] = *(const half8_t*)(nxt + a_off + i * 1024);
            if (i < 4) fb[P ^ 1][i] = *(const half8_t*)(nxt + b_off + i * 1024);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[P][j], fa[P][i], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            if (i < 4) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
    };

#pragma unroll
    for (int s = 0; s < 4; ++s)
        if (s < nk) stage(s);
    wait_stage(0, min(3, nk - 1));
    sync();
    read_frags(std::integral_constant<int, 0>{}, 0);
    for (int s = 0; s + 2 < nk; s += 2) {
        iter(std::integral_constant<int, 0>{}, s);
        iter(std::integral_constant<int, 1>{}, s + 1);
    }
    iter(std::integral_constant<int, 0>{}, nk - 2);            // reads the last stage (nk-1) into set 1
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[1][j], fa[1][i], acc[i][j], 0, 0, 0);

    // ---------------------------------------------------------------- epilogue through LDS (as variant 6)
    half_t* C = (half_t*)a.C;
    char* et = smem + wave * (64 * EP);
    const int er = lane >> 3, ec = (lane & 7) * 8;
    half4_t bias4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int n = n0 + wn * 64 + j * 16 + fg * 4;
        bias4[j] = (half4_t){(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
        if ((EPI == EPI_BIAS || EPI == EPI_BIAS_QGELU || EPI == EPI_BIAS_RES) && n + 3 < a.N)
            bias4[j] = *(const half4_t*)((const half_t*)a.bias + n);
    }
#pragma unroll
    for (int h = 0; h < MT / 4; ++h) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float4_t v = acc[h * 4 + i][j];
                half4_t o;
                if (EPI == EPI_BIAS || EPI == EPI_BIAS_QGELU || EPI == EPI_BIAS_RES) {
                    const half4_t b4 = bias4[j];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float x = (float)(half_t)(v[r] + (float)b4[r]);
                        if (EPI == EPI_BIAS_QGELU) x = quick_gelu_h(x);
                        o[r] = (half_t)x;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        float x = (float)(half_t)v[r];
                        if (EPI == EPI_SCALE) x *= a.scale;
                        o[r] = (half_t)x;
                    }
                }
                *(half4_t*)(et + (i * 16 + fr) * EP + (j * 16 + fg * 4) * 2) = o;
            }
        __syncthreads();
        const int nn = n0 + wn * 64 + ec;
#pragma unroll
        for (int it = 0; it < 8; ++it) {
            const int row = it * 8 + er;
            const int m = m0 + wm * (BM / 2) + h * 64 + row;
            if (m < a.M && nn < a.N) {
                half8_t v = *(const half8_t*)(et + row * EP + ec * 2);
                long crow = m;
                if (EPI == EPI_BIAS_RES) {
                    half8_t r8 = *(const half8_t*)((const half_t*)a.res + (long)m * a.ldres + nn);
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = (half_t)((float)v[k] + (float)r8[k]);
                }
                if (EPI == EPI_PATCH) {
                    const int b = m / a.rows_in, p = m - b * a.rows_in;
                    crow = (long)b * a.rows_out + 1 + p;
                    half8_t p8 = *(const half8_t*)((const half_t*)a.pos + (long)(1 + p) * a.N + nn);
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = (half_t)((float)v[k] + (float)p8[k]);
                }
                *(half8_t*)(C + crow * a.ldc + nn) = v;
            }
        }
    }
}

template <int EPI, int MT>
int launch_v7(const GemmArgs& a, hipStream_t s) {
    constexpr int BM = MT * 32;
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN7 - 1) / BN7;
    const size_t lds = std::max((size_t)NS7 * (BM + BN7) * 64, (size_t)8 * 64 * 144);
    static bool attr_set = false;
    if (!attr_set) {
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)gemm_f16_v7_kernel<EPI, MT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    hipLaunchKernelGGL((gemm_f16_v7_kernel<EPI, MT>), dim3(tiles_m * tiles_n), dim3(512), lds, s, a, tiles_m, tiles_n);
    return (int)hipGetLastError();
}

template <int EPI>
int pick_v7(const GemmArgs& a, hipStream_t s) {
    auto eff = [&](int bm) {
        const double t = (double)((a.M + bm - 1) / bm) * ((a.N + BN7 - 1) / BN7);
        return t / (ceil(t / 256.0) * 256.0);
    };
    return eff(256) + 0.08 >= eff(128) ? launch_v7<EPI, 8>(a, s) : launch_v7<EPI, 4>(a, s);
}

}  // namespace

int launch_gemm_f16_v7(const GemmArgs& a, hipStream_t s) {
    if (a.M < 256 || a.N < 128 || (a.N & 7) || (a.ldc & 7) || (a.epi == EPI_BIAS_RES && (a.ldres & 7)) ||
        ((uintptr_t)a.C & 15) || (a.epi == EPI_BIAS_RES && ((uintptr_t)a.res & 15)) || a.K < 128 ||
        (long)a.M * a.lda * 2 >= 0x7fffffffL || (long)a.N * a.ldw * 2 >= 0x7fffffffL)
        return -100;
    switch (a.epi) {
        case EPI_NONE: return pick_v7<EPI_NONE>(a, s);
        case EPI_BIAS: return pick_v7<EPI_BIAS>(a, s);
        case EPI_BIAS_QGELU: return pick_v7<EPI_BIAS_QGELU>(a, s);
        case EPI_BIAS_RES: return pick_v7<EPI_BIAS_RES>(a, s);
        case EPI_PATCH: return pick_v7<EPI_PATCH>(a, s);
        case EPI_SCALE: return pick_v7<EPI_SCALE>(a, s);
    }
    return -3;
}
