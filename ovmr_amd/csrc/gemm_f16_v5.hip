// fp16 MFMA GEMM, variants 6/8/9/10: the 256(128)x256x64 LDS-DMA kernel of variant 4 with
//   * EPILOGUE THROUGH LDS (all variants here).  The C^T accumulator layout gives each lane 4 consecutive
//     columns of 32 different (row, column-tile) pairs: 32 eight-byte stores per lane, each wave instruction
//     touching 16 rows x 32 B.  Measured 6.5 us per 256x256 tile (profiles/r01c).  Here h(acc + bias)
//     (+QuickGELU / scale) goes to a per-wave LDS tile and is read back row-major, so residual / positional
//     rows are LOADED and results STORED as 16 bytes per lane, 8 rows x 128 B (whole cache lines) per wave
//     instruction: 4x fewer line transactions, 2x fewer store instructions;
//   * OPT & 1: STAGGER.  After the per-K-tile barrier all 8 waves used to issue their LDS-DMA first (8 x ~100
//     cycles in which no MFMA issues on any SIMD, both waves of a SIMD being in lockstep).  Waves 4-7 now issue
//     theirs after the first k-step, so on every SIMD one wave computes while its partner issues loads;
//   * OPT & 2: BUFFER loads.  buffer_load_dwordx4 ... offen lds with the row offset in a 32-bit VGPR and the
//     K offset in an SGPR replaces global_load_lds + a 64-bit VALU add per instruction.
// (An L2-prefetch experiment -- one 4-byte LDS-DMA touch per thread two K-tiles ahead, counted vmcnt(1) --
//  was measured 5-20 % SLOWER than no prefetch and removed: profiles/r01c_gemm_bench.log, variants 5/7.)
#include "common.h"

#include <algorithm>
#include <cstdlib>

namespace {

constexpr int BK5 = 64, BN5 = 256;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int EPI, int MT, int OPT>
__global__ __launch_bounds__(512) void gemm_f16_v5_kernel(GemmArgs a, int tiles_m, int tiles_n) {
    constexpr int BM = MT * 32;
    constexpr int A_BYTES = BM * 128, STAGE = (BM + BN5) * 128;
    constexpr int AJ = BM / 64;
    constexpr int EP = 144;                            // epilogue LDS row pitch (64 halves + 16 B pad)
    constexpr bool STAGGER = OPT & 1, BUF = OPT & 2, PRIO = OPT & 4, NOLOAD = OPT & 16, NOREAD = OPT & 32;   // NOLOAD: timing-only ablation
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
    // Tile order inside the XCD's contiguous id range: N tiles in groups of G, M fastest-but-one.  A group's W panels
    // (G x 256 x K x 2 bytes, kept <= ~2.5 MB by the launcher) then stay in the XCD's 4 MiB L2 while the M panels
    // stream past; with plain row-major order every tile re-read its W panel from beyond L2 (FETCH_SIZE 8.6x the
    // algorithmic bytes on c_fc, profiles/r01d_pmc_gemm_v6.json).
    int tm, tn;
    {
        const int G = a.n_group > 0 ? a.n_group : tiles_n;
        const int full = tiles_n / G, per = tiles_m * G;
        const int g = bid / per;
        if (g < full) {
            const int r = bid - g * per;
            tm = r / G;
            tn = g * G + (r - tm * G);
        } else {
            const int Gl = tiles_n - full * G, r = bid - full * per;
            tm = r / Gl;
            tn = full * G + (r - tm * Gl);
        }
    }
    const int m0 = tm * BM, n0 = tn * BN5;

    const half_t* A = (const half_t*)a.A;
    const half_t* W = (const half_t*)a.W;

    // staging: lane -> (row within the 8-row group, destination slot); source chunk = slot ^ row
    const int srow = lane >> 3, slot = lane & 7;
    const int schunk = (slot ^ srow) * 8;
    unsigned oa[AJ], ob[4];                            // byte offsets of this lane's source rows
    const int nkt = a.K / BK5;
    // row-major: row * ld * 2 + chunk, K-tile step 128 B.  blocked [rows/128][K/64][128][64]: one (row block, K-tile)
    // is 16 KiB contiguous, so every LDS-DMA instruction reads 1 KiB of consecutive addresses
    auto src_off = [&](int row, int ld, int blocked) -> unsigned {
        return blocked ? (unsigned)(((long)(row >> 7) * nkt) * 16384 + (row & 127) * 128 + schunk * 2)
                       : (unsigned)(((long)row * ld + schunk) * 2);
    };
    const int a_step = a.a_blocked ? 16384 : 128, w_step = a.w_blocked ? 16384 : 128;
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const int r = m0 + wave * (BM / 8) + j * 8 + srow;
        oa[j] = src_off(a.a_blocked ? r : min(r, a.M - 1), a.lda, a.a_blocked);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = n0 + wave * 32 + j * 8 + srow;
        ob[j] = src_off(a.w_blocked ? r : min(r, a.N - 1), a.ldw, a.w_blocked);
    }
    const int ldsA_w = wave * (BM / 8) * 128;
    const int ldsB_w = A_BYTES + wave * 32 * 128;
#if defined(__HIP_DEVICE_COMPILE__)   // the buffer-resource type does not exist in the host pass of this TU
    __amdgpu_buffer_rsrc_t ra = __builtin_amdgcn_make_buffer_rsrc((void*)A, 0, 0x7fffffff, 0x00020000);
    __amdgpu_buffer_rsrc_t rb = __builtin_amdgcn_make_buffer_rsrc((void*)W, 0, 0x7fffffff, 0x00020000);
#endif

    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * STAGE;
        if (BUF) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
            for (int j = 0; j < AJ; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(ra, (lptr_t)(base + ldsA_w + j * 1024), 16, oa[j], kt * a_step, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rb, (lptr_t)(base + ldsB_w + j * 1024), 16, ob[j], kt * w_step, 0, 0);
#endif
        } else {
#pragma unroll
            for (int j = 0; j < AJ; ++j)
                __builtin_amdgcn_global_load_lds((gptr_t)((const char*)A + oa[j] + (long)kt * a_step), (lptr_t)(base + ldsA_w + j * 1024), 16, 0, 0);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                __builtin_amdgcn_global_load_lds((gptr_t)((const char*)W + ob[j] + (long)kt * w_step), (lptr_t)(base + ldsB_w + j * 1024), 16, 0, 0);
        }
    };

    float4_t acc[MT][4];
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fg = lane >> 4;
    const int a_row_off = (wm * (BM / 2) + fr) * 128;
    const int b_row_off = A_BYTES + (wn * 64 + fr) * 128;
    const int ch0 = ((fg) ^ (fr & 7)) << 4, ch1 = ((4 + fg) ^ (fr & 7)) << 4;
    const int nk = nkt;
    const bool late = STAGGER && wave >= 4;            // wave-uniform (readfirstlane above)

    half8_t nr_a[2][NOREAD ? MT : 1], nr_b[2][NOREAD ? 4 : 1];
    stage(0, 0);
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                  // tile kt landed for every wave; buffer (kt+1)&1 is free
        __builtin_amdgcn_sched_barrier(0);
        if (!late && kt + 1 < nk && !NOLOAD) stage((kt + 1) & 1, kt + 1);

        const char* cur = smem + (kt & 1) * STAGE;
        if (NOREAD) {   // timing-only ablation: LDS-DMA + MFMA, fragments read once (kt == 0) and reused
            static_assert(!NOREAD || MT <= 8, "");
            if (kt == 0) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
                    for (int t = 0; t < 4; ++t) nr_b[ks][t] = *(const half8_t*)(cur + b_row_off + t * 2048 + (ks ? ch1 : ch0));
#pragma unroll
                    for (int t = 0; t < MT; ++t) nr_a[ks][t] = *(const half8_t*)(cur + a_row_off + t * 2048 + (ks ? ch1 : ch0));
                }
            }
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < MT; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(nr_b[ks][j], nr_a[ks][i], acc[i][j], 0, 0, 0);
            continue;
        }
        half8_t fb[2][4], fa[3];
        // software-pipelined fragment reads: A fragment of step t+2 and the B fragments of the next k-step are
        // issued before the MFMAs of step t; sched_group_barrier pins that order for the machine scheduler
#pragma unroll
        for (int t = 0; t < 4; ++t) fb[0][t] = *(const half8_t*)(cur + b_row_off + t * 2048 + ch0);
        fa[0] = *(const half8_t*)(cur + a_row_off + ch0);
        fa[1] = *(const half8_t*)(cur + a_row_off + 2048 + ch0);
        if (PRIO) __builtin_amdgcn_s_setprio(1);
#pragma unroll
        for (int st = 0; st < MT; ++st) {
            const int nx = st + 2;
            fa[nx % 3] = *(const half8_t*)(cur + a_row_off + (nx % MT) * 2048 + (nx / MT ? ch1 : ch0));
            if (st == 1) {
#pragma unroll
                for (int t = 0; t < 4; ++t) fb[1][t] = *(const half8_t*)(cur + b_row_off + t * 2048 + ch1);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[st][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[0][j], fa[st % 3], acc[st][j], 0, 0, 0);
        }
        __builtin_amdgcn_sched_group_barrier(0x100, 6, 0);
#pragma unroll
        for (int st = 0; st < MT; ++st) {
            if (st == 1) __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
            else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
        if (late && kt + 1 < nk && !NOLOAD) stage((kt + 1) & 1, kt + 1);
#pragma unroll
        for (int st = MT; st < 2 * MT; ++st) {
            const int nx = st + 2;
            if (nx < 2 * MT) fa[nx % 3] = *(const half8_t*)(cur + a_row_off + (nx % MT) * 2048 + ch1);
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[st - MT][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[1][j], fa[st % 3], acc[st - MT][j], 0, 0, 0);
        }
#pragma unroll
        for (int st = MT; st < 2 * MT; ++st) {
            if (st + 2 < 2 * MT) __builtin_amdgcn_sched_group_barrier(0x100, 1, 1);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 1);
        }
        if (PRIO) __builtin_amdgcn_s_setprio(0);
    }

    // ---------------------------------------------------------------- epilogue through LDS
    half_t* C = (half_t*)a.C;
    char* et = smem + wave * (64 * EP);                 // this wave's 64-row x 64-column staging tile
    const int er = lane >> 3, ec = (lane & 7) * 8;     // phase 2: row within an 8-row group, first column
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
        __syncthreads();                                // main-loop reads / previous half's reads are done
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
            if (m < a.M && nn < a.N) {                  // N % 8 == 0 is guaranteed by the launcher
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

template <int EPI, int MT, int OPT>
int launch_v5(const GemmArgs& a, hipStream_t s) {
    constexpr int BM = MT * 32;
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN5 - 1) / BN5;
    const size_t lds = (size_t)2 * (BM + BN5) * 128;
    static bool attr_set = false;
    if (!attr_set) {
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)gemm_f16_v5_kernel<EPI, MT, OPT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    GemmArgs b = a;
    if (b.n_group <= 0) {
        static int force = -1;
        if (force < 0) { const char* e = getenv("OVMR_N_GROUP"); force = e ? atoi(e) : 0; }
        // measured (profiles/r01e_gemm_experiments.md): groups of 4-6 raise the L2 hit rate of qkv / c_fc from 65-68 %
        // to 72-73 % but move the run time by < 2 %, and hurt c_proj; the default therefore stays row-major (G = all)
        b.n_group = force > 0 ? std::min(force, tiles_n) : tiles_n;
    }
    hipLaunchKernelGGL((gemm_f16_v5_kernel<EPI, MT, OPT>), dim3(tiles_m * tiles_n), dim3(512), lds, s, b, tiles_m, tiles_n);
    return (int)hipGetLastError();
}

int g_force_mt = -1;   // debug: OVMR_FORCE_MT=4|8 pins the M tile (tools/gemm_bench.py)

template <int EPI, int OPT>
int pick_v5(const GemmArgs& a, hipStream_t s) {
    if (g_force_mt < 0) {
        const char* e = getenv("OVMR_FORCE_MT");
        g_force_mt = e ? atoi(e) : 0;
    }
    auto eff = [&](int bm) {
        const double t = (double)((a.M + bm - 1) / bm) * ((a.N + BN5 - 1) / BN5);
        return t / (ceil(t / 256.0) * 256.0);
    };
    const bool big = g_force_mt ? g_force_mt == 8 : eff(256) + 0.08 >= eff(128);
    return big ? launch_v5<EPI, 8, OPT>(a, s) : launch_v5<EPI, 4, OPT>(a, s);
}

template <int OPT>
int dispatch_v5(const GemmArgs& a, hipStream_t s) {
    switch (a.epi) {
        case EPI_NONE: return pick_v5<EPI_NONE, OPT>(a, s);
        case EPI_BIAS: return pick_v5<EPI_BIAS, OPT>(a, s);
        case EPI_BIAS_QGELU: return pick_v5<EPI_BIAS_QGELU, OPT>(a, s);
        case EPI_BIAS_RES: return pick_v5<EPI_BIAS_RES, OPT>(a, s);
        case EPI_PATCH: return pick_v5<EPI_PATCH, OPT>(a, s);
        case EPI_SCALE: return pick_v5<EPI_SCALE, OPT>(a, s);
    }
    return -3;
}

}  // namespace

// variant 6: LDS epilogue only; 8: + stagger; 9: + buffer loads; 10: + both
int launch_gemm_f16_v5(const GemmArgs& a, int variant, hipStream_t s) {
    if (a.M < 256 || a.N < 128 || (a.N & 7) || (a.ldc & 7) || (a.epi == EPI_BIAS_RES && (a.ldres & 7)) ||
        ((uintptr_t)a.C & 15) || (a.epi == EPI_BIAS_RES && ((uintptr_t)a.res & 15)) ||
        (long)a.M * a.lda * 2 >= 0x7fffffffL || (long)a.N * a.ldw * 2 >= 0x7fffffffL)
        return -100;
    switch (variant) {
        case 8: return dispatch_v5<1>(a, s);
        case 9: return dispatch_v5<2>(a, s);
        case 10: return dispatch_v5<3>(a, s);
        case 14: return pick_v5<EPI_BIAS, 16>(a, s);       // timing-only: no loads after the first K-tile
        case 15: return pick_v5<EPI_BIAS, 20>(a, s);       // timing-only: no loads + setprio
        case 16: return dispatch_v5<4>(a, s);               // setprio(1) around the MFMA stream
        case 17: return pick_v5<EPI_BIAS, 32>(a, s);       // timing-only: loads + MFMA, fragment reads always from buffer 0
        default: return dispatch_v5<0>(a, s);
    }
}
