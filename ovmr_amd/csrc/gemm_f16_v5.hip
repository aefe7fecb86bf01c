// fp16 MFMA GEMM, variants 6 and 8 (the default): 256(128)x256x64 tiles, both operands streamed into LDS by LDS-DMA
// (global_load_lds, 16 B per lane, XOR-swizzled 128-byte rows), 8 waves as 2 x 4.  K loops: variant 8 = the ping-pong loop
// (four half-tiles per K-tile, counted vmcnt, the two wave rows one barrier apart: see OPT & 16 below; r02: qkv 369 -> 312,
// c_fc 522 -> 461, c_proj 462 -> 387, out_proj 149 -> 132 us at batch 512); variant 6 = double buffer with a drain per K-tile
// and software-pipelined fragment reads (also the fallback of 8 for an odd number of K-tiles and for 128-row tiles).
//
// Epilogue (measured with tools/gemm_bench.py variants 6 / 18 / 19, profiles/r01g_gemm_epilogue.md: with K = 768 the
// epilogue was 24-44 % of the kernel, the K loop alone runs at 1.1-1.2 PFLOP/s):
//   * EVERYTHING THE EPILOGUE READS IS STAGED IN LDS AT KERNEL START -- bias (or the LayerNorm-fold column constants) of the
//     tile's 256 columns and, for the LN-folding epilogues, rstd / -rstd*mean of its rows.  Their global-load latency hides
//     under the first K-tile's; loading them after the K loop put ~1-2 us of exposed latency in front of every tile's stores;
//   * results go THROUGH LDS: the C^T accumulator layout gives each lane 4 consecutive columns of 32 (row, column-tile)
//     pairs; h(acc + bias) (+QuickGELU / scale / LN fold) is written to a per-wave LDS tile and read back row-major, so
//     residual / positional rows are LOADED and results STORED 16 bytes per lane, 8 rows x 128 B per wave instruction;
//   * residual rows (EPI_BIAS_RES) are requested before the accumulators are converted, not in front of each store;
//   * large outputs are stored with the NONTEMPORAL hint (template bit 512): C then stops evicting the A / W panels the
//     other tiles of the XCD are streaming out of its 4 MiB L2 (qkv 404 -> 368 us, c_fc 587 -> 556 us);
//   * EPI_BIAS_RES can emit per-row partial (sum, sum of squares) for the LayerNorm that follows (common.h);
//   * ONE workgroup barrier in the epilogue (the staging tiles are wave-private; the barriers that used to separate the
//     phases also drained the first half's global stores), packed fp32 / fp16 VALU forms, interior tiles without bounds
//     compares.
// K loop variants chosen per launch (launch_v5): the A stream with the nontemporal policy for the residual projections
// (OPT & 1), and the iteration boundary moved inside the MFMA stream (OPT & 4) for K >= 2048 and the LN-folding launches.
// Measured and removed (profiles/r01e_gemm_experiments.md, r01g_gemm_epilogue.md): L2 prefetch touches (every workgroup, and
// designated prefetcher workgroups), staggered LDS-DMA issue, buffer_load ... lds, s_setprio around the MFMA stream, a
// persistent one-workgroup-per-CU tile loop with the next tile's first K-tile prefetched under the epilogue, W fragments
// straight from global memory, tile-contiguous C stores, residual as accumulator start value, x ping-pong buffers.
#include "common.h"

#include <algorithm>
#include <cstdlib>

namespace {

// A/B switches of tools/gemm_bench.py exist only in experiment builds (python -m ovmr_amd.build --experiments ->
// libovmr_hip_exp.so); the product library reads no environment variable and carries no timing-only kernel.
#ifdef OVMR_EXPERIMENTS
inline int exp_env(const char* name) { const char* e = getenv(name); return e ? atoi(e) : 0; }
#else
inline int exp_env(const char*) { return 0; }
#endif

constexpr int BK5 = 64, BN5 = 256;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int CTRL>
__device__ __forceinline__ float dpp_f32(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, true));
}

// OPT bits: 1 / 2 nontemporal LDS-DMA for the A / W operand, 4 K loop with the iteration boundary inside the MFMA stream,
// 16 the ping-pong K loop (256-row tiles), 32 pairwise QuickGELU (experiment A/B), 64 NOSTORE / 128 NOEPI (timing-only ablations, experiment builds only),
// 512 NT (nontemporal C stores)
template <int EPI, int MT, int OPT>
__global__ __launch_bounds__(512) void gemm_f16_v5_kernel(GemmArgs a, int tiles_m, int tiles_n) {
    constexpr int BM = MT * 32;
    constexpr int A_BYTES = BM * 128, STAGE = (BM + BN5) * 128;
    constexpr int AJ = BM / 64;
    constexpr bool NOSTORE = OPT & 64, NOEPI = OPT & 128, NT = OPT & 512;
    constexpr int EP = 128;                            // epilogue staging tile: 128-byte rows, swizzled 8-byte units (below)
    constexpr bool GFAST = (OPT & 2048) != 0;          // QuickGELU: the one-rounding fp32 form (common.h quick_gelu_f32x2) instead of the reference's three fp16 rounding points
    constexpr bool LNF = EPI == EPI_LN_BIAS || EPI == EPI_LN_BIAS_QGELU;   // LayerNorm folded into this GEMM (common.h)
    constexpr bool HAS_BIAS = EPI == EPI_BIAS || EPI == EPI_BIAS_QGELU || EPI == EPI_BIAS_RES;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    // behind the two K buffers: column constants [2][256] fp32, row constants [2][BM] fp32
    float* col_c = (float*)(smem + 2 * STAGE);
    float* row_c = col_c + 2 * BN5;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 2, wn = wave & 3;

    // block id -> tile.  XCD-aware: ids congruent mod 8 run on one XCD and get a contiguous range of tiles; inside the
    // range N tiles come in groups of G with M fastest-but-one (a.n_group, see launch_v5).
    const int nwg = tiles_m * tiles_n;
    int bid = blockIdx.x;
    {
        const int xcd = bid & 7, idx = bid >> 3, q = nwg >> 3, r = nwg & 7;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
    }
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

    const int nk = a.K / BK5;
    const int a_step = a.a_blocked ? 16384 : 128, w_step = a.w_blocked ? 16384 : 128;
    const half_t* A = (const half_t*)a.A;
    const half_t* W = (const half_t*)a.W;

    // staging: lane -> (row within the 8-row group, destination slot); source chunk = slot ^ row
    const int srow = lane >> 3, slot = lane & 7;
    const int schunk = (slot ^ srow) * 8;
    unsigned oa[AJ], ob[4];                            // byte offsets of this lane's source rows
    // row-major: row * ld * 2 + chunk, K-tile step 128 B.  blocked [rows/128][K/64][128][64]: one (row block, K-tile)
    // is 16 KiB contiguous, so every LDS-DMA instruction reads 1 KiB of consecutive addresses
    auto src_off = [&](int row, int ld, int blocked) -> unsigned {
        return blocked ? (unsigned)(((long)(row >> 7) * nk) * 16384 + (row & 127) * 128 + schunk * 2)
                       : (unsigned)(((long)row * ld + schunk) * 2);
    };
    // EPI_PATCH with a.im2col_R: the A operand is the image tensor [B, 3, R, R] (fp16) and row r = patch (b, gy, gx) of the 16 x 16
    // grid cells.  K-tile kt holds k = kt*64 .. +63 = channel kt >> 2, pixel rows 4 (kt & 3) .. +3 of the patch, 16 pixels each: the
    // 16-byte chunk c8 of a row is pixel row c8 >> 1, pixels 8 (c8 & 1) .. +7.  So a lane's source is (row part + chunk part) + a
    // K-tile part that is the same for every lane -- exactly the scalar-base + lane-offset form of the DMA.
    const int imR = (EPI == EPI_PATCH) ? a.im2col_R : 0;
    auto a_off = [&](int row) -> unsigned {
        if (EPI == EPI_PATCH && imR) {
            const int G = imR >> 4, r = min(row, a.M - 1), b = r / a.rows_in, p = r - b * a.rows_in, gy = p / G, gx = p - gy * G;
            const int c8 = schunk >> 3;
            return (unsigned)(2 * (((long)b * 3 * imR + gy * 16 + (c8 >> 1)) * imR + gx * 16 + (c8 & 1) * 8));
        }
        return src_off(a.a_blocked ? row : min(row, a.M - 1), a.lda, a.a_blocked);
    };
    auto a_tile = [&](int kt) -> const char* {           // wave-uniform
        if (EPI == EPI_PATCH && imR) return (const char*)A + 2 * ((long)(kt >> 2) * imR * imR + (kt & 3) * 4 * imR);
        return (const char*)A + (long)kt * a_step;
    };
#pragma unroll
    for (int j = 0; j < AJ; ++j) {
        const int r = m0 + wave * (BM / 8) + j * 8 + srow;
        oa[j] = a_off(r);
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = n0 + wave * 32 + j * 8 + srow;
        ob[j] = src_off(a.w_blocked ? r : min(r, a.N - 1), a.ldw, a.w_blocked);
    }
    const int ldsA_w = wave * (BM / 8) * 128;
    const int ldsB_w = A_BYTES + wave * 32 * 128;

    auto stage = [&](int buf, int kt) {
        char* base = smem + buf * STAGE;
#pragma unroll
        for (int j = 0; j < AJ; ++j)
            __builtin_amdgcn_global_load_lds((gptr_t)(a_tile(kt) + oa[j]), (lptr_t)(base + ldsA_w + j * 1024), 16, 0, (OPT & 1) ? 2 : 0);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            __builtin_amdgcn_global_load_lds((gptr_t)((const char*)W + ob[j] + (long)kt * w_step), (lptr_t)(base + ldsB_w + j * 1024), 16, 0, (OPT & 2) ? 2 : 0);
    };

    float4_t acc[MT][4];
    // (OPT & 16384, below: the timing-only 32x32x16 form of the ping-pong K loop)
    typedef float float16v __attribute__((ext_vector_type(16)));
    [[maybe_unused]] float16v acc32[8];
    if constexpr ((OPT & 16384) != 0) {
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int k = 0; k < 16; ++k) acc32[i][k] = 0.f;
    }
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};

    const int fr = lane & 15, fg = lane >> 4;
    const int a_row_off = (wm * (BM / 2) + fr) * 128;
    const int b_row_off = A_BYTES + (wn * 64 + fr) * 128;
    const int ch0 = ((fg) ^ (fr & 7)) << 4, ch1 = ((4 + fg) ^ (fr & 7)) << 4;

    constexpr bool PH8 = (OPT & 16) != 0 && MT == 8;
    if (!PH8) stage(0, 0);

    // epilogue constants -> LDS, under the latency of the first K-tile (first read after the K loop: many barriers later)
    if (PH8) {
    } else if (tid < BN5) {
        const int n = n0 + tid;
        float c0 = 0.f, c1 = 0.f;
        if (n < a.N) {
            if (HAS_BIAS) c0 = (float)((const half_t*)a.bias)[n];
            if (LNF) { c0 = a.ln_g[n]; c1 = a.ln_b[n]; }
        }
        col_c[tid] = c0;
        col_c[BN5 + tid] = c1;
    } else if (LNF && tid - BN5 < BM) {
        const int r = tid - BN5;
        const float2_t* sp = (const float2_t*)a.ln_stats + (long)min(m0 + r, a.M - 1) * (a.ln_stride ? a.ln_stride : a.ln_slots);
        float su = 0.f, sq = 0.f;
        for (int sl = 0; sl < a.ln_slots; ++sl) { const float2_t p = sp[sl]; su += p[0]; sq += p[1]; }
        const float inv_k = 1.0f / (float)a.K;
        const float mean = su * inv_k;
        const float rstd = 1.0f / sqrtf(fmaxf(sq * inv_k - mean * mean, 0.f) + 1e-5f);
        row_c[r] = rstd;
        row_c[BM + r] = -rstd * mean;
    }

    if constexpr ((OPT & 16) != 0 && MT == 8) {
    // ------------------------------------------------------------------------------------------------------------------
    // Ping-pong K loop (OPT & 16; the 256x256 tile only), after the CDNA4 guide's "256^2 8-phase template":
    //   * a K-tile is staged as FOUR 16 KiB half-tiles -- B0, A0, B1, A1 -- into 2 x 4 LDS slots (128 KiB as before);
    //     A half h holds the tile rows wm*128 + h*64 + [0,64) of BOTH wave rows, B half g the tile columns
    //     wn*64 + g*32 + [0,32) of all four wave columns, so every wave works on the same half-tiles at the same time
    //     while its own 128 x 64 output block stays contiguous (the epilogue below is unchanged);
    //   * TWO phases per K-tile, two 64 x 32 quadrants of the wave's block (32 MFMAs) each: M1 reads B0, A0, B1 (16 fragment
    //     reads) and runs (A0,B0) (A0,B1); M2 reads A1 (8) and runs (A1,B1) (A1,B0); every phase issues TWO half-tiles of
    //     LDS-DMA (4 instructions per lane).  (r02: the template's FOUR 16-MFMA phases per K-tile ran first -- 12 / 4 / 8 / 0
    //     reads, one half-tile per phase, vmcnt(6); same-process A/B of the two, bit-identical outputs: qkv 334 -> 313 us,
    //     c_fc 478 -> 459, N = 768 shapes equal; end to end generation +1.4 %.  Half the barriers; levelling the four-phase
    //     reads to 8 / 4 / 8 / 4 instead had changed nothing.)
    //   * the half-tiles of the next tiles are issued two phases ahead of their first read and the wait is COUNTED (vmcnt(4)
    //     once per K-tile, in M2, never 0 inside the loop): the two half-tiles just issued stay in flight across the barriers;
    //   * the two wave rows run ONE barrier apart: while wm = 0 issues its MFMAs, wm = 1 reads fragments and issues DMA,
    //     then they swap -- each SIMD holds one wave of either row, so its matrix pipe and its LDS / VMEM issue alternate.
    // Where a K-tile's ~3100 cycles go (r03s, tools/gemm_stamps.py, shader-clock stamps of one workgroup): per phase and wave, load work
    // (fragment reads + DMA issue + counted wait) 360-600, wait at the mid barrier for the other row's MFMAs 120-400, the 32 MFMAs 590-640
    // (512 back to back), phase-end barrier turn-around ~130: the half-period is MFMA segment + barrier turn-around ~ 770, the load work
    // hides.  Taking the phase-end barrier 3 or 8 MFMAs early (so that the other row is released while this one still feeds the pipe)
    // was bit-identical and 8-10 % SLOWER on every shape (r03s): two rows' MFMAs interleaved on one SIMD cost more than the turn-around.
    // Slot reuse (WAR): every phase retires its fragment reads (lgkmcnt(0)) BEFORE its first barrier, so a half-tile may be
    // restaged one phase after its last read; a staged half-tile is first read one phase after the counted wait + barrier that
    // retires it (RAW) -- the guide's rules for two wave groups staggered by a barrier.
    // ------------------------------------------------------------------------------------------------------------------
    constexpr int SLOT = 16384, KBUF = 4 * SLOT;
    enum { hB0 = 0, hA0 = 1, hB1 = 2, hA1 = 3 };
    unsigned sa8[2][2], sb8[2][2];                      // [half][8-row piece]: byte offset of this lane's source row
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int lr = wave * 16 + j * 8 + srow;                                    // local row inside the half-tile
            const int ra = m0 + (lr >> 6) * 128 + hf * 64 + (lr & 63);
            const int rb = n0 + (lr >> 5) * 64 + hf * 32 + (lr & 31);
            sa8[hf][j] = a_off(ra);
            sb8[hf][j] = src_off(a.w_blocked ? rb : min(rb, a.N - 1), a.ldw, a.w_blocked);
        }
    const unsigned smem_lds = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lptr_t)smem);
    const unsigned wave_lds = (unsigned)__builtin_amdgcn_readfirstlane(wave * 2048);
    auto stage_half = [&](int buf, int which, int kt) {
        const int hf = which >> 1;
        // LDS-DMA as scalar base (the K-tile's) + 32-bit lane offset, from inline asm: hipcc turns the builtin's address into a 64-bit
        // vector add per instruction (v_lshl_add_u64) and the LDS address into two v_readfirstlane -- vector-port instructions of the
        // loading row beside the other row's MFMAs.  Here a DMA instruction costs scalar instructions only (r03w, same-process A/B at
        // M = 151 296: qkv 479 -> 468 us, c_fc 698 -> 691, c_proj 582 -> 578, K loop alone +0.6-1.3 %).  The compiler does not see these
        // loads: every wait for them is one of the counted s_waitcnt of the loop (tests/test_isa_sync_templates.py).
        const char* ak = a_tile(kt);
        const char* wk = (const char*)W + (long)kt * w_step;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const unsigned dst = smem_lds + (unsigned)(buf * KBUF + which * SLOT + j * 1024) + wave_lds;   // scalar arithmetic only
            if (which & 1) {
                if constexpr ((OPT & 1) != 0) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" :: "v"(sa8[hf][j]), "s"(ak), "s"(dst) : "memory", "m0");
                else asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(sa8[hf][j]), "s"(ak), "s"(dst) : "memory", "m0");
            } else {
                if constexpr ((OPT & 2) != 0) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1 nt" :: "v"(sb8[hf][j]), "s"(wk), "s"(dst) : "memory", "m0");
                else asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(sb8[hf][j]), "s"(wk), "s"(dst) : "memory", "m0");
            }
        }
    };
    const int a_lane = (wm * 64 + fr) * 128, b_lane = (wn * 32 + fr) * 128;
    half8_t fa[4][2], fb0[2][2], fb1[2][2];
    auto read_a = [&](const char* buf, int hf) {
#pragma unroll
        for (int ii = 0; ii < 4; ++ii) {
            fa[ii][0] = *(const half8_t*)(buf + (1 + 2 * hf) * SLOT + a_lane + ii * 2048 + ch0);
            fa[ii][1] = *(const half8_t*)(buf + (1 + 2 * hf) * SLOT + a_lane + ii * 2048 + ch1);
        }
    };
    auto read_b = [&](const char* buf, int g, half8_t (&fb)[2][2]) {
#pragma unroll
        for (int jj = 0; jj < 2; ++jj) {
            fb[jj][0] = *(const half8_t*)(buf + 2 * g * SLOT + b_lane + jj * 2048 + ch0);
            fb[jj][1] = *(const half8_t*)(buf + 2 * g * SLOT + b_lane + jj * 2048 + ch1);
        }
    };
    // OPT & 16384 (experiment builds, timing only, with NOEPI): the K loop's matrix work as v_mfma_f32_32x32x16_f16 -- the same fragment
    // reads, half as many MFMA instructions (8 of 32 cycles on the issue port instead of 8 of 16); the products are meaningless.
    auto quadrant = [&](int hf, int g, half8_t (&fb)[2][2]) {
        __builtin_amdgcn_s_setprio(1);
        if constexpr ((OPT & 16384) != 0) {
#pragma unroll
            for (int st = 0; st < 2; ++st)
#pragma unroll
                for (int ip = 0; ip < 2; ++ip)
#pragma unroll
                    for (int jj = 0; jj < 2; ++jj) {
                        asm volatile("" :: "v"(fa[2 * ip + 1][st]));      // the fragment a real 32x32 layout would fold into its A operand
                        acc32[(hf * 2 + g) * 2 + jj] =
                            __builtin_amdgcn_mfma_f32_32x32x16_f16(fb[jj][st], fa[2 * ip][st], acc32[(hf * 2 + g) * 2 + jj], 0, 0, 0);
                    }
            __builtin_amdgcn_s_setprio(0);
            return;
        }
#pragma unroll
        for (int st = 0; st < 2; ++st)
#pragma unroll
            for (int ii = 0; ii < 4; ++ii)
#pragma unroll
                for (int jj = 0; jj < 2; ++jj)
                    acc[hf * 4 + ii][g * 2 + jj] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[jj][st], fa[ii][st], acc[hf * 4 + ii][g * 2 + jj], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
    };
#define OVMR_PH_END()                                                     \
    __builtin_amdgcn_sched_barrier(0);                                    \
    __builtin_amdgcn_s_barrier();                                         \
    __builtin_amdgcn_sched_barrier(0);
    // OPT & 8192 (experiment builds, tools/gemm_stamps.py): shader-clock stamps of ONE workgroup at the points of a phase where no LDS
    // read is outstanding (s_memtime returns through lgkmcnt) -- phase start, MFMA start (behind lgkmcnt(0) + barrier), MFMA end.
    long long* stamp_p = nullptr;
    if constexpr ((OPT & 8192) != 0) {
        if (a.argmax_out && blockIdx.x == 300) stamp_p = (long long*)a.argmax_out + (long)wave * 4096;
    }
    int stamp_i = 0;
#define OVMR_STAMP()                                                                                   \
    if constexpr ((OPT & 8192) != 0) {                                                                 \
        if (stamp_p) {                                                                                 \
            const long long t_ = __builtin_amdgcn_s_memtime();                                         \
            if (lane == 0) stamp_p[stamp_i] = t_;                                                      \
            ++stamp_i;                                                                                 \
        }                                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                                             \
    }

    // prologue: tile 0 complete, two half-tiles of tile 1 in flight
    stage_half(0, hB0, 0); stage_half(0, hA0, 0); stage_half(0, hB1, 0); stage_half(0, hA1, 0);
    // epilogue constants -> LDS (the compiler waits for their global loads with vmcnt(0), i.e. also for tile 0, which the first
    // phase needs anyway; first read after the K loop)
    if (tid < BN5) {
        const int n = n0 + tid;
        float c0 = 0.f, c1 = 0.f;
        if (n < a.N) {
            if (HAS_BIAS) c0 = (float)((const half_t*)a.bias)[n];
            if (LNF) { c0 = a.ln_g[n]; c1 = a.ln_b[n]; }
        }
        col_c[tid] = c0;
        col_c[BN5 + tid] = c1;
    } else if (LNF && tid - BN5 < BM) {
        const int r = tid - BN5;
        const float2_t* sp = (const float2_t*)a.ln_stats + (long)min(m0 + r, a.M - 1) * (a.ln_stride ? a.ln_stride : a.ln_slots);
        float su = 0.f, sq = 0.f;
        for (int sl = 0; sl < a.ln_slots; ++sl) { const float2_t p = sp[sl]; su += p[0]; sq += p[1]; }
        const float inv_k = 1.0f / (float)a.K;
        const float mean = su * inv_k;
        const float rstd = 1.0f / sqrtf(fmaxf(sq * inv_k - mean * mean, 0.f) + 1e-5f);
        row_c[r] = rstd;
        row_c[BM + r] = -rstd * mean;
    }
    __builtin_amdgcn_sched_barrier(0);
    stage_half(1, hB0, 1); stage_half(1, hA0, 1);
    asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    if (wm == 1) __builtin_amdgcn_s_barrier();          // the second wave row runs one barrier behind the first
    __builtin_amdgcn_sched_barrier(0);
    const char* const bufE = smem;
    const char* const bufO = smem + KBUF;
#define OVMR_P4_MID()                                                     \
    __builtin_amdgcn_sched_barrier(0);                                    \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                    \
    OVMR_STAMP()                                                          \
    __builtin_amdgcn_s_barrier();                                         \
    __builtin_amdgcn_sched_barrier(0);
    for (int kt = 0; kt < nk; kt += 2) {
        const bool more = kt + 2 < nk;
        // M1 even: stage B1, A1 of the odd tile
        OVMR_STAMP()
        read_b(bufE, 0, fb0);
        read_a(bufE, 0);
        read_b(bufE, 1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        stage_half(1, hB1, kt + 1); stage_half(1, hA1, kt + 1);
        OVMR_P4_MID()
        OVMR_STAMP()
        quadrant(0, 0, fb0);
        quadrant(0, 1, fb1);
        OVMR_STAMP()
        OVMR_PH_END()
        // M2 even: stage B0, A0 of tile kt+2; the odd tile must have landed
        OVMR_STAMP()
        read_a(bufE, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (more) { stage_half(0, hB0, kt + 2); stage_half(0, hA0, kt + 2); }
        __builtin_amdgcn_sched_barrier(0);
        if (more) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        OVMR_P4_MID()
        OVMR_STAMP()
        quadrant(1, 1, fb1);
        quadrant(1, 0, fb0);
        OVMR_STAMP()
        OVMR_PH_END()
        // M1 odd: stage B1, A1 of tile kt+2
        OVMR_STAMP()
        read_b(bufO, 0, fb0);
        read_a(bufO, 0);
        read_b(bufO, 1, fb1);
        __builtin_amdgcn_sched_barrier(0);
        if (more) { stage_half(0, hB1, kt + 2); stage_half(0, hA1, kt + 2); }
        OVMR_P4_MID()
        OVMR_STAMP()
        quadrant(0, 0, fb0);
        quadrant(0, 1, fb1);
        OVMR_STAMP()
        OVMR_PH_END()
        // M2 odd: stage B0, A0 of tile kt+3; tile kt+2 must have landed
        OVMR_STAMP()
        read_a(bufO, 1);
        __builtin_amdgcn_sched_barrier(0);
        if (more) { stage_half(1, hB0, kt + 3); stage_half(1, hA0, kt + 3); }
        __builtin_amdgcn_sched_barrier(0);
        if (more) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        OVMR_P4_MID()
        OVMR_STAMP()
        quadrant(1, 1, fb1);
        quadrant(1, 0, fb0);
        OVMR_STAMP()
        OVMR_PH_END()
    }
#undef OVMR_P4_MID
    if (wm == 0) __builtin_amdgcn_s_barrier();          // balances the extra barrier of the second wave row
    __builtin_amdgcn_sched_barrier(0);
#undef OVMR_PH_END
#undef OVMR_STAMP
    } else
    if (OPT & 4) {
    // K loop with the iteration boundary moved INSIDE the MFMA stream.  All of a K-tile's fragment reads are issued two steps
    // before its last MFMAs, so the wait for the next tile's LDS-DMA, the workgroup barrier, the next tile's first six fragment
    // reads and the LDS-DMA issue for the tile after it all happen in front of the last two steps (8 MFMAs per wave): the matrix
    // pipe works through those while the new fragments are in flight, instead of every wave of the CU waiting on LDS at once.
    constexpr int D = 2, R = D + 2;                 // deferred steps / A-fragment ring
    static_assert((2 * MT) % R == 0 && D <= MT, "");
    half8_t fb[2][4], fa[R];
    auto first_reads = [&](const char* buf) {
#pragma unroll
        for (int t = 0; t < 4; ++t) fb[0][t] = *(const half8_t*)(buf + b_row_off + t * 2048 + ch0);
#pragma unroll
        for (int t = 0; t < D; ++t) fa[t] = *(const half8_t*)(buf + a_row_off + (t % MT) * 2048 + (t / MT ? ch1 : ch0));
    };
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
    first_reads(smem);
    __builtin_amdgcn_sched_barrier(0);
    if (nk > 1) stage(1, 1);
    __builtin_amdgcn_sched_barrier(0);
    for (int kt = 0; kt < nk; ++kt) {
        const char* cur = smem + (kt & 1) * STAGE;
#pragma unroll
        for (int st = 0; st < 2 * MT - D; ++st) {
            const int nx = st + D;
            fa[nx % R] = *(const half8_t*)(cur + a_row_off + (nx % MT) * 2048 + (nx / MT ? ch1 : ch0));
            if (st == 1) {
#pragma unroll
                for (int t = 0; t < 4; ++t) fb[1][t] = *(const half8_t*)(cur + b_row_off + t * 2048 + ch1);
            }
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[st % MT][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[st / MT][j], fa[st % R], acc[st % MT][j], 0, 0, 0);
        }
#pragma unroll
        for (int st = 0; st < 2 * MT - D; ++st) {
            if (st == 1) __builtin_amdgcn_sched_group_barrier(0x100, 5, 0);
            else __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x008, 4, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // tile kt+1 landed (issued one iteration ago)
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // every fragment of tile kt is in registers
            __builtin_amdgcn_s_barrier();                          // ... for every wave: buffer kt & 1 is free
            __builtin_amdgcn_sched_barrier(0);
            first_reads(smem + ((kt + 1) & 1) * STAGE);
            __builtin_amdgcn_sched_barrier(0);
            if (kt + 2 < nk) stage(kt & 1, kt + 2);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int st = 2 * MT - D; st < 2 * MT; ++st)
#pragma unroll
            for (int j = 0; j < 4; ++j)
                acc[st % MT][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[st / MT][j], fa[st % R], acc[st % MT][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    } else {
    for (int kt = 0; kt < nk; ++kt) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                  // tile kt landed for every wave; buffer (kt+1)&1 is free
        __builtin_amdgcn_sched_barrier(0);
        const char* cur = smem + (kt & 1) * STAGE;
        half8_t fb[2][4], fa[3];
        // software-pipelined fragment reads: A fragment of step t+2 and the B fragments of the next k-step are
        // issued before the MFMAs of step t; sched_group_barrier pins that order for the machine scheduler.
        // The first six reads go out BEFORE the next K-tile's eight LDS-DMA instructions: their LDS latency then runs
        // under the DMA issue (address adds + VMEM issue) instead of after it (same-process A/B: within noise, kept).
#pragma unroll
        for (int t = 0; t < 4; ++t) fb[0][t] = *(const half8_t*)(cur + b_row_off + t * 2048 + ch0);
        fa[0] = *(const half8_t*)(cur + a_row_off + ch0);
        fa[1] = *(const half8_t*)(cur + a_row_off + 2048 + ch0);
        __builtin_amdgcn_sched_barrier(0);
        if (kt + 1 < nk) stage((kt + 1) & 1, kt + 1);
        __builtin_amdgcn_sched_barrier(0);
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
    }

    }
    // ---------------------------------------------------------------- epilogue through LDS
    half_t* C = (half_t*)a.C;
    if (NOEPI) {            // timing-only ablation: the K loop alone (the store below never happens on real data)
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) sum += acc[i][j][0] + acc[i][j][1] + acc[i][j][2] + acc[i][j][3];
        if constexpr ((OPT & 16384) != 0) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int k = 0; k < 16; ++k) sum += acc32[i][k];
        }
        if (sum == 12345.678f) C[tid] = (half_t)sum;
        return;
    }
    if constexpr (EPI == EPI_SCALE_ARGMAX) {
        // Logits that are never stored: u = h(h(acc) * scale) exactly as EPI_SCALE, then per row the maximum of this tile's
        // columns and the LOWEST column holding it.  A row's 64 columns of one wave sit in four lanes (fg) x 16 values;
        // lanes walk their columns in increasing order, so a strict > keeps the first maximum; lanes and the four column
        // waves are merged with ties to the lower column.
        __syncthreads();                                               // every wave is done with the K buffers
        int* arg_lds = (int*)smem;                                     // [BM rows][4 column waves][value bits, column]
#pragma unroll
        for (int i = 0; i < MT; ++i) {
            float bv = -INFINITY;
            int bc = 0x7fffffff;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int c0 = n0 + wn * 64 + j * 16 + fg * 4;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float u = (float)(half_t)((float)(half_t)acc[i][j][r] * a.scale);
                    if (c0 + r < a.N && (u > bv || bc == 0x7fffffff)) { bv = u; bc = c0 + r; }
                }
            }
#pragma unroll
            for (int o = 16; o <= 32; o <<= 1) {
                const float ov = __shfl_xor(bv, o, 64);
                const int oc = __shfl_xor(bc, o, 64);
                if (ov > bv || (ov == bv && oc < bc)) { bv = ov; bc = oc; }
            }
            if (fg == 0) {
                int* d = arg_lds + ((wm * (BM / 2) + i * 16 + fr) * 4 + wn) * 2;
                d[0] = __float_as_int(bv);
                d[1] = bc;
            }
        }
        __syncthreads();
        if (tid < BM && m0 + tid < a.M) {
            float bv = __int_as_float(arg_lds[tid * 8]);
            int bc = arg_lds[tid * 8 + 1];
#pragma unroll
            for (int w = 1; w < 4; ++w) {                               // column waves in increasing column order
                const float v = __int_as_float(arg_lds[tid * 8 + 2 * w]);
                const int vc = arg_lds[tid * 8 + 2 * w + 1];
                if (v > bv || (v == bv && vc < bc)) { bv = v; bc = vc; }
            }
            int* dst = (int*)a.argmax_out + ((long)(m0 + tid) * tiles_n + tn) * 2;
            dst[0] = __float_as_int(bv);
            dst[1] = bc;
        }
        return;
    }
    auto gelu4 = [](half4_t v) -> half4_t {
        if constexpr ((OPT & 32) != 0) {                // experiment builds: round 1's pairwise form, for the same-process A/B
            const half2_t lo = quick_gelu_h2((half2_t){v[0], v[1]}), hi = quick_gelu_h2((half2_t){v[2], v[3]});
            return (half4_t){lo[0], lo[1], hi[0], hi[1]};
        } else {
            return quick_gelu_h4(v);
        }
    };
    char* et = smem + wave * (64 * EP);                 // this wave's 64-row x 64-column staging tile
    const int er = lane >> 3, ec = (lane & 7) * 8;     // phase 2: row within an 8-row group, first column
    // The staging tile has 128-byte rows of sixteen 8-byte units; unit u of row r is stored at u ^ f(r & 15),
    // f(r) = ((r & 7) << 1) | (r >> 3): the 16-byte chunk index is XORed with r & 7 (the K loop's swizzle) and the two halves of
    // a chunk trade places in rows 8..15.  Phase 1 (8-byte writes, 16 lanes = the 16 rows of one unit column, 32 banks): f is a
    // bijection, 16 different units = all 32 banks.  Phase 2 reads whole chunks (ds_read_b128, four lanes each of four rows per
    // group, 64 banks): the chunk XOR keeps the two rows that share a bank half on different chunks, and whether a lane's chunk
    // arrives with its halves exchanged is a compile-time fact (rows it*8 + er: odd it), so putting them back costs nothing.
    // SQ_LDS_BANK_CONFLICT = 0 (profiles/r03b_pmc_gemm_v109.json); the 144-byte padded rows this replaces were 2-way on every
    // write and on one read group in four (r02o: 11.7 % of the LDS cycles) -- at equal run time (r03c: -0.3 ... +0.8 %).
    const int fsw = ((fr & 7) << 1) | (fr >> 3);
    int wr_off[4];                                      // phase 1: byte offset of column block j inside this lane's row fr
#pragma unroll
    for (int j = 0; j < 4; ++j) wr_off[j] = fr * EP + (((j * 4 + fg) ^ fsw) << 3);
    const int rd_off = er * EP + (((lane & 7) ^ er) << 4);
    auto read_staged = [&](int it) -> half8_t {         // rows it*8 + er, this lane's eight columns
        const half8_t v = *(const half8_t*)(et + it * 8 * EP + rd_off);
        if (it & 1) return (half8_t){v[4], v[5], v[6], v[7], v[0], v[1], v[2], v[3]};
        return v;
    };
    const bool emit_stats = EPI == EPI_BIAS_RES && a.stats_out != nullptr;
    const bool full_tile = EPI != EPI_PATCH && m0 + BM <= a.M && n0 + BN5 <= a.N;   // workgroup-uniform
    float* stat_lds = (float*)(smem + 8 * 64 * EP);      // [BM rows][4 column waves][2], behind the staging tiles
#pragma unroll
    for (int h = 0; h < MT / 4; ++h) {
        // The ONLY workgroup barrier of the epilogue: every wave's K-loop fragment reads must be done before staging tiles
        // overwrite the K buffers.  The staging tile itself is private to its wave (LDS executes a wave's instructions in
        // order), so neither the write -> read turn inside a half nor the read -> write turn between the halves needs one;
        // the barriers that used to sit there also drained vmcnt, i.e. made every wave wait for the first half's global
        // stores before converting the second half.
        if (h == 0) __syncthreads();
        // residual rows of this half: requested now, consumed after phase 1 (their latency used to sit in front of every
        // store: ~6 us per 256x256 tile on out_proj / c_proj).  Requesting them earlier -- the first half under the last
        // K-tile's MFMAs (249 VGPRs), or both halves here -- measured 4-7 % SLOWER on out_proj / c_proj.
        half8_t res8[EPI == EPI_BIAS_RES ? 8 : 1];
        if (EPI == EPI_BIAS_RES) {
            const int nn = n0 + wn * 64 + ec;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int m = min(m0 + wm * (BM / 2) + h * 64 + it * 8 + er, a.M - 1);
                res8[EPI == EPI_BIAS_RES ? it : 0] = *(const half8_t*)((const half_t*)a.res + (long)m * a.ldres + min(nn, a.N - 8));   // (a nontemporal load here: out_proj +3 %, c_proj equal, r03)
            }
        }
        float ln_rs[LNF ? 4 : 1], ln_ts[LNF ? 4 : 1];   // LNF: rstd and -rstd * mean of this lane's four rows
        if (LNF) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int r = wm * (BM / 2) + h * 64 + i * 16 + fr;
                ln_rs[LNF ? i : 0] = row_c[r];
                ln_ts[LNF ? i : 0] = row_c[BM + r];
            }
        }
        // column tile outermost: its constants are read from LDS once per half tile (not once per 16x16 accumulator block).
        // (Reading all of them into registers ahead of the h loop measured 4 % SLOWER on the LN-folding kernels: that
        // changed the K loop's register allocation.)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int cc = wn * 64 + j * 16 + fg * 4;
            const float4_t c0 = *(const float4_t*)(col_c + cc);
            float4_t c1 = c0;
            if (LNF) c1 = *(const float4_t*)(col_c + BN5 + cc);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float4_t v = acc[h * 4 + i][j];
                const float ln_r = ln_rs[LNF ? i : 0], ln_t = ln_ts[LNF ? i : 0];
                half4_t o;
                if (LNF) {
#pragma unroll
                    for (int r = 0; r < 4; r += 2) {
                        // two columns per instruction (v_pk_fma_f32): same two fused multiply-adds per element as the scalar form
                        const float2_t t2 = __builtin_elementwise_fma((float2_t){ln_t, ln_t}, (float2_t){c0[r], c0[r + 1]}, (float2_t){c1[r], c1[r + 1]});
                        float2_t x2 = __builtin_elementwise_fma((float2_t){ln_r, ln_r}, (float2_t){v[r], v[r + 1]}, t2);
                        if (EPI == EPI_LN_BIAS_QGELU && GFAST) x2 = quick_gelu_f32x2(x2);
                        const half2_t u2 = __builtin_convertvector(x2, half2_t);
                        o[r] = u2[0];
                        o[r + 1] = u2[1];
                    }
                    if (EPI == EPI_LN_BIAS_QGELU && !GFAST) o = gelu4(o);
                } else {
                    // float2 sums and __builtin_convertvector: v_pk_add_f32 + v_cvt_pk_f16_f32 (round to nearest even), two
                    // elements per instruction; element-wise casts made hipcc emit cvt + pack + alignbit chains (3.8 -> ~1.6
                    // vector instructions per element in this phase)
#pragma unroll
                    for (int r = 0; r < 4; r += 2) {
                        float2_t x2 = {v[r], v[r + 1]};
                        if (HAS_BIAS) x2 += (float2_t){c0[r], c0[r + 1]};
                        if (EPI == EPI_BIAS_QGELU && GFAST) x2 = quick_gelu_f32x2(x2);
                        half2_t u2 = __builtin_convertvector(x2, half2_t);
                        if (EPI == EPI_SCALE) {                                   // h(h(acc) * scale)
                            float2_t y2 = __builtin_convertvector(u2, float2_t);
                            y2 *= a.scale;
                            u2 = __builtin_convertvector(y2, half2_t);
                        }
                        o[r] = u2[0];
                        o[r + 1] = u2[1];
                    }
                    if (EPI == EPI_BIAS_QGELU && !GFAST) o = gelu4(o);
                }
                *(half4_t*)(et + i * 16 * EP + wr_off[j]) = o;
            }
        }
        const int nn = n0 + wn * 64 + ec;
        // residual add + row statistics of one 8-column slice (EPI_BIAS_RES); 8 lanes share a row
        auto finish = [&](half8_t v, int it, int row) -> half8_t {
            if (EPI == EPI_BIAS_RES) {
                const half8_t r8 = res8[EPI == EPI_BIAS_RES ? it : 0];
#pragma unroll
                for (int k = 0; k < 8; ++k) v[k] = (half_t)((float)v[k] + (float)r8[k]);
                if (emit_stats) {                       // statistics of the STORED fp16 row slice
                    // v_dot2_f32_f16 (two fp16 products + fp32 accumulate) and DPP lane exchanges inside the 8-lane
                    // group: 14 instructions where cvt/add chains and ds_bpermute shuffles took ~45 per store
                    float su = 0.f, sq = 0.f;
                    const half2_t one2 = {(half_t)1.f, (half_t)1.f};
#pragma unroll
                    for (int k = 0; k < 8; k += 2) {
                        const half2_t v2 = {v[k], v[k + 1]};
                        su = __builtin_amdgcn_fdot2(v2, one2, su, false);
                        sq = __builtin_amdgcn_fdot2(v2, v2, sq, false);
                    }
                    su += dpp_f32<0xB1>(su); sq += dpp_f32<0xB1>(sq);       // quad_perm [1,0,3,2]: lane ^ 1
                    su += dpp_f32<0x4E>(su); sq += dpp_f32<0x4E>(sq);       // quad_perm [2,3,0,1]: lane ^ 2
                    su += dpp_f32<0x141>(su); sq += dpp_f32<0x141>(sq);     // row_half_mirror: the other quad of the 8
                    if ((lane & 7) == 0)
                        *(float2_t*)(stat_lds + ((wm * (BM / 2) + h * 64 + row) * 4 + wn) * 2) = (float2_t){su, sq};
                }
            }
            return v;
        };
        if (full_tile) {
            // interior tile (all but the last row / column of tiles): the eight LDS reads go out back to back and the stores
            // walk one pointer; the guarded loop below costs 13 instructions and one exposed LDS round trip per store
            half8_t vv[8];
#pragma unroll
            for (int it = 0; it < 8; ++it) vv[it] = read_staged(it);
            half_t* dst = C + (long)(m0 + wm * (BM / 2) + h * 64 + er) * a.ldc + nn;
            const long step = 8L * a.ldc;
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const half8_t v = finish(vv[it], it, it * 8 + er);
                if (NT) __builtin_nontemporal_store(v, (half8_t*)(dst + it * step));
                else if (!NOSTORE || (float)v[0] == 12345.678f) *(half8_t*)(dst + it * step) = v;
            }
        } else {
#pragma unroll
            for (int it = 0; it < 8; ++it) {
                const int row = it * 8 + er;
                const int m = m0 + wm * (BM / 2) + h * 64 + row;
                if (m < a.M && nn < a.N) {                  // N % 8 == 0 is guaranteed by the launcher
                    half8_t v = finish(read_staged(it), it, row);
                    long crow = m;
                    if (EPI == EPI_PATCH) {
                        const int b = m / a.rows_in, p = m - b * a.rows_in;
                        crow = (long)b * a.rows_out + 1 + p;
                        half8_t p8 = *(const half8_t*)((const half_t*)a.pos + (long)(1 + p) * a.N + nn);
#pragma unroll
                        for (int k = 0; k < 8; ++k) v[k] = (half_t)((float)v[k] + (float)p8[k]);
                    }
                    half8_t* dst = (half8_t*)(C + crow * a.ldc + nn);
                    if (NT) __builtin_nontemporal_store(v, dst);
                    else if (!NOSTORE || (float)v[0] == 12345.678f) *dst = v;   // NOSTORE: timing-only ablation
                }
            }
        }
    }
    if (emit_stats) {       // one (sum, sum of squares) pair per row and N tile, the four column waves added in fixed order
        // LDS-only barrier: __syncthreads() would also wait for vmcnt(0), i.e. for this tile's global stores to drain
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (tid < BM && m0 + tid < a.M) {
            const float* p = stat_lds + tid * 8;
            const float su = ((p[0] + p[2]) + p[4]) + p[6], sq = ((p[1] + p[3]) + p[5]) + p[7];
            *(float2_t*)(a.stats_out + ((long)(m0 + tid) * tiles_n + tn) * 2) = (float2_t){su, sq};
        }
    }
}

template <int EPI, int MT, int OPT>
int launch_v5_k(const GemmArgs& b, int tiles_m, int tiles_n, hipStream_t s) {
    constexpr int BM = MT * 32;
    const size_t lds = (size_t)2 * (BM + BN5) * 128 + 2 * BN5 * 4 + 2 * BM * 4;
    static bool attr_set[OVMR_MAX_DEVICES] = {};      // the attribute is per device (one process may drive several)
    int dev = 0;
    HIP_CHECK_RET(hipGetDevice(&dev));
    if (dev >= 0 && dev < OVMR_MAX_DEVICES && !attr_set[dev]) {
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)gemm_f16_v5_kernel<EPI, MT, OPT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set[dev] = true;
    }
    hipLaunchKernelGGL((gemm_f16_v5_kernel<EPI, MT, OPT>), dim3(tiles_m * tiles_n), dim3(512), lds, s, b, tiles_m, tiles_n);
    return (int)hipGetLastError();
}

template <int EPI, int MT, int OPT>
int launch_v5(const GemmArgs& a, hipStream_t s) {
    constexpr int BM = MT * 32;
    const int tiles_m = (a.M + BM - 1) / BM, tiles_n = (a.N + BN5 - 1) / BN5;
    GemmArgs b = a;
    if (b.n_group <= 0) {
        static const int force = exp_env("OVMR_N_GROUP");
        // measured (profiles/r01e_gemm_experiments.md): groups of 4-6 raise the L2 hit rate of qkv / c_fc from 65-68 %
        // to 72-73 % but move the run time by < 2 %, and hurt c_proj; the default therefore stays row-major (G = all)
        // with the ping-pong K loop (r02c, same-process A/B at batch 512): groups of 4 N tiles take 1.5-2.5 % off qkv / c_fc
        // (c_fc_ln 496 -> 484 us, qkv_ln 329 -> 325 us), nothing off the N = 768 shapes
        b.n_group = force > 0 ? std::min(force, tiles_n) : (((OPT & 16) && MT == 8 && tiles_n >= 8) ? 4 : tiles_n);
    }
    if (b.nt_store == 0) {
        // C written with the nontemporal hint does not evict the A / W panels the other tiles of the XCD are streaming from
        // its 4 MiB L2 (profiles/r01g_gemm_epilogue.md).  Not for the in-place residual updates (out_proj / c_proj: +4 %
        // slower) nor for small outputs the next kernel reads straight back (logits for the argmax).  The hint has to be
        // a template parameter: a run-time branch around two stores of the same value is merged by the compiler, which
        // drops the hint.
        static const int force = exp_env("OVMR_NT_STORE");
        b.nt_store = force ? force : (b.epi != EPI_BIAS_RES && (size_t)b.M * b.N * 2 >= ((size_t)48 << 20) ? 2 : 1);
    }
    constexpr int XB = OPT & 2048;                      // QuickGELU form: carried into every K-loop choice below
    constexpr int P8 = OPT & (16 | 32 | XB);            // (32: experiment bit, rides along) ping-pong K loop (256-row tiles; it replaces the OPT & 4 loop there)
    constexpr bool PLAIN = (OPT & ~(16 | 32 | XB)) == 0;
    constexpr bool OV_OK = !((OPT & 16) && MT == 8);
    if constexpr (PLAIN && OV_OK && (EPI == EPI_LN_BIAS || EPI == EPI_LN_BIAS_QGELU)) {
        // the LayerNorm-folding launches also run the K loop with the boundary inside the MFMA stream (qkv_ln 354 -> 343 us,
        // c_fc_ln 525 -> 518 us; the plain bias / QuickGELU launches of the same shapes do not gain)
        static const int ov_force = exp_env("OVMR_K_OVERLAP");
        if (ov_force != 1) {
            if (b.nt_store == 2) return launch_v5_k<EPI, MT, XB | 4 | 512>(b, tiles_m, tiles_n, s);
            return launch_v5_k<EPI, MT, XB | 4>(b, tiles_m, tiles_n, s);
        }
    }
    if constexpr (PLAIN && EPI != EPI_BIAS_RES) {
#ifdef OVMR_EXPERIMENTS
        static const bool a_nt_all = exp_env("OVMR_A_NT") == 3;   // nontemporal A stream on every epilogue: measured slower (r02c)
        if (a_nt_all) {
            if (b.nt_store == 2) return launch_v5_k<EPI, MT, P8 | 1 | 512>(b, tiles_m, tiles_n, s);
            return launch_v5_k<EPI, MT, P8 | 1>(b, tiles_m, tiles_n, s);
        }
#endif
        if (b.nt_store == 2) return launch_v5_k<EPI, MT, P8 | 512>(b, tiles_m, tiles_n, s);
    }
    if constexpr (PLAIN && EPI == EPI_BIAS_RES) {
        // Residual projections.
        // (1) N <= 1024 (at most four N tiles share an A panel): the A stream is loaded with the nontemporal policy so that it
        //     does not push the W panels every tile re-reads out of the L2 (out_proj 146 -> 141 us, c_proj 477 -> 459 us).  With
        //     9-12 N tiles per A panel the same hint costs 6-11 %, and on the W operand it always costs.
        // (2) K >= 2048: the K loop with the iteration boundary inside the MFMA stream (OPT & 4): c_proj 467 -> 449 us; at
        //     K = 768 (12 K-tiles) it is neutral to 2 % slower.
        static const int a_nt_force = exp_env("OVMR_A_NT"), ov_force = exp_env("OVMR_K_OVERLAP");   // 1 = never, 2 = always
        const bool a_nt = a_nt_force != 1 && (a_nt_force == 2 || (tiles_n <= 4 && tiles_m * tiles_n >= 512));
        const bool ov = OV_OK && ov_force != 1 && (ov_force == 2 || b.K >= 2048);
        if (a_nt && ov) return launch_v5_k<EPI, MT, XB | 5>(b, tiles_m, tiles_n, s);
        if (ov) return launch_v5_k<EPI, MT, XB | 4>(b, tiles_m, tiles_n, s);
        if (a_nt) return launch_v5_k<EPI, MT, P8 | 1>(b, tiles_m, tiles_n, s);
    }
    return launch_v5_k<EPI, MT, OPT>(b, tiles_m, tiles_n, s);
}

template <int EPI, int OPT>
int pick_v5(const GemmArgs& a, hipStream_t s) {
    static const int g_force_mt = exp_env("OVMR_FORCE_MT");   // experiment builds: 4 | 8 pins the M tile
    auto eff = [&](int bm) {
        const double t = (double)((a.M + bm - 1) / bm) * ((a.N + BN5 - 1) / BN5);
        return t / (ceil(t / 256.0) * 256.0);
    };
    const double t256 = (double)((a.M + 255) / 256) * ((a.N + BN5 - 1) / BN5);
    bool big;
    if (g_force_mt) big = g_force_mt == 8;
    else if ((OPT & 16) && (a.K % 128) == 0) {
        // ping-pong K loop on 256-row tiles against the double-buffered loop on 128-row tiles: a round of 128-row tiles takes
        // ~0.74 of a round of 256-row tiles (r02d, batch 256: out_proj 19.8 vs 26.6 us, c_proj 59 vs 78 us per round), so the
        // big tile wins unless the small one saves a whole round -- e.g. 591 tiles (batch 256, N = 768): 3 rounds against
        // 5 x 0.74; the CLS-only tail (6 tiles) stays on 128-row tiles.
        const double t128 = (double)((a.M + 127) / 128) * ((a.N + BN5 - 1) / BN5);
        big = ceil(t256 / 256.0) <= 0.74 * ceil(t128 / 256.0);
    } else {
        // grids far below one round of CUs (the CLS-only tail of the last vision block: 512 rows): the smaller tile doubles the
        // workgroups and shortens each K-tile (out_proj 22.5 -> 14.0 us, c_proj 67.7 -> 42.6 us at 512 rows)
        big = t256 >= 64 && eff(256) + 0.08 >= eff(128);
    }
    // (r02q, measured and removed: running the rows beyond the last whole round of 256-row tiles as a second launch of 128-row
    // tiles -- batch-256 inference, N = 768: 2.31 rounds -> 2 + 0.78 -- took 6 % off out_proj (81 -> 76 us) but added 2 % to
    // c_proj (K = 3072: the small-tile round is no shorter there); 0.1 % of a step.)
    return big ? launch_v5<EPI, 8, OPT>(a, s) : launch_v5<EPI, 4, OPT>(a, s);
}

template <int EPI, int OPT>
int pick_gelu(const GemmArgs& a, hipStream_t s) {      // QuickGELU form (GemmArgs::gelu_mode) -> template bit 2048
    return a.gelu_mode ? pick_v5<EPI, OPT | 2048>(a, s) : pick_v5<EPI, OPT>(a, s);
}

template <int OPT>
int dispatch_v5(const GemmArgs& a, hipStream_t s) {
    switch (a.epi) {
        case EPI_NONE: return pick_v5<EPI_NONE, OPT>(a, s);
        case EPI_BIAS: return pick_v5<EPI_BIAS, OPT>(a, s);
        case EPI_BIAS_QGELU: return pick_gelu<EPI_BIAS_QGELU, OPT>(a, s);
        case EPI_BIAS_RES: return pick_v5<EPI_BIAS_RES, OPT>(a, s);
        case EPI_PATCH: return pick_v5<EPI_PATCH, OPT>(a, s);
        case EPI_SCALE: return pick_v5<EPI_SCALE, OPT>(a, s);
        case EPI_LN_BIAS: return pick_v5<EPI_LN_BIAS, OPT>(a, s);
        case EPI_LN_BIAS_QGELU: return pick_gelu<EPI_LN_BIAS_QGELU, OPT>(a, s);
        case EPI_SCALE_ARGMAX: return pick_v5<EPI_SCALE_ARGMAX, OPT>(a, s);
    }
    return -3;
}

}  // namespace

// variant 8: the default (ping-pong K loop); 6: the double-buffered K loop (also what 8 runs for an odd number of K-tiles and
// for 128-row tiles)
int launch_gemm_f16_v5(const GemmArgs& a, int variant, hipStream_t s) {
    if (a.epi == EPI_SCALE_ARGMAX) {
        if (a.M < 256 || a.N < 128 || !a.argmax_out || (long)a.M * a.lda * 2 >= 0x7fffffffL || (long)a.N * a.ldw * 2 >= 0x7fffffffL)
            return -100;
    } else
    if (a.M < 256 || a.N < 128 || (a.N & 7) || (a.ldc & 7) || (a.epi == EPI_BIAS_RES && (a.ldres & 7)) ||
        ((uintptr_t)a.C & 15) || (a.epi == EPI_BIAS_RES && ((uintptr_t)a.res & 15)) ||
        (long)a.M * a.lda * 2 >= 0x7fffffffL || (long)a.N * a.ldw * 2 >= 0x7fffffffL)
        return -100;
    if (a.im2col_R) {                                  // A = fp16 images [B, 3, R, R], 16 x 16 patches: K = 768, byte offsets in 32 bits
        const int G = a.im2col_R >> 4;
        if (a.epi != EPI_PATCH || a.K != 768 || (a.im2col_R & 15) || a.rows_in != G * G || ((uintptr_t)a.A & 15) ||
            (long)((a.M + a.rows_in - 1) / a.rows_in) * 3 * a.im2col_R * a.im2col_R * 2 >= 0xffffffffL)
            return -2;
    }
    const bool lnf = a.epi == EPI_LN_BIAS || a.epi == EPI_LN_BIAS_QGELU;
    if (lnf && ((a.N & 63) || !a.ln_stats || a.ln_slots < 1 || !a.ln_g || !a.ln_b)) return -2;
    if (a.stats_out && (a.epi != EPI_BIAS_RES || (a.N & 255))) return -2;
    switch (variant) {
#ifdef OVMR_EXPERIMENTS   // timing-only ablations (their outputs are wrong by construction): no stores / no epilogue at all
        case 18: return a.epi == EPI_BIAS_QGELU ? pick_v5<EPI_BIAS_QGELU, 64>(a, s) : pick_v5<EPI_BIAS, 64>(a, s);
        case 19: return pick_v5<EPI_BIAS, 128>(a, s);
        case 28: return a.epi == EPI_BIAS_QGELU ? pick_v5<EPI_BIAS_QGELU, 16 | 64>(a, s) : pick_v5<EPI_BIAS, 16 | 64>(a, s);
        case 29: return pick_v5<EPI_BIAS, 16 | 128>(a, s);
        case 38: return (a.K % 128) == 0 ? dispatch_v5<16 | 32>(a, s) : dispatch_v5<32>(a, s);   // variant 8 with the pairwise QuickGELU
        case 59: return pick_v5<EPI_BIAS, 16 | 128 | 16384>(a, s);   // variant 29 (K loop alone) with 32x32x16 MFMAs: timing only
        case 58: return pick_v5<EPI_BIAS, 16 | 8192>(a, s);   // variant 8 with shader-clock stamps of workgroup 300 (a.argmax_out = stamp buffer; tools/gemm_stamps.py)
#endif
        case 6: return dispatch_v5<0>(a, s);
        case 8: return (a.K % 128) == 0 ? dispatch_v5<16>(a, s) : dispatch_v5<0>(a, s);   // ping-pong K loop: two K-tiles per iteration
        default: return -5;                                                                // unknown variant
    }
}
