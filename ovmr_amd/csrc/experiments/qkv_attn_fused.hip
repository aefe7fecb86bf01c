// Experiment (round 6, experiment build only): in_proj + attention of one vision block FUSED per (image, head), L = 197, hd = 64.
//
// Today (clip/model.py:184-188 on the HIP path): in_proj (ln_1 folded, gemm_f16_v5) writes qkv [B L, 3 W] -- 703 MB at 775 ViT-B/16
// images -- and attention (attention_v3.hip) reads 620 MB of it straight back, every block.  Here ONE persistent workgroup per CU takes
// (image b, head h) pairs and never lets qkv leave the CU:
//
//   phase A   [Q | K | V]_h = LN(x_b) W_h^T : the 197 token rows of the image (padded to 208 = 13 x 16) times the 192 folded weight rows
//             of the head (64 of each third of in_proj), K = 768 in 12 K-tiles of 64, operands by LDS-DMA into two stages, 8 waves as
//             2 x 4 with (7 | 6) x 3 accumulator tiles of v_mfma_f32_16x16x32_f16 -- the same products in the same order as the
//             in_proj launch, and the same LayerNorm-fold epilogue arithmetic (common.h EPI_LN_BIAS), so every q / k / v value has the
//             bits of today's path;
//   epilogue  h(rstd (acc - mean g[n]) + b[n]) written to LDS in the image attention_v3 stages from global memory (128-byte rows,
//             16-byte chunk c of row r in slot c ^ (r & 7)): K rows, V rows, Q rows;
//   phase B   the 13 query tiles of the head through attn_single_pass.h (the body of attention variant 3), 8 waves -> two rounds;
//             output rows stored as variant 3 stores them.
// LDS: three K-tile stages of 50 KiB (two tiles in flight while one is consumed); K | V (26 KiB each) overlay the third stage, Q the second --
// the waves take their query fragments into registers at once, so the NEXT pair's first two K-tiles stream in under the attention arithmetic.  Pairs are dealt so that the 12 heads of an image
// run on ONE XCD (its L2 then serves 11 of the 12 reads of the image's rows).
//
// ovmr_debug_qkv_attn (below) runs today's pair of launches and this kernel on the same inputs, times both with HIP events and returns
// both outputs: tools/fused_qkv_bench.py compares them bit for bit and prints the times.
#include "../attn_single_pass.h"

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace {

typedef __attribute__((address_space(3))) void* lptr_t;

constexpr int FQ_ROWS = 208, FQ_COLS = 192, FQ_NT = 13;
constexpr int FQ_A_BYTES = FQ_ROWS * 128, FQ_B_BYTES = FQ_COLS * 128, FQ_STAGE = FQ_A_BYTES + FQ_B_BYTES;   // 51200
// LDS map: three K-tile stages S0 | S1 | S2 (a K-tile's arithmetic is ~0.35 us, a load's latency 1.5-2: two tiles must be in flight while
// a third is consumed); after the K loop the K | V images overlay S2 (and 2 KiB behind it), the Q image S1 -- the waves take their
// query fragments into registers right away, so during phase B S0 AND S1 are free for the next pair's first two K-tiles.
constexpr int FQ_KV = 2 * FQ_STAGE;                                 // K image, then V image
constexpr int FQ_Q = FQ_STAGE;                                      // Q image (dead once phase B has its fragments)
constexpr int FQ_CONST = FQ_KV + 2 * FQ_A_BYTES;                    // column constants [2][192] f32, row constants [2][208] f32
constexpr int FQ_LDS = FQ_CONST + 2 * FQ_COLS * 4 + 2 * FQ_ROWS * 4;
static_assert(FQ_LDS <= 160 * 1024, "LDS");

__device__ __forceinline__ void fq_glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

__device__ __forceinline__ int fq_swz(int row, int chunk) { return row * 128 + ((chunk ^ (row & 7)) << 4); }

// NWV: waves per workgroup -- 8 (2 wave rows x 4 wave columns: (7 | 6) x 3 accumulator tiles per wave, two waves per SIMD) or 16 (4 x 4:
// (4 | 3 | 3 | 3) x 3 tiles, four waves per SIMD: more fragment reads per MFMA, but other waves' reads and loads under every wave's MFMAs, and
// the 13 query tiles of phase B in ONE round).  PP: the ping-pong K loop (8 waves only; OVMR_FQ_ABL bit 16).  HP4: pairs dealt in passes of 4 heads.
template <int NWV, bool PP, bool HP4>
__global__ __launch_bounds__(NWV * 64) void qkv_attn_fused_kernel(const half_t* __restrict__ x, const half_t* __restrict__ wf,
                                                              const float* __restrict__ ln_g, const float* __restrict__ ln_b,
                                                              const float* __restrict__ stats, half_t* __restrict__ out,
                                                              int B, int L, int W, int H, float scale_log2e, int abl) {
    // abl (timing-only ablations, OVMR_FQ_ABL): 1 no phase B, 2 no MFMAs in the K loop, 4 no operand loads after a pair's first two tiles,
    // 8 no epilogue, 16 the ping-pong K loop
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    constexpr int NT_THREADS = NWV * 64, WM = NWV / 4, TMX = (FQ_NT + WM - 1) / WM;   // wave rows; row tiles of the first wave row (7 | 4)
    const int wm = wave >> 2, wn = wave & 3;
    const int fr = lane & 15, fg = lane >> 4;
    const int srow = lane >> 3, schunk = ((lane & 7) ^ srow) * 8;
    const int tm0 = wm * (FQ_NT / WM) + min(wm, FQ_NT % WM), tm_n = FQ_NT / WM + (wm < FQ_NT % WM ? 1 : 0);   // 13 row tiles: 7 + 6, or 4 + 3 + 3 + 3
    const int nk = W / 64, slots = W / 256;
    const unsigned lds_base = (unsigned)(uintptr_t)(lptr_t)smem;
    float* col_c = (float*)(smem + FQ_CONST);                       // [0..191] = ln_g, [192..383] = ln_b of the head's columns
    float* row_c = col_c + 2 * FQ_COLS;                             // [0..207] = rstd, [208..415] = -rstd * mean

    // pairs of this workgroup.  XCD x = blockIdx & 7 owns the images x, x + 8, ... and walks them in passes of HP heads: its per_xcd
    // workgroups work on per_xcd / HP images x HP heads at a time, so that what the XCD's L2 (4 MiB) has to hold is HP heads' weights
    // (HP x 288 KiB) plus per_xcd / HP images' rows (296 KiB each) -- all 12 heads at once (3.4 MiB of weights + the rows) thrashed it:
    // every K-tile then came over the fabric at ~7 TB/s in all, 1.8 us per K-tile whatever the prefetch depth.
    constexpr int HP = 4;
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3, per_xcd = gridDim.x >> 3, ipg = per_xcd / HP;
    const int n_img_x = (B - xcd + 7) / 8, groups = (n_img_x + ipg - 1) / ipg;                // images of this XCD, groups of ipg images
    auto pair_of = [&](int it, int& b, int& h) {                     // wave-uniform; pairs past the end: b >= B
        if constexpr (!HP4) {                                        // image-major: the 12 heads of an image on one XCD, one after the other
            const int q = slot + it * per_xcd;
            b = xcd + 8 * (q / H);
            h = q % H;
            return b < B;
        }
        const int pass = it / max(groups, 1), g = it - pass * max(groups, 1);
        const int k = g * ipg + slot / HP;
        h = pass * HP + slot % HP;
        b = (pass * HP < H && k < n_img_x) ? xcd + 8 * k : B;
        return pass * HP < H;                                        // false: this workgroup has no more pairs at all
    };
    auto next_pair = [&](int& it, int& b, int& h) {                  // the next iteration that holds a pair for this workgroup
        for (;;) {
            ++it;
            if (!pair_of(it, b, h)) return false;
            if (b < B) return true;
        }
    };
    // one K-tile of a pair into a stage: 26 A pieces (8 token rows x 128 B) + 24 B pieces (8 weight rows), dealt to the waves.  A wave's
    // pieces are the same for every K-tile of a pair but for the K offset, so their source addresses and LDS offsets are set up ONCE per
    // pair (the first version recomputed them per K-tile: 2 056 scalar instructions per wave and pair, and the vector multiplies with them)
    constexpr int PMAX = (50 + NWV - 1) / NWV;
    unsigned poff[PMAX];                                             // byte offset of this lane's 16 bytes of piece p inside x / wf, K-tile 0
    auto pieces = [&](int b, int h) {
#pragma unroll
        for (int p = 0; p < PMAX; ++p) {
            const int ins = min(wave + p * NWV, 49);
            if (ins < 26) {
                const int r = min(ins * 8 + srow, L - 1);           // rows past the last token repeat it (finite values, masked as keys, never stored as queries)
                poff[p] = (unsigned)((((long)b * L + r) * W + schunk) * 2);
            } else {
                const int j = (ins - 26) * 8 + srow, part = j >> 6;
                poff[p] = (unsigned)((((long)part * W + h * 64 + (j & 63)) * W + schunk) * 2);
            }
        }
    };
    // (scalar base + 32-bit lane offset form of the DMA, M0 = LDS address of the wave's 1 KiB piece: as gemm_f16_v5.hip issues it)
    auto stage = [&](int st, int kt) {
        const char* ak = (const char*)x + kt * 128;
        const char* wk = (const char*)wf + kt * 128;
#pragma unroll
        for (int p = 0; p < PMAX; ++p) {
            const int ins = wave + p * NWV;
            if (ins >= 50) break;
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + st * FQ_STAGE + (ins < 26 ? ins * 1024 : FQ_A_BYTES + (ins - 26) * 1024));
            if (ins < 26) asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(poff[p]), "s"(ak), "s"(dst) : "memory", "m0");
            else asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %0, %1" :: "v"(poff[p]), "s"(wk), "s"(dst) : "memory", "m0");
        }
    };

    int b, h, it = -1;
    if (!next_pair(it, b, h)) return;
    pieces(b, h);
    stage(0, 0);
    stage(1, 1);
    for (;;) {
        // ---- epilogue constants of this pair (read after the K loop: many barriers later)
        if (tid < FQ_COLS) {
            const int n = (tid >> 6) * W + h * 64 + (tid & 63);
            col_c[tid] = ln_g[n];
            col_c[FQ_COLS + tid] = ln_b[n];
        } else if (tid - FQ_COLS < FQ_ROWS) {
            const int r = tid - FQ_COLS;
            const float2_t* sp = (const float2_t*)stats + ((long)b * L + min(r, L - 1)) * slots;
            float su = 0.f, sq = 0.f;
            for (int sl = 0; sl < slots; ++sl) { const float2_t p = sp[sl]; su += p[0]; sq += p[1]; }
            const float inv_k = 1.0f / (float)W;
            const float mean = su * inv_k;
            const float rstd = 1.0f / sqrtf(fmaxf(sq * inv_k - mean * mean, 0.f) + 1e-5f);
            row_c[r] = rstd;
            row_c[FQ_ROWS + r] = -rstd * mean;
        }
        // ---- phase A: K loop over three stages, two K-tiles in flight
        float4_t acc[TMX][3];
#pragma unroll
        for (int i = 0; i < TMX; ++i)
#pragma unroll
            for (int j = 0; j < 3; ++j) acc[i][j] = (float4_t){0.f, 0.f, 0.f, 0.f};
        if constexpr (PP) {
            // ---- ping-pong form (OVMR_FQ_ABL bit 16): the two wave rows run one slot apart -- while row 0 issues the MFMAs of half-step h,
            // row 1 reads the fragments of h, and vice versa -- so that LDS fragment traffic and matrix-pipe time overlap instead of adding
            // (they are 7.9 and 8.6 us per pair in lockstep).  Two barriers per half-step (X, Y); the K-tile bookkeeping rides on Y of a
            // tile's second half-step: tile kt + 1 is waited for BEFORE it (so Y publishes it), tile kt + 3 is issued AFTER it (every read
            // of tile kt is done) -- three tiles resident or in flight.
            half8_t fa[TMX], fb[3];
            auto R = [&](int hh) {
                const int kt = hh >> 1, ks = hh & 1;
                const char* sA = smem + (kt % 3) * FQ_STAGE;
                const char* sB = sA + FQ_A_BYTES;
#pragma unroll
                for (int i = 0; i < TMX; ++i)
                    if (i < tm_n) fa[i] = *(const half8_t*)(sA + fq_swz((tm0 + i) * 16 + fr, ks * 4 + fg));
#pragma unroll
                for (int j = 0; j < 3; ++j) fb[j] = *(const half8_t*)(sB + fq_swz((wn * 3 + j) * 16 + fr, ks * 4 + fg));
            };
            auto M = [&]() {
#pragma unroll
                for (int i = 0; i < TMX; ++i)
                    if (i < tm_n) {
#pragma unroll
                        for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j], fa[i], acc[i][j], 0, 0, 0);
                    }
            };
            auto wait_tile = [&](int kt) {                          // tile kt + 1 has landed for this wave (tile kt + 2 may still be in flight)
                if (kt + 2 >= nk) __builtin_amdgcn_s_waitcnt(0x0F70);
                else if (wave < 2) __builtin_amdgcn_s_waitcnt(NWV == 8 ? 0x0F77 : 0x0F74);
                else __builtin_amdgcn_s_waitcnt(NWV == 8 ? 0x0F76 : 0x0F73);
            };
            __builtin_amdgcn_s_waitcnt(0x0F70);                     // tiles 0 and 1 (and phase B's stores) are back
            __syncthreads();
            if (2 < nk) stage(2, 2);
            for (int hh = 0; hh < 2 * nk; ++hh) {
                const int kt = hh >> 1, ks = hh & 1;
                if (wm == 0) {
                    R(hh);
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();                   // X
                    __builtin_amdgcn_sched_barrier(0);
                    M();
                } else {
                    if (hh > 0) M();
                    __builtin_amdgcn_sched_barrier(0);
                    __builtin_amdgcn_s_barrier();                   // X
                    __builtin_amdgcn_sched_barrier(0);
                    R(hh);
                    __builtin_amdgcn_s_waitcnt(0xC07F);             // lgkmcnt(0): the reads are back before Y lets anybody overwrite the stage
                }
                if (ks && kt + 1 < nk) wait_tile(kt);
                __builtin_amdgcn_sched_barrier(0);
                __builtin_amdgcn_s_barrier();                       // Y
                __builtin_amdgcn_sched_barrier(0);
                if (ks && kt + 3 < nk) stage(kt % 3, kt + 3);
            }
            if (wm == 1) M();
        } else
        for (int kt = 0; kt < nk; ++kt) {
            // K-tile kt has landed for this wave: everything older than the pieces of tile kt + 1 (7 per wave for waves 0 / 1, 6 for the
            // others) is back.  The first tile of a pair waits for everything: phase B's stores and the constants' loads sit in between.
            if (kt == 0 || kt + 1 >= nk) __builtin_amdgcn_s_waitcnt(0x0F70);          // vmcnt(0)
            else if (wave < 2) __builtin_amdgcn_s_waitcnt(NWV == 8 ? 0x0F77 : 0x0F74); // vmcnt(7 | 4): 50 pieces over 8 | 16 waves
            else __builtin_amdgcn_s_waitcnt(NWV == 8 ? 0x0F76 : 0x0F73);               // vmcnt(6 | 3)
            __syncthreads();                                        // ... everybody's has, and everybody is done with tile kt - 1 (and phase B)
            if (kt + 2 < nk && !(abl & 4)) stage((kt + 2) % 3, kt + 2);
            const char* sA = smem + (kt % 3) * FQ_STAGE;
            const char* sB = sA + FQ_A_BYTES;
#pragma unroll
            for (int ks = 0; ks < 2; ++ks) {
                half8_t fa[TMX], fb[3];
#pragma unroll
                for (int i = 0; i < TMX; ++i)
                    if (i < tm_n) fa[i] = *(const half8_t*)(sA + fq_swz((tm0 + i) * 16 + fr, ks * 4 + fg));
#pragma unroll
                for (int j = 0; j < 3; ++j) fb[j] = *(const half8_t*)(sB + fq_swz((wn * 3 + j) * 16 + fr, ks * 4 + fg));
                if (abl & 2) continue;
#pragma unroll
                for (int i = 0; i < TMX; ++i)
                    if (i < tm_n) {
#pragma unroll
                        for (int j = 0; j < 3; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(fb[j], fa[i], acc[i][j], 0, 0, 0);
                    }
            }
        }
        __syncthreads();                                            // everybody is done with the last K-tiles: the stages may be overwritten
        // ---- epilogue: LayerNorm fold (the arithmetic of gemm_f16_v5's EPI_LN_BIAS), fp16, into the K | V | Q images
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            if (abl & 8) break;
            const int col = (wn * 3 + j) * 16 + fg * 4, part = col >> 6, d = col & 63;     // a 16-column tile lies inside one third
            const float4_t c0 = *(const float4_t*)(col_c + col), c1 = *(const float4_t*)(col_c + FQ_COLS + col);
            // thirds: 0 = Q, 1 = K, 2 = V
            char* img = smem + (part == 0 ? FQ_Q : FQ_KV + (part - 1) * FQ_A_BYTES);
#pragma unroll
            for (int i = 0; i < TMX; ++i) {
                if (i >= tm_n) continue;
                const int row = (tm0 + i) * 16 + fr;
                const float ln_r = row_c[row], ln_t = row_c[FQ_ROWS + row];
                const float4_t v = acc[i][j];
                half4_t o;
#pragma unroll
                for (int r = 0; r < 4; r += 2) {
                    const float2_t t2 = __builtin_elementwise_fma((float2_t){ln_t, ln_t}, (float2_t){c0[r], c0[r + 1]}, (float2_t){c1[r], c1[r + 1]});
                    const float2_t x2 = __builtin_elementwise_fma((float2_t){ln_r, ln_r}, (float2_t){v[r], v[r + 1]}, t2);
                    const half2_t u2 = __builtin_convertvector(x2, half2_t);
                    o[r] = u2[0];
                    o[r + 1] = u2[1];
                }
                *(half4_t*)(img + fq_swz(row, d >> 3) + (d & 7) * 2) = o;
            }
        }
        __syncthreads();
        // ---- the next pair's first two K-tiles stream into S0 / S1 under phase B (S1 once every wave holds its query fragments)
        int nb, nh, nit = it;
        const bool more = next_pair(nit, nb, nh);
        if (more) {
            pieces(nb, nh);                                  // (the K loop of this pair is over: the table now describes the next pair)
            stage(0, 0);
        }
        constexpr int QR = (FQ_NT + NWV - 1) / NWV;          // rounds of query tiles in phase B: 2 with 8 waves, 1 with 16
        half8_t qf[QR][2];
        {
            const char* sQ = smem + FQ_Q;
#pragma unroll
            for (int u = 0; u < QR; ++u) {
                const int qr = min((wave + NWV * u) * 16 + fr, FQ_ROWS - 1);
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) qf[u][ks] = *(const half8_t*)(sQ + fq_swz(qr, ks * 4 + fg));
            }
        }
        __syncthreads();
        if (more) stage(1, 1);
        // ---- phase B: the 13 query tiles of the head, two rounds of the 8 waves
        {
            const half_t* sK = (const half_t*)(smem + FQ_KV);
            const half_t* sV = sK + FQ_ROWS * 64;
            const int D = H * 64;
#pragma unroll
            for (int u = 0; u < QR; ++u) {
                const int t = wave + NWV * u;
                if (t >= FQ_NT || (abl & 1)) break;
                const int qr = t * 16 + fr;
                float4_t o[4];
                const float lsum = attn_sp::tile<FQ_NT, 0>(sK, sV, qf[u], L, scale_log2e, fr, fg, o);
                if (qr < L) attn_sp::store_row(out + ((long)b * L + qr) * D + h * 64, o, 1.0f / lsum, fg);
            }
        }
        if (!more) break;
        it = nit;
        b = nb;
        h = nh;
        // (the K loop's first wait + barrier orders phase B of every wave before S2 -- the K | V images -- is written again)
    }
}

}  // namespace

int launch_qkv_attn_fused(const half_t* x, const half_t* wf, const float* ln_g, const float* ln_b, const float* stats, half_t* out,
                          int B, int L, int W, hipStream_t s) {
    if (L <= 192 || L > 208 || W % 256 || W / 64 * 64 != W) return -100;
    static bool attr_set[OVMR_MAX_DEVICES] = {};
    static int n_cu[OVMR_MAX_DEVICES] = {};
    int dev = 0;
    HIP_CHECK_RET(hipGetDevice(&dev));
    if (dev < 0 || dev >= OVMR_MAX_DEVICES) return -100;
    typedef void (*kern_t)(const half_t*, const half_t*, const float*, const float*, const float*, half_t*, int, int, int, int, float, int);
    static const int abl = getenv("OVMR_FQ_ABL") ? atoi(getenv("OVMR_FQ_ABL")) : 0;
    // bit 16: ping-pong K loop (8 waves); bit 32: 16 waves per workgroup; bit 64: pairs in passes of 4 heads
    const kern_t kern = (abl & 32) ? ((abl & 64) ? (kern_t)qkv_attn_fused_kernel<16, false, true> : (kern_t)qkv_attn_fused_kernel<16, false, false>)
                      : (abl & 16) ? (kern_t)qkv_attn_fused_kernel<8, true, false>
                      : (abl & 64) ? (kern_t)qkv_attn_fused_kernel<8, false, true> : (kern_t)qkv_attn_fused_kernel<8, false, false>;
    if (!attr_set[dev]) {
        for (kern_t k : {(kern_t)qkv_attn_fused_kernel<16, false, true>, (kern_t)qkv_attn_fused_kernel<16, false, false>, (kern_t)qkv_attn_fused_kernel<8, true, false>,
                         (kern_t)qkv_attn_fused_kernel<8, false, true>, (kern_t)qkv_attn_fused_kernel<8, false, false>})
            HIP_CHECK_RET(hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, FQ_LDS));
        HIP_CHECK_RET(hipDeviceGetAttribute(&n_cu[dev], hipDeviceAttributeMultiprocessorCount, dev));
        attr_set[dev] = true;
    }
    const int grid = std::max(8, n_cu[dev] / 8 * 8);
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3((abl & 32) ? 1024 : 512), FQ_LDS, s, x, wf, ln_g, ln_b, stats, out, B, L, W, W / 64,
                       0.125f * 1.4426950408889634f, abl);
    return (int)hipGetLastError();
}

// x [B L, W] fp16 (the residual stream), in_w [3W, W] / in_b [3W] fp16, gamma / beta [W] fp32 (ln_1).  out_ref / out_fused [B L, W] fp16.
// us_out (HOST, 3 floats): in_proj, attention, fused -- mean of `reps` launches each, HIP events on `stream`.
extern "C" int ovmr_debug_qkv_attn(const void* x, const void* in_w, const void* in_b, const float* gamma, const float* beta, void* out_ref,
                                   void* out_fused, int B, int L, int W, int reps, float* us_out, void* stream) {
    hipStream_t s = (hipStream_t)stream;
    if (W % 256 || L < 1 || B < 1 || reps < 1) return -1;
    const int M = B * L, slots = W / 256, H = W / 64;
    float *stats = nullptr, *g = nullptr, *bf = nullptr;
    half_t *wf = nullptr, *qkv = nullptr;
    int rc = 0;
    if (hipMalloc((void**)&stats, (size_t)M * slots * 8) != hipSuccess || hipMalloc((void**)&g, (size_t)3 * W * 4) != hipSuccess ||
        hipMalloc((void**)&bf, (size_t)3 * W * 4) != hipSuccess || hipMalloc((void**)&wf, (size_t)3 * W * W * 2) != hipSuccess ||
        hipMalloc((void**)&qkv, (size_t)M * 3 * W * 2) != hipSuccess)
        rc = -5;
    hipEvent_t e[2];
    hipEventCreate(&e[0]);
    hipEventCreate(&e[1]);
    if (!rc) rc = launch_fold_ln((const half_t*)in_w, gamma, beta, (const half_t*)in_b, wf, g, bf, 3 * W, W, s);
    if (!rc) rc = launch_row_stats((const half_t*)x, stats, M, W, slots, s);
    GemmArgs a;
    memset(&a, 0, sizeof a);
    a.A = x; a.lda = W; a.W = wf; a.ldw = W; a.C = qkv; a.ldc = 3 * W; a.M = M; a.N = 3 * W; a.K = W; a.epi = EPI_LN_BIAS; a.scale = 1.f;
    a.ln_stats = stats; a.ln_slots = slots; a.ln_g = g; a.ln_b = bf;
    auto timed = [&](auto fn, float* us) {
        for (int i = 0; i < 2 && !rc; ++i) rc = fn();
        hipEventRecord(e[0], s);
        for (int i = 0; i < reps && !rc; ++i) rc = fn();
        hipEventRecord(e[1], s);
        hipEventSynchronize(e[1]);
        float ms = 0.f;
        hipEventElapsedTime(&ms, e[0], e[1]);
        *us = ms * 1000.f / reps;
    };
    if (!rc) timed([&] { return launch_gemm_f16(a, 8, s); }, us_out + 0);
    if (!rc) timed([&] { return launch_attention_f16(qkv, (half_t*)out_ref, B, L, H, 0, 3, s); }, us_out + 1);
    if (!rc) timed([&] { return launch_qkv_attn_fused((const half_t*)x, wf, g, bf, stats, (half_t*)out_fused, B, L, W, s); }, us_out + 2);
    (void)hipStreamSynchronize(s);
    hipEventDestroy(e[0]);
    hipEventDestroy(e[1]);
    (void)hipFree(stats); (void)hipFree(g); (void)hipFree(bf); (void)hipFree(wf); (void)hipFree(qkv);
    return rc;
}
