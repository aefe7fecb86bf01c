// Attention variant 5 (fp16, hd = 64, non-causal, L >= 256: the ViT-L towers, L = 257 / 577; clip/model.py:184-188).
//
// Variant 1's flash structure rebuilt around v_mfma_f32_32x32x16_f16:
//   * ONE 32-row query tile per wave, S^T = K Q^T and O^T = V^T P^T as 32x32x16 products: 8 + 8 MFMAs per 64 keys where variant 1
//     issues 36 of the 16x16x32 shape for the same 32 queries (a 16x16x32 MFMA holds the vector issue port for 8 of its 16 cycles, a
//     32x32x16 one for 8 of 32: profiles/r03a_valu_rate.log); the row maximum needs one v_permlane32_swap (a query's scores sit in
//     lanes l and l + 32), the softmax is variant 1's lazily rescaled online form;
//   * the S^T accumulator IS the B operand of the PV product (CDNA4 guide, "An accumulator tile as the next MFMA's operand"):
//     registers 8t .. 8t+7 of a 32-key score tile, converted to fp16, are the P^T fragment of k-step t with the keys in the
//     order 16t + 8(j>>2) + 4h + (j&3); the V^T fragment is fetched in that same order by two transposing reads
//     (ds_read_b64_tr_b16: 4 keys x 16 d per 16-lane group) -- no lane exchange, no LDS round trip for P;
//   * K / V in 64-key blocks through a ring of LDS stages filled by LDS-DMA (inline asm, so that hipcc does not wait for it on its
//     own), a counted vmcnt and one raw barrier per block; the ring is walked with the stage as a compile-time constant (block loop
//     unrolled by the ring depth) and the DMA source pointers advance by a constant: every LDS address is a per-lane constant plus an
//     immediate, no vector address arithmetic per block (it was a third of the loop's vector instructions);
//   * a last block of <= 16 keys (every CLIP ViT: L = G*G + 1) is peeled into a body of 7 MFMAs and 8 exponentials;
//   * LDS images: 128-byte rows; K chunk c of row r in slot c ^ ((r >> 1) & 7) (ds_read_b128 of 32 rows: conflict-free), V
//     chunk c in slot c ^ (((r >> 1) & 1) << 2) (the 4 keys x 4 chunks of a transposing read: conflict-free); applied on the
//     per-lane SOURCE address of the DMA and on the read (SQ_LDS_BANK_CONFLICT = 0);
//   * the output tile leaves through LDS (the epilogue staging of gemm_f16_v5.hip): 8 rows x 128 B per store instruction.
// Measured (profiles/r03t_attn_bench_sched.log, 64 x 16 x 577 / 128 x 16 x 577 / 256 x 16 x 257): 131 / 250 / 151 us = 666 / 698 / 458 TFLOP/s
// against 152 / 290 / 186 us for variant 1.  Neither pipe is the bound (profiles/r03p_pmc_attn_l577.json: matrix pipe 36 % busy, vector ALU
// 64 %, 28 % of the wave cycles in s_waitcnt / barriers); tools/attn_stamps.py shows where a block goes, and attention_v6.hip (two tiles per
// wave at two waves per SIMD, experiment build) that fewer, fatter waves lose: with 124 VGPRs and one tile per wave the kernel lives on
// occupancy -- a fourth wave per SIMD (32 KiB ring) beat a second block in flight (48 KiB ring, 3 waves).
#include "common.h"

#include <algorithm>
#include <type_traits>

namespace {

typedef float float16_t __attribute__((ext_vector_type(16)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef short short4v __attribute__((__vector_size__(8)));

constexpr int KB5 = 64;
constexpr int STAGE5 = 2 * KB5 * 64;                    // halves per stage: 64 K rows, then 64 V rows

// LDS-DMA from inline asm (M0 = LDS byte address of the wave's 1 KiB piece, saved / restored inside the statement), as in
// attention_v3.hip: with the builtin hipcc puts an s_waitcnt vmcnt(0) in front of the first transposing V read of every key block
// (seen in the ISA of this kernel too), which drains the two blocks in flight.  Ordering is by hand: counted wait + barrier at the
// top of the block loop.  (Waits hipcc computes for its own loads ignore these DMAs and can therefore only be too strict.)
// (scalar base + 32-bit per-lane byte offset: the block walk is one scalar add, the per-lane offsets never change)
__device__ __forceinline__ void glds16_asm5(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

__device__ __forceinline__ half4_t tr_read5(const half_t* p) {
    short4v r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)p);
    return __builtin_bit_cast(half4_t, r);
}

#ifdef OVMR_EXPERIMENTS
// SCH & 8: shader-clock stamps of one workgroup (tools/attn_stamps.py): eight per key block and wave
__device__ long long g_attn5_stamps[4 * 256];
#define A5_STAMP()                                                                     \
    if constexpr ((SCH & 8) != 0) {                                                    \
        if (stamp_on && stamp_i < 256) {                                               \
            const long long t_ = __builtin_amdgcn_s_memtime();                         \
            if (lane == 0) g_attn5_stamps[wave * 256 + stamp_i] = t_;                  \
            ++stamp_i;                                                                 \
        }                                                                              \
        __builtin_amdgcn_sched_barrier(0);                                             \
    }
#else
#define A5_STAMP()
#endif

// STAGES5: LDS ring depth; RS_MFMA: row sums from the matrix pipe (ones . P^T, 16 more registers) or 32 v_add_f32 per block.  Measured
// (r03p; 64 x 16 x 577 / 128 x 16 x 577 / 256 x 16 x 257, us): <3, MFMA> 142 / 276 / 168, <3, VALU> 144 / 281 / 167, <2, VALU> 140 / 263 / 157,
// <2, MFMA> 146 / 275 / 170.  <2, false> is what the launcher runs.
// NW: waves (32-row query tiles) per workgroup, 3 or 4: the launcher takes the one that wastes fewer wave slots (L = 257: 9 tiles = 3 x 3).
// SCH (experiment build): 1 = the two score chains interleaved, 2 = the next block's DMA issued behind the score MFMAs.
// FOLD: the exponent's multiply-add folded into the score product itself.  Q is scaled by scale * log2(e) once per tile (fp16: one more
// rounding of 2^-11 per element, below the reference's own rounding of every score to fp16, clip/model.py:184-188 through
// nn.MultiheadAttention), and the C operand of the first score MFMA is a 16-register tile holding -m_run in every register (a lane's
// 16 scores belong to ONE query), so the matrix pipe hands back t - m_run and the exponential reads it directly: 32 v_fma_f32 per
// 64-key block and wave less (attention table).  The reference m_run starts at the first block's row maximum exactly as
// before; the rare rescale (a row maximum more than 8 above the reference) also re-bases the live scores and rewrites the tile.
// OCC: waves per SIMD the register budget is set for (4: 128 registers; FOLD needs ~140, so its tile costs the fourth wave or spills).
template <int STAGES5, bool RS_MFMA, int NW, int SCH, bool FOLD = false, int OCC = (STAGES5 == 3 ? 3 : 4), bool HALF = false, int PRIO = 0>
__global__ __launch_bounds__(NW * 64, OCC) void attn_f16_v5(const half_t* __restrict__ qkv, half_t* __restrict__ out,
                                                                           int L, int Lq, int H, int nT, int nWG, int nBH, float scale_log2e) {
    static_assert(!HALF || !RS_MFMA, "the half-block body keeps its row sums on the vector ALU");
    static_assert(STAGES5 == 2 || NW == 4, "the counted wait of the three-stage ring assumes four DMA instructions per wave");
    __shared__ __attribute__((aligned(16))) half_t smem[STAGES5 * STAGE5];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int D = H * 64, ld = 3 * D;
    // the workgroups of one (sequence, head) get block ids congruent mod 8: one XCD, its L2 serves their K / V re-reads
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int wg = slot % nWG, bh = (slot / nWG) * 8 + xcd;
    if (bh >= nBH) return;
    const int hh = bh % H, b = bh / H;
    const half_t* base = qkv + (long)b * L * ld + hh * 64;

    [[maybe_unused]] const bool stamp_on = (SCH & 8) && blockIdx.x == (gridDim.x / 16) * 8;      // a mid-grid workgroup, its first query tiles
    [[maybe_unused]] int stamp_i = 0;
    const int qt = wg * NW + wave;                         // this wave's 32-row query tile
    const bool act = qt < nT;                              // wave-uniform
    const int q = qt * 32 + r;
    half8_t qf[4];
    {
        const int qc = min(q, L - 1);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[ks] = *(const half8_t*)(base + (long)qc * ld + ks * 16 + h * 8);
        if (FOLD) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks)
#pragma unroll
                for (int j = 0; j < 8; ++j) qf[ks][j] = (half_t)((float)qf[ks][j] * scale_log2e);
        }
    }

    // staging: 16 LDS-DMA instructions of 8 rows x 128 B per block (8 of K rows, then 8 of V rows), dealt round-robin to the waves.  The
    // per-lane byte offsets are set up once; the block walk is a scalar base; only a block that reaches past the last key clamps its rows.
    constexpr int NI = (16 + NW - 1) / NW;
    const int srow = lane >> 3, sslot = lane & 7;
    const unsigned lds_base = (unsigned)(uintptr_t)(lptr_t)smem;
    unsigned soff[NI], sdst[NI];
    int srow_t[NI];
#pragma unroll
    for (int t = 0; t < NI; ++t) {
        const int ins = wave + NW * t, isv = (ins >> 3) & 1, r0 = (ins & 7) * 8, row = r0 + srow;
        const int swz = isv ? (((row >> 1) & 1) << 2) : ((row >> 1) & 7);
        srow_t[t] = row;
        soff[t] = 2u * (unsigned)((1 + isv) * D + row * ld + ((sslot ^ swz) << 3));
        sdst[t] = lds_base + 2u * (unsigned)(isv * (KB5 * 64) + r0 * 64);
    }
    const unsigned row_bytes = 2u * (unsigned)ld;
    auto stage = [&](int st, int kb_) {
        const int k0 = kb_ * KB5;
        const half_t* kbase = base + (long)k0 * ld;          // wave-uniform
        if (k0 + KB5 <= L) {
#pragma unroll
            for (int t = 0; t < NI; ++t)
                if (NW * (t + 1) <= 16 || wave + NW * t < 16) glds16_asm5(kbase, soff[t], __builtin_amdgcn_readfirstlane(sdst[t] + 2u * (unsigned)(st * STAGE5)));
        } else {                                           // the last block: rows past the last key repeat it (finite values, masked below)
#pragma unroll
            for (int t = 0; t < NI; ++t)
                if (NW * (t + 1) <= 16 || wave + NW * t < 16)
                    glds16_asm5(kbase, soff[t] - (unsigned)max(k0 + srow_t[t] - (L - 1), 0) * row_bytes,
                                __builtin_amdgcn_readfirstlane(sdst[t] + 2u * (unsigned)(st * STAGE5)));
        }
    };

    float m_run = FOLD ? 0.f : -INFINITY;                  // reference maximum of the exponentials (scaled domain), variant 1's lazy form
    [[maybe_unused]] float thr = -INFINITY, dfloor = -INFINITY;   // FOLD: rescale threshold and lower bound of the reference shift (first block: none)
    float16_t cm;                                          // FOLD: -m_run in every register: the C operand of the first score MFMA of a chain
    if (FOLD) {
#pragma unroll
        for (int k = 0; k < 16; ++k) cm[k] = 0.f;
    }
    float16_t o[2];
#pragma unroll
    for (int k = 0; k < 16; ++k) { o[0][k] = 0.f; o[1][k] = 0.f; }
    float16_t ol;                                          // RS_MFMA: ones . P^T, every row holds the row sums
    float lsum = 0.f;                                      // !RS_MFMA: this lane's share of the row sum (lanes l and l + 32 are added at the end)
    half8_t ones;
    if (RS_MFMA) {
#pragma unroll
        for (int k = 0; k < 16; ++k) ol[k] = 0.f;
#pragma unroll
        for (int j = 0; j < 8; ++j) ones[j] = (half_t)1.f;
    }

    // per-lane LDS offsets (halves).  K fragment of k-step ks: row r, logical chunk 2 ks + h.  V^T fragment of d block blk,
    // k-step t: the 16-lane group g = lane >> 4 serves d columns blk*32 + (g & 1)*16 + [0,16) and the keys 16 t + 4 (g >> 1) + [0,4)
    // (second read: + 8); inside a group lane 4 qq + p addresses key row qq, columns 4 p .. 4 p + 3.
    // Every LDS read address = stage offset (one v_add per base and block) + a per-lane constant + an immediate: the chunk index
    // (2 ks + h) ^ kswz equals (2 ks) ^ (h ^ kswz), four per-lane values; the V chunk blk*4 ^ ... two.
    const int kswz = (r >> 1) & 7;
    int koff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) koff[ks] = r * 64 + (((2 * ks) ^ h ^ kswz) << 3);
    const int g = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
    const int vkey = 4 * (g >> 1) + qq;                    // + 16 t (+ 8): (key >> 1) & 1 = (qq >> 1) & 1 for every one of them
    const int vswz = ((qq >> 1) & 1) << 2;
    int voff[2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
        voff[blk] = KB5 * 64 + vkey * 64 + (((blk * 4 + (g & 1) * 2 + (pp >> 1)) ^ vswz) << 3) + (pp & 1) * 4;

    const int nb = (L + KB5 - 1) / KB5;
    // the query rows must be back BEFORE the first DMA is issued, and hipcc's own scoreboard must know it (the builtin form of the
    // wait): otherwise it sinks counted waits for them into the block loop, where they would drain the DMAs of every iteration
    __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int p = 0; p < STAGES5 - 1; ++p)
        if (p < nb) stage(p, p);
    // The ring is walked with the stage as a compile-time constant (the block loop is unrolled by three): every LDS read address is then
    // a per-lane constant plus an immediate, no vector address arithmetic per block (r03m: it was a third of the loop's VALU instructions).
    auto block = [&](auto stage_c, int kb) {
        constexpr int ST = decltype(stage_c)::value;
        // block kb has landed for THIS wave's four DMA instructions (block kb + 1 may stay in flight) ...
        A5_STAMP()                                         // 0: block top
        if (STAGES5 == 3 && kb + 1 < nb) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                      // ... and for every wave; every wave is done with block kb - 1
        __builtin_amdgcn_sched_barrier(0);
        A5_STAMP()                                         // 1: through the barrier
        const bool more = kb + STAGES5 - 1 < nb;           // the next DMA goes into the stage block kb - 1 occupied
        if (more && (!(SCH & 2) || !act)) stage((ST + STAGES5 - 1) % STAGES5, kb + STAGES5 - 1);
        if (!act) return;
        A5_STAMP()                                         // 2: DMA issued
        constexpr int st_off = ST * STAGE5;
        const int nvalid = min(L - kb * KB5, KB5);
        float16_t zero16;
#pragma unroll
        for (int k = 0; k < 16; ++k) zero16[k] = 0.f;      // (folds into the MFMA's inline constant 0: no accumulator zeroing)
        if constexpr (HALF) {
            // One 32-key score tile at a time: 4 score MFMAs, the row maximum of ITS 16 registers, the (lazy) reference check, 16
            // exponentials, two PV steps -- then the second half.  Only one score tile is live (16 registers less), which is what lets
            // FOLD's -m_run tile in without costing the fourth wave per SIMD; the rescale decision is taken per 32 keys.
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) {
                float16_t sc;
                if (FOLD) __builtin_amdgcn_sched_barrier(0);      // (the second half's score tile must not be hoisted over the first half's softmax: registers)
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const half8_t kf = *(const half8_t*)(smem + koff[ks] + (st_off + sb * 32 * 64));
                    sc = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], ks == 0 ? (FOLD ? cm : zero16) : sc, 0, 0, 0);
                }
                if (nvalid < KB5) {
                    const int thr_k = nvalid - sb * 32 - 4 * h;
#pragma unroll
                    for (int k = 0; k < 16; ++k) sc[k] = ((k & 3) + 8 * (k >> 2) < thr_k) ? sc[k] : -INFINITY;
                }
                float mx = sc[0];
#pragma unroll
                for (int k = 1; k < 16; ++k) mx = fmaxf(mx, sc[k]);
                {
                    const unsigned u = __builtin_bit_cast(unsigned, mx);
                    auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                    mx = fmaxf(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1]));
                }
                float m_ref;
                if constexpr (FOLD) {
                    if (__builtin_amdgcn_ballot_w64(mx > thr) != 0) {
                        const float delta = fmaxf(mx, dfloor);
                        const float alpha = fminf(__builtin_amdgcn_exp2f(-delta), 3.0e38f);
                        m_run += delta;
#pragma unroll
                        for (int k = 0; k < 16; ++k) { o[0][k] *= alpha; o[1][k] *= alpha; }
                        lsum *= alpha;
#pragma unroll
                        for (int k = 0; k < 16; ++k) { sc[k] -= delta; cm[k] = -m_run; }
                    }
                    thr = 8.0f;
                    dfloor = 0.f;
                    m_ref = 0.f;
                } else {
                    const float mxs = mx * scale_log2e;
                    if (__builtin_amdgcn_ballot_w64(mxs > m_run + 8.0f) != 0) {
                        const float m_new = fmaxf(m_run, mxs);
                        const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                        m_run = m_new;
#pragma unroll
                        for (int k = 0; k < 16; ++k) { o[0][k] *= alpha; o[1][k] *= alpha; }
                        lsum *= alpha;
                    }
                    m_ref = m_run;
                }
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const int st = sb * 2 + t;
                    half8_t pf;
#pragma unroll
                    for (int j = 0; j < 8; j += 2) {
                        const float2_t e2 = {FOLD ? __builtin_amdgcn_exp2f(sc[8 * t + j]) : __builtin_amdgcn_exp2f(__builtin_fmaf(sc[8 * t + j], scale_log2e, -m_ref)),
                                             FOLD ? __builtin_amdgcn_exp2f(sc[8 * t + j + 1]) : __builtin_amdgcn_exp2f(__builtin_fmaf(sc[8 * t + j + 1], scale_log2e, -m_ref))};
                        lsum += e2[0] + e2[1];
                        const half2_t p2 = __builtin_convertvector(e2, half2_t);
                        pf[j] = p2[0];
                        pf[j + 1] = p2[1];
                    }
#pragma unroll
                    for (int blk = 0; blk < 2; ++blk) {
                        const half4_t v0 = tr_read5(smem + voff[blk] + (st_off + st * 16 * 64));
                        const half4_t v1 = tr_read5(smem + voff[blk] + (st_off + (st * 16 + 8) * 64));
                        const half8_t vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                        o[blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, o[blk], 0, 0, 0);
                    }
                }
            }
            return;
        }
        float16_t s[2];
        if (PRIO & 1) __builtin_amdgcn_s_setprio(1);       // PRIO 1: the score chain issues ahead of the other waves' softmax; 2: the PV phase too
        if (SCH & 1) {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
                for (int sb = 0; sb < 2; ++sb) {
                    const half8_t kf = *(const half8_t*)(smem + koff[ks] + (st_off + sb * 32 * 64));
                    s[sb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], ks == 0 ? (FOLD ? cm : zero16) : s[sb], 0, 0, 0);
                }
            }
        } else {
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {
                    const half8_t kf = *(const half8_t*)(smem + koff[ks] + (st_off + sb * 32 * 64));
                    s[sb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], ks == 0 ? (FOLD ? cm : zero16) : s[sb], 0, 0, 0);
                }
            }
        }
        if (SCH & 2) {
            __builtin_amdgcn_sched_barrier(0);
            if (more) stage((ST + STAGES5 - 1) % STAGES5, kb + STAGES5 - 1);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (PRIO == 1) __builtin_amdgcn_s_setprio(0);
        A5_STAMP()                                         // 3: score MFMAs issued
        // register k of a score tile holds key (k & 3) + 8 (k >> 2) + 4 h of its 32-key half
        if (nvalid < KB5) {                                // (wave-uniform) the last block: keys >= L get -inf, i.e. p = 0
#pragma unroll
            for (int sb = 0; sb < 2; ++sb) {
                const int thr = nvalid - sb * 32 - 4 * h;
#pragma unroll
                for (int k = 0; k < 16; ++k) s[sb][k] = ((k & 3) + 8 * (k >> 2) < thr) ? s[sb][k] : -INFINITY;
            }
        }
        float mx = s[0][0];
#pragma unroll
        for (int sb = 0; sb < 2; ++sb)
#pragma unroll
            for (int k = 0; k < 16; ++k) mx = fmaxf(mx, s[sb][k]);
        {
            const unsigned u = __builtin_bit_cast(unsigned, mx);
            auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            mx = fmaxf(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1]));
        }
        A5_STAMP()                                         // 4: row maxima (the scores are back)
        float m_ref;
        if constexpr (FOLD) {
            // the scores arrive as t - m_run: a row needs a new reference when its block maximum exceeds 8; the first block sets the
            // reference to its row maximum whatever its sign (so that every row keeps an exponential equal to 1)
            // (thr / dfloor: -inf in the first block, 8 / 0 afterwards -- wave-uniform values instead of a second copy of the block body)
            if (__builtin_amdgcn_ballot_w64(mx > thr) != 0) {
                const float delta = fmaxf(mx, dfloor);
                const float alpha = fminf(__builtin_amdgcn_exp2f(-delta), 3.0e38f);   // (first block: o = 0 and exp2(-delta) may overflow: 0 * finite)
                m_run += delta;
#pragma unroll
                for (int k = 0; k < 16; ++k) { o[0][k] *= alpha; o[1][k] *= alpha; }
                if (RS_MFMA) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) ol[k] *= alpha;
                } else lsum *= alpha;
#pragma unroll
                for (int k = 0; k < 16; ++k) { s[0][k] -= delta; s[1][k] -= delta; cm[k] = -m_run; }
            }
            thr = 8.0f;
            dfloor = 0.f;
            m_ref = 0.f;
        } else {
        const float mxs = mx * scale_log2e;
        if (__builtin_amdgcn_ballot_w64(mxs > m_run + 8.0f) != 0) {      // wave-uniform: some row needs a new reference
            const float m_new = fmaxf(m_run, mxs);
            const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);   // <= 0 (first block: -inf -> 0)
            m_run = m_new;
#pragma unroll
            for (int k = 0; k < 16; ++k) { o[0][k] *= alpha; o[1][k] *= alpha; }
            if (RS_MFMA) {
#pragma unroll
                for (int k = 0; k < 16; ++k) ol[k] *= alpha;
            } else lsum *= alpha;
        }
        m_ref = m_run;
        }
        A5_STAMP()                                         // 5: reference settled
#pragma unroll
        for (int st = 0; st < 4; ++st) {                   // 16-key PV steps: registers 8 t .. 8 t + 7 of half sb = st >> 1, t = st & 1
            half8_t pf;
#pragma unroll
            for (int j = 0; j < 8; j += 2) {               // pairs: v_cvt_pk_f16_f32 (element-wise casts came out as quarter-rate v_fma_mixlo_f16)
                const float2_t e2 = {FOLD ? __builtin_amdgcn_exp2f(s[st >> 1][8 * (st & 1) + j])
                                          : __builtin_amdgcn_exp2f(__builtin_fmaf(s[st >> 1][8 * (st & 1) + j], scale_log2e, -m_ref)),
                                     FOLD ? __builtin_amdgcn_exp2f(s[st >> 1][8 * (st & 1) + j + 1])
                                          : __builtin_amdgcn_exp2f(__builtin_fmaf(s[st >> 1][8 * (st & 1) + j + 1], scale_log2e, -m_ref))};
                if (!RS_MFMA) lsum += e2[0] + e2[1];
                const half2_t p2 = __builtin_convertvector(e2, half2_t);
                pf[j] = p2[0];
                pf[j + 1] = p2[1];
            }
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const half4_t v0 = tr_read5(smem + voff[blk] + (st_off + st * 16 * 64));
                const half4_t v1 = tr_read5(smem + voff[blk] + (st_off + (st * 16 + 8) * 64));
                const half8_t vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                o[blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, o[blk], 0, 0, 0);
            }
            if (RS_MFMA) ol = __builtin_amdgcn_mfma_f32_32x32x16_f16(ones, pf, ol, 0, 0, 0);
            if (st == 1) { A5_STAMP() }                    // 6: half of the PV steps issued
        }
        if (PRIO & 2) __builtin_amdgcn_s_setprio(0);
        A5_STAMP()                                         // 7: block end
    };
    // A last block of at most 16 keys -- every CLIP ViT has one: L = G*G + 1 leaves ONE key behind the last full block -- is peeled
    // into its own body: one 32-key score tile of which registers 0..7 (keys 0..15) are live, 8 exponentials, one PV step -- 7 MFMAs
    // and 8 exponentials where the full body spends 20 and 32 (a tenth of the kernel at L = 577, a fifth at L = 257).
    const int tail = L - (nb - 1) * KB5;
    const bool peel = nb > 1 && tail <= 16;
    const int nb_main = peel ? nb - 1 : nb;
    for (int kb = 0; kb < nb_main; kb += STAGES5) {
        block(std::integral_constant<int, 0>{}, kb);
        if (kb + 1 < nb_main) block(std::integral_constant<int, 1>{}, kb + 1);
        if (STAGES5 == 3 && kb + 2 < nb_main) block(std::integral_constant<int, STAGES5 - 1>{}, kb + 2);
    }
    if (peel) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        if (act) {
            const int st_off = ((nb - 1) % STAGES5) * STAGE5;
            float16_t zero16;
#pragma unroll
            for (int k = 0; k < 16; ++k) zero16[k] = 0.f;
            float16_t s0;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const half8_t kf = *(const half8_t*)(smem + koff[ks] + st_off);
                s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[ks], ks == 0 ? (FOLD ? cm : zero16) : s0, 0, 0, 0);
            }
            const int thr_k = tail - 4 * h;                // registers 0..7: keys (k & 3) + 8 (k >> 2) + 4 h  (NOT `thr`: that is the FOLD
            float mx = -INFINITY;                          //  variant's float rescale threshold, which the test below must see)
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                s0[k] = ((k & 3) + 8 * (k >> 2) < thr_k) ? s0[k] : -INFINITY;
                mx = fmaxf(mx, s0[k]);
            }
            {
                const unsigned u = __builtin_bit_cast(unsigned, mx);
                auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                mx = fmaxf(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1]));
            }
            float m_ref;
            if constexpr (FOLD) {
                if (__builtin_amdgcn_ballot_w64(mx > thr) != 0) {
                    const float delta = fmaxf(mx, dfloor);
                    const float alpha = fminf(__builtin_amdgcn_exp2f(-delta), 3.0e38f);
                    m_run += delta;
#pragma unroll
                    for (int k = 0; k < 16; ++k) { o[0][k] *= alpha; o[1][k] *= alpha; }
                    if (RS_MFMA) {
#pragma unroll
                        for (int k = 0; k < 16; ++k) ol[k] *= alpha;
                    } else lsum *= alpha;
#pragma unroll
                    for (int k = 0; k < 8; ++k) s0[k] -= delta;
                }
                m_ref = 0.f;
            } else {
            const float mxs = mx * scale_log2e;
            if (__builtin_amdgcn_ballot_w64(mxs > m_run + 8.0f) != 0) {
                const float m_new = fmaxf(m_run, mxs);
                const float alpha = __builtin_amdgcn_exp2f(m_run - m_new);
                m_run = m_new;
#pragma unroll
                for (int k = 0; k < 16; ++k) { o[0][k] *= alpha; o[1][k] *= alpha; }
                if (RS_MFMA) {
#pragma unroll
                    for (int k = 0; k < 16; ++k) ol[k] *= alpha;
                } else lsum *= alpha;
            }
            m_ref = m_run;
            }
            half8_t pf;
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                const float2_t e2 = {FOLD ? __builtin_amdgcn_exp2f(s0[j]) : __builtin_amdgcn_exp2f(__builtin_fmaf(s0[j], scale_log2e, -m_ref)),
                                     FOLD ? __builtin_amdgcn_exp2f(s0[j + 1]) : __builtin_amdgcn_exp2f(__builtin_fmaf(s0[j + 1], scale_log2e, -m_ref))};
                if (!RS_MFMA) lsum += e2[0] + e2[1];
                const half2_t p2 = __builtin_convertvector(e2, half2_t);
                pf[j] = p2[0];
                pf[j + 1] = p2[1];
            }
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const half4_t v0 = tr_read5(smem + voff[blk] + st_off);
                const half4_t v1 = tr_read5(smem + voff[blk] + (st_off + 8 * 64));
                const half8_t vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                o[blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, o[blk], 0, 0, 0);
            }
            if (RS_MFMA) ol = __builtin_amdgcn_mfma_f32_32x32x16_f16(ones, pf, ol, 0, 0, 0);
        }
    }

    // ---------------------------------------------------------------- output: O^T / row sum -> fp16 -> LDS -> 128-byte rows
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // every wave is done with the K / V stages: they become staging tiles
    if (!act) return;
    char* et = (char*)smem + wave * 4096;                  // this wave's 32 rows x 128 B
    float inv;
    if (RS_MFMA) inv = 1.0f / ol[0];
    else {
        const unsigned u = __builtin_bit_cast(unsigned, lsum);
        auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        inv = 1.0f / (__builtin_bit_cast(float, (unsigned)sw[0]) + __builtin_bit_cast(float, (unsigned)sw[1]));
    }
    // register k of o[blk] holds d = blk*32 + (k & 3) + 8 (k >> 2) + 4 h: 8-byte unit u = blk*8 + 2 (k >> 2) + h of row r, stored at
    // u ^ f(r & 15), f(x) = ((x & 7) << 1) | (x >> 3) (gemm_f16_v5.hip epilogue: conflict-free 8-byte writes and 16-byte reads)
    const int fsw = ((r & 7) << 1) | ((r >> 3) & 1);
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
#pragma unroll
        for (int kq = 0; kq < 4; ++kq) {
            const half4_t w = {(half_t)(o[blk][4 * kq] * inv), (half_t)(o[blk][4 * kq + 1] * inv), (half_t)(o[blk][4 * kq + 2] * inv),
                               (half_t)(o[blk][4 * kq + 3] * inv)};
            *(half4_t*)(et + r * 128 + (((blk * 8 + 2 * kq + h) ^ fsw) << 3)) = w;
        }
    const int er = lane >> 3, ec = lane & 7;
#pragma unroll
    for (int it = 0; it < 4; ++it) {
        const int row = it * 8 + er;                       // (row & 15) >> 3 = it & 1: the halves of a chunk trade places in rows 8..15
        half8_t v = *(const half8_t*)(et + row * 128 + ((ec ^ (row & 7)) << 4));
        if (it & 1) v = (half8_t){v[4], v[5], v[6], v[7], v[0], v[1], v[2], v[3]};
        const int qrow = qt * 32 + row;
        if (qrow < Lq) *(half8_t*)(out + ((long)b * Lq + qrow) * D + hh * 64 + ec * 8) = v;
    }
}

}  // namespace

namespace {
template <int NW, int SCH, bool FOLD = false, int STAGES = 2, int OCC = 4, bool HALF = false, int PRIO = 0>
int launch_v5(const half_t* qkv, half_t* out, int B, int L, int Lq, int H, hipStream_t s) {
    const int nT = (Lq + 31) / 32, nWG = (nT + NW - 1) / NW, nBH = B * H;
    const float sl2e = 0.125f * 1.4426950408889634f;
    const dim3 grid((unsigned)((long)((nBH + 7) / 8) * 8 * nWG));
    hipLaunchKernelGGL((attn_f16_v5<STAGES, false, NW, SCH, FOLD, OCC, HALF, PRIO>), grid, dim3(NW * 64), 0, s, qkv, out, L, Lq, H, nT, nWG, nBH, sl2e);
    return (int)hipGetLastError();
}
}  // namespace

// -100: shape not taken (causal, short sequences, a handful of query rows): the caller falls back to variant 1.
// mode (experiment build): bits 0-1 = SCH, bit 2 = force four waves per workgroup, bit 3 = stamps (ovmr_debug_attn5_stamps).
int launch_attention_f16_v5(const half_t* qkv, half_t* out, int B, int L, int Lq, int H, int causal, int mode, hipStream_t s) {
    if (causal || L < 256 || Lq < 32) return -100;
    const int nT = (Lq + 31) / 32;
    const bool three = ((nT + 2) / 3) * 3 < ((nT + 3) / 4) * 4 && !(mode & 4);      // fewer idle wave slots with 3-wave workgroups
#ifdef OVMR_EXPERIMENTS
    if (mode & 128) return (mode & 1) ? (three ? launch_v5<3, 0, false, 2, 4, false, 3>(qkv, out, B, L, Lq, H, s) : launch_v5<4, 0, false, 2, 4, false, 3>(qkv, out, B, L, Lq, H, s))
                                      : (three ? launch_v5<3, 0, false, 2, 4, false, 1>(qkv, out, B, L, Lq, H, s) : launch_v5<4, 0, false, 2, 4, false, 1>(qkv, out, B, L, Lq, H, s));
    if (mode & 64) return (mode & 16) ? (three ? launch_v5<3, 0, true, 2, 4, true>(qkv, out, B, L, Lq, H, s) : launch_v5<4, 0, true, 2, 4, true>(qkv, out, B, L, Lq, H, s))
                                      : (three ? launch_v5<3, 0, false, 2, 4, true>(qkv, out, B, L, Lq, H, s) : launch_v5<4, 0, false, 2, 4, true>(qkv, out, B, L, Lq, H, s));
    if ((mode & 48) == 48) return (mode & 1) ? launch_v5<4, 0, true, 3, 3>(qkv, out, B, L, Lq, H, s)                               // folded, 3 waves / SIMD: 3-stage ring (4-wave workgroups)
                                             : (three ? launch_v5<3, 0, true, 2, 3>(qkv, out, B, L, Lq, H, s) : launch_v5<4, 0, true, 2, 3>(qkv, out, B, L, Lq, H, s));
    if ((mode & 16) && (mode & 3) == 1) return three ? launch_v5<3, 1, true>(qkv, out, B, L, Lq, H, s) : launch_v5<4, 1, true>(qkv, out, B, L, Lq, H, s);
    if ((mode & 16) && (mode & 3) == 3) return three ? launch_v5<3, 3, true>(qkv, out, B, L, Lq, H, s) : launch_v5<4, 3, true>(qkv, out, B, L, Lq, H, s);
    if (mode & 16) return three ? launch_v5<3, 0, true>(qkv, out, B, L, Lq, H, s) : launch_v5<4, 0, true>(qkv, out, B, L, Lq, H, s);
    switch (mode & 3) {
        case 1: return three ? launch_v5<3, 1>(qkv, out, B, L, Lq, H, s) : launch_v5<4, 1>(qkv, out, B, L, Lq, H, s);
        case 2: return three ? launch_v5<3, 2>(qkv, out, B, L, Lq, H, s) : launch_v5<4, 2>(qkv, out, B, L, Lq, H, s);
        case 3: return three ? launch_v5<3, 3>(qkv, out, B, L, Lq, H, s) : launch_v5<4, 3>(qkv, out, B, L, Lq, H, s);
        default: break;
    }
    if (mode & 8) return launch_v5<4, 8>(qkv, out, B, L, Lq, H, s);
#endif
    return three ? launch_v5<3, 0>(qkv, out, B, L, Lq, H, s) : launch_v5<4, 0>(qkv, out, B, L, Lq, H, s);
}

#ifdef OVMR_EXPERIMENTS
extern "C" int ovmr_debug_attn5_stamps(long long* host_out, int n) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_attn5_stamps), (size_t)std::min(n, 4 * 256) * sizeof(long long));
}
#endif
