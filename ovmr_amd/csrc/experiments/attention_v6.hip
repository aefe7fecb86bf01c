// Attention variant 6 (fp16, hd = 64, non-causal, L >= 256): variant 5's arithmetic with TWO 32-row query tiles per wave.
//
// What the stamps of variant 5 show (profiles/r03t_attn_stamps_l577.log: ~4000 cycles per key block and wave at four waves per SIMD):
// a fifth of a block is the wait at the workgroup barrier, a tenth the issue of the wave's four LDS-DMA instructions, and the matrix
// and the vector phases of a wave follow each other (score MFMAs -> row maxima -> exponentials with the PV MFMAs among them), so the
// overlap of the two pipes is left to chance meetings of the four waves of a SIMD.  tools/coissue.hip: ONE wave that alternates a
// 32x32x16 MFMA with a few vector instructions runs both pipes at once (the MFMA holds the issue port for 8 of its 32 cycles).
// Here a wave owns two query tiles A and B and one barrier, one set of DMA instructions and one walk over the K / V stage serve
// both; the block body is written as ONE interleaved stream: while tile A's softmax runs on the vector pipe the matrix pipe works
// on tile B's scores, then on A's PV products under B's softmax:
//     S_A (8 MFMAs), S_B 0-3 | rowmax A | S_B 4 | reference A | S_B 5 | exp A 0-5 with S_B 6, 7 | exp A 6-15 with PV_A steps 0, 1 |
//     rowmax B | reference B | exp B 0-5 with PV_A steps 2, 3 | exp B 6-15 with PV_B steps 0, 1 | PV_B steps 2, 3
// (sched_barriers keep the MFMAs where they are written; LDS reads may move across them).  Two waves per SIMD (<= 256 registers);
// NW = 4 waves per workgroup share a K / V block among 256 queries (half the L2 -> LDS traffic of variant 5), NW = 2 among 128.
// Numerics: the same operations in the same order per tile as variant 5 (lazily rescaled online softmax, fp16 P, fp32 sums).
//
// MEASURED AND NOT ADOPTED (profiles/r03t_attn_bench_v6.log; passes the variant-5 tests): 64 x 16 x 577 / 128 x 16 x 577 / 256 x 16 x 257 run
// in 149 / 287 / 175 us with two-wave workgroups and 158-166 / 306 / 227-232 us with four-wave ones, against 131 / 253-260 / 153-158 us
// for variant 5 -- and the shared-fragment schedule (SH: half the LDS reads) times exactly like the interleaved one.  So neither the LDS
// traffic nor the order of a wave's own instructions is what holds variant 5 back; four resident waves per SIMD with one tile each
// issue vector work faster (tools/valu_rate.hip: v_exp_f32 6.3 cycles per SIMD at four waves, 8.3 at two; v_fma_f32 1.9 / 2.6) and
// cover each other's waits better than two waves with two tiles.  Experiment build only (variants 60 + mode).
#ifdef OVMR_EXPERIMENTS
#include "../common.h"

#include <algorithm>
#include <type_traits>

namespace {

typedef float float16_t __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) void* lptr_t;
typedef short short4v __attribute__((__vector_size__(8)));

constexpr int KB6 = 64;
constexpr int STAGE6 = 2 * KB6 * 64;                    // halves per stage: 64 K rows, then 64 V rows

__device__ __forceinline__ void glds16_asm6(const void* sbase, unsigned voff, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

__device__ __forceinline__ half4_t tr_read6(const half_t* p) {
    short4v r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)p);
    return __builtin_bit_cast(half4_t, r);
}

template <int N> using ic = std::integral_constant<int, N>;
// nothing but LDS reads and scalar instructions moves across: the MFMAs stay between the vector chunks they are written between
#define PIN6() __builtin_amdgcn_sched_barrier(0x0104)

// SH: the K and the V^T fragments are read from LDS ONCE per block and serve both tiles (score and PV MFMAs of A and B in pairs)
template <int NW, int SH>
__global__ __launch_bounds__(NW * 64, 2) void attn_f16_v6(const half_t* __restrict__ qkv, half_t* __restrict__ out,
                                                          int L, int Lq, int H, int nT, int nWG, int nBH, float scale_log2e) {
    __shared__ __attribute__((aligned(16))) half_t smem[2 * STAGE6];

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

    const int qt0 = (wg * NW + wave) * 2;                  // this wave's query tiles: qt0 (A) and qt0 + 1 (B)
    const bool actA = qt0 < nT, actB = qt0 + 1 < nT;       // wave-uniform
    half8_t qf[2][4];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int qc = min((qt0 + t) * 32 + r, L - 1);     // (a tile B past the last one computes on clamped rows and is not stored)
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) qf[t][ks] = *(const half8_t*)(base + (long)qc * ld + ks * 16 + h * 8);
    }

    // staging as in variant 5: 16 LDS-DMA instructions of 8 rows x 128 B per block, 16 / NW per wave
    constexpr int NI = 16 / NW;
    static_assert(NI * NW == 16, "NW must divide 16");
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
        sdst[t] = lds_base + 2u * (unsigned)(isv * (KB6 * 64) + r0 * 64);
    }
    const unsigned row_bytes = 2u * (unsigned)ld;
    auto stage = [&](int st, int kb_) {
        const int k0 = kb_ * KB6;
        const half_t* kbase = base + (long)k0 * ld;          // wave-uniform
        if (k0 + KB6 <= L) {
#pragma unroll
            for (int t = 0; t < NI; ++t) glds16_asm6(kbase, soff[t], __builtin_amdgcn_readfirstlane(sdst[t] + 2u * (unsigned)(st * STAGE6)));
        } else {                                           // the last block: rows past the last key repeat it (finite values, masked below)
#pragma unroll
            for (int t = 0; t < NI; ++t)
                glds16_asm6(kbase, soff[t] - (unsigned)max(k0 + srow_t[t] - (L - 1), 0) * row_bytes,
                            __builtin_amdgcn_readfirstlane(sdst[t] + 2u * (unsigned)(st * STAGE6)));
        }
    };

    float m_run[2] = {-INFINITY, -INFINITY};               // reference maxima of the exponentials (scaled domain), lazily moved
    float lsum[2] = {0.f, 0.f};                            // this lane's share of the row sums (lanes l and l + 32 are added at the end)
    float16_t o[2][2];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int k = 0; k < 16; ++k) { o[t][0][k] = 0.f; o[t][1][k] = 0.f; }

    // per-lane LDS offsets (halves): variant 5's
    const int kswz = (r >> 1) & 7;
    int koff[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) koff[ks] = r * 64 + (((2 * ks) ^ h ^ kswz) << 3);
    const int g = lane >> 4, qq = (lane & 15) >> 2, pp = lane & 3;
    const int vkey = 4 * (g >> 1) + qq;
    const int vswz = ((qq >> 1) & 1) << 2;
    int voff[2];
#pragma unroll
    for (int blk = 0; blk < 2; ++blk)
        voff[blk] = KB6 * 64 + vkey * 64 + (((blk * 4 + (g & 1) * 2 + (pp >> 1)) ^ vswz) << 3) + (pp & 1) * 4;

    const int nb = (L + KB6 - 1) / KB6;
    __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0): the query rows are back before the first DMA (see variant 5)
    __builtin_amdgcn_sched_barrier(0);
    stage(0, 0);

    float16_t s[2][2];
    half8_t pf[2][4];
    float16_t zero16;
#pragma unroll
    for (int k = 0; k < 16; ++k) zero16[k] = 0.f;          // (folds into the MFMA's inline constant 0)

    // MASKED: a last block with fewer than 64 keys that is not peeled (its own instantiation: the full blocks carry no mask code)
    auto block = [&](auto stage_c, auto masked_c, int kb) {
        constexpr bool MASKED = decltype(masked_c)::value != 0;
        const int ST = stage_c;                            // an integral_constant for the full blocks: every LDS address an immediate
        const int st_off = ST * STAGE6;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // block kb has landed for THIS wave's DMA instructions ...
        __builtin_amdgcn_s_barrier();                      // ... and for every wave; every wave is done with block kb - 1
        __builtin_amdgcn_sched_barrier(0);
        if (kb + 1 < nb) stage(ST ^ 1, kb + 1);
        if (!actA) return;
        [[maybe_unused]] const int nvalid = L - kb * KB6;

        // one score MFMA: i = 2 ks + sb (the two 32-key chains alternate)
        auto score = [&](auto T, auto I) {
            constexpr int t = decltype(T)::value, i = decltype(I)::value, ks = i >> 1, sb = i & 1;
            const half8_t kf = *(const half8_t*)(smem + koff[ks] + (st_off + sb * 32 * 64));
            s[t][sb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[t][ks], ks == 0 ? zero16 : s[t][sb], 0, 0, 0);
        };
        // register k of a score tile holds key (k & 3) + 8 (k >> 2) + 4 h of its 32-key half
        auto rowmax = [&](auto T) -> float {
            constexpr int t = decltype(T)::value;
            if constexpr (MASKED) {                        // keys >= L get -inf, i.e. p = 0
#pragma unroll
                for (int sb = 0; sb < 2; ++sb) {
                    const int thr = nvalid - sb * 32 - 4 * h;
#pragma unroll
                    for (int k = 0; k < 16; ++k) s[t][sb][k] = ((k & 3) + 8 * (k >> 2) < thr) ? s[t][sb][k] : -INFINITY;
                }
            }
            float mx = s[t][0][0];
#pragma unroll
            for (int sb = 0; sb < 2; ++sb)
#pragma unroll
                for (int k = 0; k < 16; ++k) mx = fmaxf(mx, s[t][sb][k]);
            const unsigned u = __builtin_bit_cast(unsigned, mx);
            auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
            mx = fmaxf(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1]));
            return mx * scale_log2e;
        };
        auto reference = [&](auto T, float mxs) {
            constexpr int t = decltype(T)::value;
            if (__builtin_amdgcn_ballot_w64(mxs > m_run[t] + 8.0f) != 0) {      // wave-uniform: some row needs a new reference
                const float m_new = fmaxf(m_run[t], mxs);
                const float alpha = __builtin_amdgcn_exp2f(m_run[t] - m_new);   // <= 0 (first block: -inf -> 0)
                m_run[t] = m_new;
#pragma unroll
                for (int k = 0; k < 16; ++k) { o[t][0][k] *= alpha; o[t][1][k] *= alpha; }
                lsum[t] *= alpha;
            }
        };
        // pair p of the 32 scores of a lane: half sb = p >> 3, registers 2 (p & 7), + 1 -> elements 2 (p & 3), + 1 of the P^T fragment of step p >> 2
        auto epair = [&](auto T, auto P) {
            constexpr int t = decltype(T)::value, p = decltype(P)::value, sb = p >> 3, k = 2 * (p & 7);
            const float2_t e2 = {__builtin_amdgcn_exp2f(__builtin_fmaf(s[t][sb][k], scale_log2e, -m_run[t])),
                                 __builtin_amdgcn_exp2f(__builtin_fmaf(s[t][sb][k + 1], scale_log2e, -m_run[t]))};
            lsum[t] += e2[0] + e2[1];
            const half2_t p2 = __builtin_convertvector(e2, half2_t);
            pf[t][p >> 2][2 * (p & 3)] = p2[0];
            pf[t][p >> 2][2 * (p & 3) + 1] = p2[1];
        };
        auto pv = [&](auto T, auto S_, auto BLK) {
            constexpr int t = decltype(T)::value, st = decltype(S_)::value, blk = decltype(BLK)::value;
            const half4_t v0 = tr_read6(smem + voff[blk] + (st_off + st * 16 * 64));
            const half4_t v1 = tr_read6(smem + voff[blk] + (st_off + (st * 16 + 8) * 64));
            const half8_t vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            o[t][blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[t][st], o[t][blk], 0, 0, 0);
        };
        constexpr ic<0> A{};
        constexpr ic<1> B{};

        if constexpr (SH != 0) {
            auto score2 = [&](auto I) {
                constexpr int i = decltype(I)::value, ks = i >> 1, sb = i & 1;
                const half8_t kf = *(const half8_t*)(smem + koff[ks] + (st_off + sb * 32 * 64));
                s[0][sb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[0][ks], ks == 0 ? zero16 : s[0][sb], 0, 0, 0);
                s[1][sb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[1][ks], ks == 0 ? zero16 : s[1][sb], 0, 0, 0);
            };
            auto pv2 = [&](auto S_, auto BLK) {
                constexpr int st = decltype(S_)::value, blk = decltype(BLK)::value;
                const half4_t v0 = tr_read6(smem + voff[blk] + (st_off + st * 16 * 64));
                const half4_t v1 = tr_read6(smem + voff[blk] + (st_off + (st * 16 + 8) * 64));
                const half8_t vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                o[0][blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[0][st], o[0][blk], 0, 0, 0);
                o[1][blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf[1][st], o[1][blk], 0, 0, 0);
            };
            score2(ic<0>{}); score2(ic<1>{}); score2(ic<2>{}); score2(ic<3>{});
            score2(ic<4>{}); score2(ic<5>{}); score2(ic<6>{}); score2(ic<7>{});
            PIN6();
            const float mxa = rowmax(A);
            reference(A, mxa);
            epair(A, ic<0>{}); epair(A, ic<1>{}); epair(A, ic<2>{}); epair(A, ic<3>{});
            epair(A, ic<4>{}); epair(A, ic<5>{}); epair(A, ic<6>{}); epair(A, ic<7>{});
            epair(A, ic<8>{}); epair(A, ic<9>{}); epair(A, ic<10>{}); epair(A, ic<11>{});
            epair(A, ic<12>{}); epair(A, ic<13>{}); epair(A, ic<14>{}); epair(A, ic<15>{});
            const float mxb = rowmax(B);
            reference(B, mxb);
            epair(B, ic<0>{}); epair(B, ic<1>{}); epair(B, ic<2>{}); epair(B, ic<3>{});
            PIN6(); pv2(ic<0>{}, ic<0>{}); PIN6();
            epair(B, ic<4>{}); epair(B, ic<5>{});
            PIN6(); pv2(ic<0>{}, ic<1>{}); PIN6();
            epair(B, ic<6>{}); epair(B, ic<7>{});
            PIN6(); pv2(ic<1>{}, ic<0>{}); PIN6();
            epair(B, ic<8>{}); epair(B, ic<9>{});
            PIN6(); pv2(ic<1>{}, ic<1>{}); PIN6();
            epair(B, ic<10>{}); epair(B, ic<11>{});
            PIN6(); pv2(ic<2>{}, ic<0>{}); PIN6();
            epair(B, ic<12>{}); epair(B, ic<13>{});
            PIN6(); pv2(ic<2>{}, ic<1>{}); PIN6();
            epair(B, ic<14>{}); epair(B, ic<15>{});
            PIN6(); pv2(ic<3>{}, ic<0>{}); pv2(ic<3>{}, ic<1>{});
            return;
        }
        score(A, ic<0>{}); score(A, ic<1>{}); score(A, ic<2>{}); score(A, ic<3>{});
        score(A, ic<4>{}); score(A, ic<5>{}); score(A, ic<6>{}); score(A, ic<7>{});
        score(B, ic<0>{}); score(B, ic<1>{}); score(B, ic<2>{}); score(B, ic<3>{});
        PIN6();
        const float mxa = rowmax(A);
        PIN6(); score(B, ic<4>{}); PIN6();
        reference(A, mxa);
        PIN6(); score(B, ic<5>{}); PIN6();
        epair(A, ic<0>{}); epair(A, ic<1>{}); epair(A, ic<2>{});
        PIN6(); score(B, ic<6>{}); PIN6();
        epair(A, ic<3>{}); epair(A, ic<4>{}); epair(A, ic<5>{});
        PIN6(); score(B, ic<7>{}); PIN6();
        epair(A, ic<6>{}); epair(A, ic<7>{});
        PIN6(); pv(A, ic<0>{}, ic<0>{}); PIN6();
        epair(A, ic<8>{}); epair(A, ic<9>{}); epair(A, ic<10>{});
        PIN6(); pv(A, ic<0>{}, ic<1>{}); PIN6();
        epair(A, ic<11>{}); epair(A, ic<12>{}); epair(A, ic<13>{});
        PIN6(); pv(A, ic<1>{}, ic<0>{}); PIN6();
        epair(A, ic<14>{}); epair(A, ic<15>{});
        PIN6(); pv(A, ic<1>{}, ic<1>{}); PIN6();
        const float mxb = rowmax(B);
        PIN6(); pv(A, ic<2>{}, ic<0>{}); PIN6();
        reference(B, mxb);
        PIN6(); pv(A, ic<2>{}, ic<1>{}); PIN6();
        epair(B, ic<0>{}); epair(B, ic<1>{}); epair(B, ic<2>{});
        PIN6(); pv(A, ic<3>{}, ic<0>{}); PIN6();
        epair(B, ic<3>{}); epair(B, ic<4>{}); epair(B, ic<5>{});
        PIN6(); pv(A, ic<3>{}, ic<1>{}); PIN6();
        epair(B, ic<6>{}); epair(B, ic<7>{});
        PIN6(); pv(B, ic<0>{}, ic<0>{}); PIN6();
        epair(B, ic<8>{}); epair(B, ic<9>{}); epair(B, ic<10>{});
        PIN6(); pv(B, ic<0>{}, ic<1>{}); PIN6();
        epair(B, ic<11>{}); epair(B, ic<12>{}); epair(B, ic<13>{});
        PIN6(); pv(B, ic<1>{}, ic<0>{}); PIN6();
        epair(B, ic<14>{}); epair(B, ic<15>{});
        PIN6(); pv(B, ic<1>{}, ic<1>{});
        pv(B, ic<2>{}, ic<0>{}); pv(B, ic<2>{}, ic<1>{}); pv(B, ic<3>{}, ic<0>{}); pv(B, ic<3>{}, ic<1>{});
    };

    // A last block of at most 16 keys (every CLIP ViT: L = G*G + 1) is peeled into a small body per tile, as in variant 5
    const int tail = L - (nb - 1) * KB6;
    const bool peel = nb > 1 && tail <= 16;
    const int nb_main = peel ? nb - 1 : nb;
    const int nb_full = min(nb_main, L / KB6);             // full 64-key blocks; at most one masked block follows
    for (int kb = 0; kb < nb_full; kb += 2) {
        block(ic<0>{}, ic<0>{}, kb);
        if (kb + 1 < nb_full) block(ic<1>{}, ic<0>{}, kb + 1);
    }
    if (nb_full < nb_main) block(nb_full & 1, ic<1>{}, nb_full);
    if (peel) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        __builtin_amdgcn_sched_barrier(0);
        const int st_off = ((nb - 1) & 1) * STAGE6;
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (!(t == 0 ? actA : actB)) continue;
            float16_t s0;
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const half8_t kf = *(const half8_t*)(smem + koff[ks] + st_off);
                s0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[t][ks], ks == 0 ? zero16 : s0, 0, 0, 0);
            }
            const int thr = tail - 4 * h;                  // registers 0..7: keys (k & 3) + 8 (k >> 2) + 4 h
            float mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                s0[k] = ((k & 3) + 8 * (k >> 2) < thr) ? s0[k] : -INFINITY;
                mx = fmaxf(mx, s0[k]);
            }
            {
                const unsigned u = __builtin_bit_cast(unsigned, mx);
                auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                mx = fmaxf(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1]));
            }
            const float mxs = mx * scale_log2e;
            if (__builtin_amdgcn_ballot_w64(mxs > m_run[t] + 8.0f) != 0) {
                const float m_new = fmaxf(m_run[t], mxs);
                const float alpha = __builtin_amdgcn_exp2f(m_run[t] - m_new);
                m_run[t] = m_new;
#pragma unroll
                for (int k = 0; k < 16; ++k) { o[t][0][k] *= alpha; o[t][1][k] *= alpha; }
                lsum[t] *= alpha;
            }
            const float m_ref = m_run[t];
            half8_t pt;
#pragma unroll
            for (int j = 0; j < 8; j += 2) {
                const float2_t e2 = {__builtin_amdgcn_exp2f(__builtin_fmaf(s0[j], scale_log2e, -m_ref)),
                                     __builtin_amdgcn_exp2f(__builtin_fmaf(s0[j + 1], scale_log2e, -m_ref))};
                lsum[t] += e2[0] + e2[1];
                const half2_t p2 = __builtin_convertvector(e2, half2_t);
                pt[j] = p2[0];
                pt[j + 1] = p2[1];
            }
#pragma unroll
            for (int blk = 0; blk < 2; ++blk) {
                const half4_t v0 = tr_read6(smem + voff[blk] + st_off);
                const half4_t v1 = tr_read6(smem + voff[blk] + (st_off + 8 * 64));
                const half8_t vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                o[t][blk] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pt, o[t][blk], 0, 0, 0);
            }
        }
    }

    // ---------------------------------------------------------------- output: O^T / row sum -> fp16 -> LDS -> 128-byte rows
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();                          // every wave is done with the K / V stages: they become staging tiles
    if (!actA) return;
    const int fsw = ((r & 7) << 1) | ((r >> 3) & 1);       // swizzled staging tile: gemm_f16_v5.hip's epilogue / variant 5
    const int er = lane >> 3, ec = lane & 7;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        if (t == 1 && !actB) break;
        char* et = (char*)smem + (wave * 2 + t) * 4096;    // 32 rows x 128 B
        const unsigned u = __builtin_bit_cast(unsigned, lsum[t]);
        auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
        const float inv = 1.0f / (__builtin_bit_cast(float, (unsigned)sw[0]) + __builtin_bit_cast(float, (unsigned)sw[1]));
#pragma unroll
        for (int blk = 0; blk < 2; ++blk)
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) {
                const half4_t w = {(half_t)(o[t][blk][4 * kq] * inv), (half_t)(o[t][blk][4 * kq + 1] * inv),
                                   (half_t)(o[t][blk][4 * kq + 2] * inv), (half_t)(o[t][blk][4 * kq + 3] * inv)};
                *(half4_t*)(et + r * 128 + (((blk * 8 + 2 * kq + h) ^ fsw) << 3)) = w;
            }
#pragma unroll
        for (int it = 0; it < 4; ++it) {
            const int row = it * 8 + er;
            half8_t v = *(const half8_t*)(et + row * 128 + ((ec ^ (row & 7)) << 4));
            if (it & 1) v = (half8_t){v[4], v[5], v[6], v[7], v[0], v[1], v[2], v[3]};
            const int qrow = (qt0 + t) * 32 + row;
            if (qrow < Lq) *(half8_t*)(out + ((long)b * Lq + qrow) * D + hh * 64 + ec * 8) = v;
        }
    }
}

template <int NW, int SH>
int launch_v6(const half_t* qkv, half_t* out, int B, int L, int Lq, int H, hipStream_t s) {
    const int nT = (Lq + 31) / 32, nJ = (nT + 1) / 2, nWG = (nJ + NW - 1) / NW, nBH = B * H;
    const float sl2e = 0.125f * 1.4426950408889634f;
    const dim3 grid((unsigned)((long)((nBH + 7) / 8) * 8 * nWG));
    hipLaunchKernelGGL((attn_f16_v6<NW, SH>), grid, dim3(NW * 64), 0, s, qkv, out, L, Lq, H, nT, nWG, nBH, sl2e);
    return (int)hipGetLastError();
}

}  // namespace

// -100: shape not taken (causal, short sequences, a handful of query rows): the caller falls back.
// mode & 3 = 0: the workgroup size with fewer idle wave slots (ties: four waves); 1: four waves; 2: two waves.  mode & 4: the
// interleaved two-stream schedule (fragments read per tile) instead of the shared-fragment one.
int launch_attention_f16_v6(const half_t* qkv, half_t* out, int B, int L, int Lq, int H, int causal, int mode, hipStream_t s) {
    if (causal || L < 256 || Lq < 32) return -100;
    const int nJ = ((Lq + 31) / 32 + 1) / 2, m = mode & 3;
    const bool two = m == 2 || (m == 0 && ((nJ + 1) / 2) * 2 < ((nJ + 3) / 4) * 4);
    if (mode & 4) return two ? launch_v6<2, 0>(qkv, out, B, L, Lq, H, s) : launch_v6<4, 0>(qkv, out, B, L, Lq, H, s);
    return two ? launch_v6<2, 1>(qkv, out, B, L, Lq, H, s) : launch_v6<4, 1>(qkv, out, B, L, Lq, H, s);
}
#endif  // OVMR_EXPERIMENTS
