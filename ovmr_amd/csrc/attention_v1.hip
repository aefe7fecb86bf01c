// Attention variant 1 (fp16, hd = 64): same swapped-product flash structure as attention.hip, with
//   * K and V blocks staged by LDS-DMA (global_load_lds_dwordx4) into a DOUBLE buffer: block k+1 is in
//     flight while block k is consumed, no VGPR staging and no transposing LDS writes;
//   * V kept row-major [key][d] like K (128-byte rows, chunk c of row r in slot c ^ (r & 7)) and consumed
//     through ds_read_b64_tr_b16: per 16-lane group the instruction reads 4 keys x 16 d and hands lane i
//     the 4 keys of column d0+i -- exactly the V^T fragment (A operand of O^T = V^T P^T);
//   * (r01g) the VALU issue port bounds this kernel (75 % busy against ~20 % for the MFMA pipe, tools/pmc_attn.sh): lazy
//     rescaling against a reference maximum and row sums from the matrix pipe cut vector instructions (194 -> 183 us); tiles packed two per wave, no accumulator zeroing in complete key blocks and
//     threshold-form masking (183 -> 167 us);
//     a variant with a specialised body for complete unmasked key blocks spilled registers at the 128-VGPR cap and ran
//     at 255 us -- not kept;
//   * (r02) a tail key block that holds a single 16-key sub-tile (every CLIP ViT has one: L = G*G + 1) is peeled off the block
//     loop into its own small body (4 scores per lane instead of 16), and the row maximum crosses lanes with
//     v_permlane16_swap / v_permlane32_swap instead of two LDS round trips: 170 -> 162 us at 512 x 12 x 197 (same box);
//     the ViT-B/16 image shape itself now runs attention_v3.hip (142 us);
//   * two 16-row query tiles per wave, 4 waves (8 tiles) per workgroup, both tiles of a wave sharing every K / V fragment
//     read from LDS; <= 128 VGPRs, i.e. 4 waves per SIMD (3 waves/SIMD: 233 us instead of 193 at B = 512); the workgroups
//     of a head run on one XCD; 16 tiles / 7 waves per workgroup (K/V read once) measured 198 us, 4 tiles 321 us.
#include "common.h"

#include <algorithm>
#include <cstdlib>

namespace {

constexpr int KB1 = 64;
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef short short4v __attribute__((__vector_size__(8)));

__device__ __forceinline__ half4_t tr_read(const half_t* p) {
    short4v r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)p);
    return __builtin_bit_cast(half4_t, r);
}

// max over the four lanes {l, l^16, l^32, l^48} that hold one query row's scores, without the LDS crossbar: v_permlane16_swap /
// v_permlane32_swap exchange 16- / 32-lane halves between two registers, so with both operands = x the two results hold
// x of this lane and x of the partner lane (ds_bpermute: two ~100-cycle LDS round trips in the max -> exp dependency chain).
__device__ __forceinline__ float row_max4(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    x = fmaxf(__builtin_bit_cast(float, (unsigned)a[0]), __builtin_bit_cast(float, (unsigned)a[1]));
    const unsigned w = __builtin_bit_cast(unsigned, x);
    auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return fmaxf(__builtin_bit_cast(float, (unsigned)b[0]), __builtin_bit_cast(float, (unsigned)b[1]));
}

template <bool CAUSAL>
__global__ __launch_bounds__(512, 4) void attn_f16_v1(const half_t* __restrict__ qkv, half_t* __restrict__ out,
                                                   int L, int Lq, int H, int nT, int nWG, int nBH, float scale_log2e) {
    __shared__ __attribute__((aligned(16))) half_t smem[2 * 2 * KB1 * 64];   // [buf][K|V][64 keys][64 d]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int D = H * 64, ld = 3 * D;
    // the workgroups of one (sequence, head) get block ids congruent mod 8: same XCD, so the second one finds K / V in that L2
    const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
    const int wg = slot % nWG, bh = (slot / nWG) * 8 + xcd;
    if (bh >= nBH) return;
    const int h = bh % H, b = bh / H;
    // query tiles of this workgroup: full groups of 2 * nW first, the remainder in the last one, and wave w takes tiles 2w
    // and 2w+1 -- at L = 197 (13 tiles, two 4-wave workgroups) that leaves ONE half-empty wave and one wave without tiles
    // (it only helps staging K / V) instead of three half-empty ones whose second tile still cost a full softmax
    const int tpw = 2 * ((int)blockDim.x >> 6);
    const int t0 = wg * tpw, t1 = min(nT, t0 + tpw);
    const int nW = (int)blockDim.x >> 6;                               // waves: ceil(tiles per workgroup / 2)
    const half_t* base = qkv + (long)b * L * ld + h * 64;

    int qrow[2];
    bool act[2];
    half8_t qf[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int qt = t0 + 2 * wave + u;
        act[u] = qt < t1;
        qrow[u] = qt * 16 + fr;
        const int qc = min(qrow[u], L - 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf[u][ks] = *(const half8_t*)(base + (long)qc * ld + ks * 32 + fg * 8);
    }
    // m_run: the reference maximum the exponentials are taken against (scaled domain).  It only moves when a block's
    // maximum exceeds it by more than 8 (a factor 256 in p, harmless for fp16 P and fp32 accumulators): then o and the row
    // sums are rescaled.  After the first key block that is rare, so the 16 packed multiplies and the exp of the usual
    // every-block rescale disappear.  The row sums are accumulated by the matrix pipe (ones . P^T, accumulator ol), not by
    // 32 VALU adds and two cross-lane shuffles per tile: the VALU issue port is what bounds this kernel (75 % busy,
    // profiles/r01g_pmc_attn.json), the MFMA pipe is not.
    float m_run[2] = {-INFINITY, -INFINITY};
    float4_t ol[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
    half8_t ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (half_t)1.f;
    float4_t o[2][4];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int i = 0; i < 4; ++i) o[u][i] = (float4_t){0.f, 0.f, 0.f, 0.f};

    // staging: 16 LDS-DMA instructions per block (8 K + 8 V) dealt round-robin to the waves; one instruction = 8 rows x 128 B
    const int srow = lane >> 3, schunk = ((lane & 7) ^ srow) * 8;
    auto stage = [&](int buf, int k0) {
        half_t* dst = smem + buf * (2 * KB1 * 64);
        for (int ins = wave; ins < 16; ins += nW) {       // 0..7: K rows, 8..15: V rows
            const int isv = ins >> 3, r0 = (ins & 7) * 8;
            const int kc = min(k0 + r0 + srow, L - 1);
            __builtin_amdgcn_global_load_lds((gptr_t)(base + (1 + isv) * D + (long)kc * ld + schunk),
                                             (lptr_t)(dst + isv * (KB1 * 64) + r0 * 64), 16, 0, 0);
        }
    };

    const int last_tile = t1 - 1;
    const int kmax = CAUSAL ? min(L, (last_tile + 1) * 16) : L;
    const int nb = (kmax + KB1 - 1) / KB1;
    // (r02) a last key block that holds a single 16-key sub-tile is peeled off the loop (see the tail body below)
    const bool peel = nb > 1 && ((kmax - (nb - 1) * KB1 + 15) >> 4) == 1;
    const int nb_main = peel ? nb - 1 : nb;
    stage(0, 0);
    for (int kb = 0; kb < nb_main; ++kb) {
        // every wave must have its own LDS-DMA of block kb retired BEFORE the barrier (hipcc does not always put the
        // vmcnt(0) of __syncthreads() ahead of the barrier: it was found sunk to the first V read, a cross-wave race)
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                   // block kb landed for all waves, other buffer free
        if (kb + 1 < nb) stage((kb + 1) & 1, (kb + 1) * KB1);
        const half_t* sK = smem + (kb & 1) * (2 * KB1 * 64);
        const half_t* sV = sK + KB1 * 64;
        const int k0 = kb * KB1;
        // Both query tiles of the wave go through a key block TOGETHER: every K fragment and every transposed V fragment
        // is read from LDS once and feeds two MFMAs (one per tile).  With one tile at a time the LDS pipe was ~90 % busy
        // (16 KiB of fragment reads per (tile, block) against 16 MFMAs) and set the pace, not the matrix cores.
        bool on[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) on[u] = act[u] && !(CAUSAL && k0 > (t0 + 2 * wave + u) * 16 + 15);   // wave-uniform
        if (!on[0] && !on[1]) continue;
        // 16-key sub-tiles / 32-key halves of this block that hold any valid key (L = 197: the last block has 5 keys)
        const int nvalid = min(kmax - k0, KB1);
        const int ntv = (nvalid + 15) >> 4, nsv = (nvalid + 31) >> 5;
        float4_t s[2][4];
        const float4_t zero = {0.f, 0.f, 0.f, 0.f};
        auto s_tile = [&](int nt) {
            const half8_t kf0 = *(const half8_t*)(sK + (nt * 16 + fr) * 64 + ((fg ^ (fr & 7)) << 3));
            const half8_t kf1 = *(const half8_t*)(sK + (nt * 16 + fr) * 64 + (((4 + fg) ^ (fr & 7)) << 3));
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                s[u][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf0, qf[u][0], zero, 0, 0, 0);
                s[u][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf1, qf[u][1], s[u][nt], 0, 0, 0);
            }
        };
        if (ntv == 4) {          // complete key block (3 of 4 at L = 197): no accumulator zeroing in front of the MFMAs
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) s_tile(nt);
        } else {
#pragma unroll
            for (int nt = 0; nt < 4; ++nt) {
                if (nt < ntv) s_tile(nt);
                else {
#pragma unroll
                    for (int u = 0; u < 2; ++u) s[u][nt] = zero;
                }
            }
        }
        half8_t pf[2][2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int q = qrow[u];
            // raw-score maximum first (masking only in blocks that need it: the last key block, or the diagonal blocks of
            // the causal case), then p = exp2(fma(s, scale*log2e, -max*scale*log2e)): one VALU op less per element.
            // A tile that is switched off for this block sees every key masked: alpha = 1, p = 0, nothing changes.
            const bool need_mask = !on[u] || (k0 + KB1 > L) || (CAUSAL && k0 + KB1 - 1 > (t0 + 2 * wave + u) * 16);   // wave-uniform
            float mx = -INFINITY;
            if (need_mask) {
                // keys of this lane are k0 + fg*4 + (nt*16 + r): valid while (nt*16 + r) < thr -- one compare against a
                // constant and one select per element, no per-element key index
                const int thr = (on[u] ? (CAUSAL ? min(L, q + 1) : L) : 0) - k0 - fg * 4;
#pragma unroll
                for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                    for (int r = 0; r < 4; ++r) s[u][nt][r] = (nt * 16 + r < thr) ? s[u][nt][r] : -INFINITY;
            }
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[u][nt][r]);
            mx = row_max4(mx);
            const float mxs = mx * scale_log2e;
            if (__builtin_amdgcn_ballot_w64(mxs > m_run[u] + 8.0f) != 0) {     // wave-uniform: some row needs a new reference
                const float m_new = fmaxf(m_run[u], mxs);
                const float alpha = __builtin_amdgcn_exp2f(m_run[u] - m_new);   // raw v_exp_f32: arguments are <= 0 (first block: -inf)
                m_run[u] = m_new;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) o[u][dt] *= alpha;
                ol[u] *= alpha;
            }
            const float m_ref = m_run[u];
#pragma unroll
            for (int nt = 0; nt < 4; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    s[u][nt][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[u][nt][r], scale_log2e, -m_ref));
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    pf[u][s2][j] = (half_t)s[u][2 * s2][j];
                    pf[u][s2][4 + j] = (half_t)s[u][2 * s2 + 1][j];
                }
        }
#pragma unroll
        for (int s2 = 0; s2 < 2; ++s2) {
            if (s2 >= nsv) continue;
            // V^T fragment through the transposing read: lane (fr, fg) addresses key row kr, 4 d-columns
            const int kr = s2 * 32 + fg * 4 + (fr >> 2);
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const int c = dt * 2 + ((fr & 3) >> 1);
                const int off = (((c ^ (kr & 7)) << 3) + (fr & 1) * 4);      // halves; (kr+16)&7 == kr&7
                half4_t v0 = tr_read(sV + kr * 64 + off);
                half4_t v1 = tr_read(sV + (kr + 16) * 64 + off);
                half8_t vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
#pragma unroll
                for (int u = 0; u < 2; ++u) o[u][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf[u][s2], o[u][dt], 0, 0, 0);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) ol[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, pf[u][s2], ol[u], 0, 0, 0);   // row sums
        }
    }
    if (peel) {
        const int kb = nb - 1, k0 = kb * KB1;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();                                   // the tail block landed for all waves
        const half_t* sK = smem + (kb & 1) * (2 * KB1 * 64);
        const half_t* sV = sK + KB1 * 64;
        bool on[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) on[u] = act[u] && !(CAUSAL && k0 > (t0 + 2 * wave + u) * 16 + 15);
        // (r02) Tail block with a single 16-key sub-tile -- every CLIP ViT has one: L = G*G + 1 leaves 5 (G = 14) or 1 key
        // behind the last full block.  Its own small body: 4 scores per lane instead of 16, so the maximum / exponential /
        // conversion passes cost a quarter; before, this block ran the full 64-key softmax for 5 valid keys (a quarter of
        // the kernel's vector instructions at L = 197).  Keys 16..63 of the block get P = 0; their V rows are finite
        // (the staging clamps the row index), so they add exact zeros.
        const float4_t zero4 = {0.f, 0.f, 0.f, 0.f};
        const half8_t kf0 = *(const half8_t*)(sK + fr * 64 + ((fg ^ (fr & 7)) << 3));
        const half8_t kf1 = *(const half8_t*)(sK + fr * 64 + (((4 + fg) ^ (fr & 7)) << 3));
        half8_t pt[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            float4_t s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf0, qf[u][0], zero4, 0, 0, 0);
            s1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf1, qf[u][1], s1, 0, 0, 0);
            const int thr = (on[u] ? (CAUSAL ? min(L, qrow[u] + 1) : L) : 0) - k0 - fg * 4;
#pragma unroll
            for (int r = 0; r < 4; ++r) s1[r] = (r < thr) ? s1[r] : -INFINITY;
            const float mx = row_max4(fmaxf(fmaxf(s1[0], s1[1]), fmaxf(s1[2], s1[3])));
            const float mxs = mx * scale_log2e;
            if (__builtin_amdgcn_ballot_w64(mxs > m_run[u] + 8.0f) != 0) {
                const float m_new = fmaxf(m_run[u], mxs);
                const float alpha = __builtin_amdgcn_exp2f(m_run[u] - m_new);
                m_run[u] = m_new;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) o[u][dt] *= alpha;
                ol[u] *= alpha;
            }
            const float m_ref = m_run[u];
#pragma unroll
            for (int r = 0; r < 4; ++r) s1[r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s1[r], scale_log2e, -m_ref));
            pt[u] = (half8_t){(half_t)s1[0], (half_t)s1[1], (half_t)s1[2], (half_t)s1[3], (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
        }
        const int kr = fg * 4 + (fr >> 2);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const int c = dt * 2 + ((fr & 3) >> 1);
            const int off = (((c ^ (kr & 7)) << 3) + (fr & 1) * 4);
            const half4_t v0 = tr_read(sV + kr * 64 + off);
            const half8_t vf = {v0[0], v0[1], v0[2], v0[3], (half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
#pragma unroll
            for (int u = 0; u < 2; ++u) o[u][dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pt[u], o[u][dt], 0, 0, 0);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) ol[u] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, pt[u], ol[u], 0, 0, 0);
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        if (act[u] && qrow[u] < Lq) {
            const float inv = 1.0f / ol[u][0];            // every d-row of ones . P^T holds the row sum of this lane's query
            half_t* op = out + ((long)b * Lq + qrow[u]) * D + h * 64 + fg * 4;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                half4_t w = {(half_t)(o[u][dt][0] * inv), (half_t)(o[u][dt][1] * inv), (half_t)(o[u][dt][2] * inv),
                             (half_t)(o[u][dt][3] * inv)};
                *(half4_t*)(op + dt * 16) = w;
            }
        }
    }
}

}  // namespace

int launch_attention_f16_v1(const half_t* qkv, half_t* out, int B, int L, int Lq, int H, int causal, hipStream_t s) {
    int tpw = 8;                                           // query tiles per workgroup: 8 (4 waves x 2 tiles) measured best: 16 -> 198 us, 8 -> 193 us, 4 -> 321 us at B = 512
#ifdef OVMR_EXPERIMENTS
    if (const char* e = getenv("OVMR_ATTN_TPW")) { tpw = atoi(e); if (tpw < 2 || tpw > 16) tpw = 8; }
#endif
    const int nT = (Lq + 15) / 16, nWG = (nT + tpw - 1) / tpw;
    const int per = (nT + nWG - 1) / nWG;                  // most tiles any workgroup gets
    const dim3 block(64 * std::max(4, (per + 1) / 2));      // at least 4 waves: the spare ones only help staging K / V
    const float sl2e = 0.125f * 1.4426950408889634f;
    const int nBH = B * H;
    const dim3 grid((unsigned)((long)((nBH + 7) / 8) * 8 * nWG));
    if (causal) hipLaunchKernelGGL(attn_f16_v1<true>, grid, block, 0, s, qkv, out, L, Lq, H, nT, nWG, nBH, sl2e);
    else hipLaunchKernelGGL(attn_f16_v1<false>, grid, block, 0, s, qkv, out, L, Lq, H, nT, nWG, nBH, sl2e);
    return (int)hipGetLastError();
}
