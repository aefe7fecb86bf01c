// Attention for SHORT sequences (fp16, hd = 64, L <= 32) of SEVERAL groups in one launch (round 4): the text tower of a classifier
// head runs three prompt families through one pass -- multimodal prompts (L = eos + n_ctx + 1, ~12), vision prompts (L = 2 + n_ctx = 4),
// zero-shot prompts (L = eos + 1, ~10) -- and attention is the one launch of a block that needs the sequence structure
// (clip/model.py:184-188, causal mask :802-809).  The general kernel (attention.hip, attn_f16_v0) gives a (sequence, head) a 256-thread
// workgroup and stages 64-key blocks; at these lengths three of its four waves idle and most of a block is padding, and one launch per
// group costs what a launch costs here (~8 us, r04: tools/graph_head_probe.py).  This kernel:
//   * ONE WAVE per (sequence, head), four per workgroup, all groups of the pass in one grid (group table by value);
//   * same arithmetic as attn_f16_v0 with one key block: S^T = K Q^T as 16x16x32 MFMAs (K and Q fragments straight from global memory:
//     16 bytes per lane, the operands of these passes are cache resident), masked, exact row maximum, p = exp2(s - m) in fp32, row sum
//     in fp32, P rounded to fp16 as the B operand of O^T = V^T P^T, V transposed through a per-wave LDS tile ([d][key] rows of 72 B);
//   * L <= 16: one query tile; 16 < L <= 32: two, sharing the K fragments and the V^T tile.
#include "common.h"

struct AttnShortGroups {       // up to 4 groups of sequences in one [rows, 3 H 64] qkv buffer
    int n;                     // groups
    int L[4];                  // tokens per sequence
    int nseq[4];               // sequences
    int row0[4];               // first token row of the group
    int pair0[5];              // prefix sums of nseq * H
};

namespace {

constexpr int VS_LD = 36;      // halves per V^T row: 32 keys + 4 pad (72 B: conflict-free 8-byte reads of 16 rows)

template <bool CAUSAL>
__global__ __launch_bounds__(256) void attn_f16_short(const half_t* __restrict__ qkv, half_t* __restrict__ out, AttnShortGroups g,
                                                      int H, float scale_log2e) {
    __shared__ __attribute__((aligned(16))) half_t sVt_all[4][64 * VS_LD];

    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int pair = blockIdx.x * 4 + wave;
    if (pair >= g.pair0[g.n]) return;                      // (no workgroup barrier below: a wave may leave)
    int gi = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k) gi += (k < g.n && pair >= g.pair0[k]) ? 1 : 0;
    const int L = g.L[gi], local = pair - g.pair0[gi];
    const int h = local % H, b = local / H;
    const int D = H * 64, ld = 3 * D;
    const long row_base = (long)g.row0[gi] + (long)b * L;
    const half_t* base = qkv + row_base * ld + h * 64;
    half_t* sVt = sVt_all[wave];

    // K fragments of the two 16-key blocks (rows past the sequence repeat its last key: finite, masked below)
    half8_t kf[2][2];
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
        const int kc = min(nt * 16 + fr, L - 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) kf[nt][ks] = *(const half8_t*)(base + D + (long)kc * ld + ks * 32 + fg * 8);
    }
    // V^T tile: lane = (4 d) x (4 keys), two key halves
    {
        const int dg = lane & 15, kg = lane >> 4;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            half4_t r[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int kc = min(half * 16 + kg * 4 + i, L - 1);
                r[i] = *(const half4_t*)(base + 2 * D + (long)kc * ld + dg * 4);
            }
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const half4_t w = {r[0][d], r[1][d], r[2][d], r[3][d]};
                *(half4_t*)(sVt + (dg * 4 + d) * VS_LD + half * 16 + kg * 4) = w;
            }
        }
    }
    const int nqt = (L + 15) >> 4;                          // 1 or 2 query tiles
    const int nkb = (L + 15) >> 4;                          // key blocks that hold a valid key
    for (int qt = 0; qt < nqt; ++qt) {
        const int q = qt * 16 + fr, qc = min(q, L - 1);
        half8_t qf[2];
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) qf[ks] = *(const half8_t*)(base + (long)qc * ld + ks * 32 + fg * 8);
        float4_t s[2];
#pragma unroll
        for (int nt = 0; nt < 2; ++nt) {
            s[nt] = (float4_t){0.f, 0.f, 0.f, 0.f};
            if (nt < nkb) {
#pragma unroll
                for (int ks = 0; ks < 2; ++ks) s[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[nt][ks], qf[ks], s[nt], 0, 0, 0);
            }
        }
        float mx = -INFINITY;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int key = nt * 16 + fg * 4 + r;
                const bool ok = (key < L) && (!CAUSAL || key <= q);
                const float v = ok ? s[nt][r] * scale_log2e : -INFINITY;
                s[nt][r] = v;
                mx = fmaxf(mx, v);
            }
        mx = fmaxf(mx, __shfl_xor(mx, 16, 64));
        mx = fmaxf(mx, __shfl_xor(mx, 32, 64));
        float psum = 0.f;
#pragma unroll
        for (int nt = 0; nt < 2; ++nt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float p = exp2f(s[nt][r] - mx);
                s[nt][r] = p;
                psum += p;
            }
        psum += __shfl_xor(psum, 16, 64);
        psum += __shfl_xor(psum, 32, 64);
        half8_t pf;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            pf[j] = (half_t)s[0][j];
            pf[4 + j] = (half_t)s[1][j];
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // this wave's V^T tile is written (LDS executes a wave's accesses in order)
        float4_t o[4];
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const half_t* vr = sVt + (dt * 16 + fr) * VS_LD + fg * 4;
            const half4_t v0 = *(const half4_t*)vr;
            const half4_t v1 = *(const half4_t*)(vr + 16);
            const half8_t vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf, (float4_t){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
        }
        if (q < L) {
            const float inv = 1.0f / psum;
            half_t* op = out + (row_base + q) * D + h * 64 + fg * 4;
#pragma unroll
            for (int dt = 0; dt < 4; ++dt) {
                const half4_t w = {(half_t)(o[dt][0] * inv), (half_t)(o[dt][1] * inv), (half_t)(o[dt][2] * inv), (half_t)(o[dt][3] * inv)};
                *(half4_t*)(op + dt * 16) = w;
            }
        }
    }
}

}  // namespace

// Every group: sequences of L[i] <= 32 tokens, rows row0[i] .. of qkv [rows, 3 H 64]; out [rows, H 64].  -100: a group is too long.
int launch_attention_f16_short(const half_t* qkv, half_t* out, int n_groups, const int* nseq, const int* L, const long* row0, int H,
                               int causal, hipStream_t s) {
    if (n_groups < 1 || n_groups > 4) return -100;
    AttnShortGroups g;
    g.n = n_groups;
    g.pair0[0] = 0;
    for (int i = 0; i < 4; ++i) {
        const bool live = i < n_groups;
        if (live && (L[i] < 1 || L[i] > 32 || row0[i] > 0x7fffffffL)) return -100;
        g.L[i] = live ? L[i] : 1;
        g.nseq[i] = live ? nseq[i] : 0;
        g.row0[i] = live ? (int)row0[i] : 0;
        g.pair0[i + 1] = g.pair0[i] + g.nseq[i] * H;
    }
    if (g.pair0[n_groups] == 0) return 0;
    const float sl2e = 0.125f * 1.4426950408889634f;       // hd^-0.5 * log2(e), hd = 64
    const dim3 grid((unsigned)((g.pair0[n_groups] + 3) / 4));
    if (causal) hipLaunchKernelGGL(attn_f16_short<true>, grid, dim3(256), 0, s, qkv, out, g, H, sl2e);
    else hipLaunchKernelGGL(attn_f16_short<false>, grid, dim3(256), 0, s, qkv, out, g, H, sl2e);
    return (int)hipGetLastError();
}
