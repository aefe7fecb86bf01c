// One 16-row query tile against ALL keys of a head in LDS (single pass, hd = 64): the tile body of attention variants 3 and 4.
//
// LDS image of a head: K rows then V rows, 128-byte rows, 16-byte chunk c of row r stored in slot c ^ (r & 7).
// Lane (fr, fg) of the wave holds query fr of the tile; S^T = K Q^T puts keys on the accumulator rows, so a query's scores sit
// in four lanes (fg) x NT x 4 registers, P^T goes straight into the B operands of the PV products, V^T comes through
// ds_read_b64_tr_b16 and the row sums come from the matrix pipe (ones . P^T).
//
// Instruction order is pinned (r02m): a wave is the only thing that hides its own LDS latency here (3-4 waves per SIMD, all in the
// same phase), and left alone hipcc read each K fragment pair right in front of the two MFMAs that use it.
//   * S^T: the fragment reads run PD sub-tiles ahead of the MFMAs (sched_group_barrier);
//   * softmax / PV in steps of 32 keys: [8 transposing V reads] [exponentials + packing of the step: ~180 VALU cycles that cover
//     the reads] [5 MFMAs], fenced so that the MFMAs of step s execute in the matrix pipe under the vector work of step s + 1.
#pragma once
#include "common.h"

namespace attn_sp {

typedef short short4v __attribute__((__vector_size__(8)));

__device__ __forceinline__ half4_t tr_read(const half_t* p) {
    short4v r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)p);
    return __builtin_bit_cast(half4_t, r);
}

// max over the four lanes {l, l^16, l^32, l^48} that hold one query's scores
__device__ __forceinline__ float row_max4(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    x = fmaxf(__builtin_bit_cast(float, (unsigned)a[0]), __builtin_bit_cast(float, (unsigned)a[1]));
    const unsigned w = __builtin_bit_cast(unsigned, x);
    auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return fmaxf(__builtin_bit_cast(float, (unsigned)b[0]), __builtin_bit_cast(float, (unsigned)b[1]));
}

// o[dt][r]: un-normalised O^T (d = dt*16 + fg*4 + r) of this lane's query; returns the row sum
template <int NT, int PIPE_ABL>      // bit 0: pinned order; bits 1.. : timing-only ablations (2 no exp, 4 no LDS reads, 8 one PV MFMA per step)
__device__ __forceinline__ float tile(const half_t* sK, const half_t* sV, const half8_t (&qf)[2], int L, float scale_log2e,
                                      int fr, int fg, float4_t (&o)[4]) {
    constexpr int NS = (NT + 1) / 2, PD = 3;
    constexpr int PIPE = PIPE_ABL & 1, ABL = PIPE_ABL >> 1;
    constexpr bool odd_tail = (NT & 1) != 0;
    const float4_t zero = {0.f, 0.f, 0.f, 0.f};
    const int c0 = (fg ^ (fr & 7)) << 3, c1 = ((4 + fg) ^ (fr & 7)) << 3;
    half8_t ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (half_t)1.f;

    // ---- S^T = K Q^T for all keys: lane (fr, fg) holds query fr, keys nt*16 + fg*4 + r
    float4_t s[NT];
    if (PIPE) __builtin_amdgcn_sched_barrier(0);
    {
        half8_t kf[NT][2];
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            if (ABL & 2) { kf[nt][0] = qf[1]; kf[nt][1] = qf[0]; continue; }
            kf[nt][0] = *(const half8_t*)(sK + (nt * 16 + fr) * 64 + c0);
            kf[nt][1] = *(const half8_t*)(sK + (nt * 16 + fr) * 64 + c1);
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            s[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[nt][0], qf[0], zero, 0, 0, 0);
            s[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf[nt][1], qf[1], s[nt], 0, 0, 0);
        }
        if (PIPE) {
            __builtin_amdgcn_sched_group_barrier(0x100, 2 * PD, 0);
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                if (nt + PD < NT) __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    // V^T fragments of one 32-key step: lane (fr, fg) addresses key row kr, 4 d-columns
    auto read_v = [&](int s2, half4_t (&v0)[4], half4_t (&v1)[4]) {
        const bool two = !(odd_tail && s2 == NS - 1);          // the last step of an odd NT holds one sub-tile
        const int kr = s2 * 32 + fg * 4 + (fr >> 2);
#pragma unroll
        for (int dt = 0; dt < 4; ++dt) {
            const int c = dt * 2 + ((fr & 3) >> 1);
            const int off = (((c ^ (kr & 7)) << 3) + (fr & 1) * 4);          // halves; (kr + 16) & 7 == kr & 7
            v1[dt] = (half4_t){(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
            if (ABL & 2) { v0[dt] = (half4_t){qf[0][0], qf[0][1], qf[1][2], qf[1][3]}; continue; }
            v0[dt] = tr_read(sV + kr * 64 + off);
            if (two) v1[dt] = tr_read(sV + (kr + 16) * 64 + off);
        }
    };
    half4_t v0[4], v1[4];
    if (PIPE) read_v(0, v0, v1);                               // the first step's V travels under the row maximum
    if (PIPE) __builtin_amdgcn_sched_barrier(0);
    {   // keys past L (only in the last sub-tile): -inf
        const int thr = L - (NT - 1) * 16 - fg * 4;
#pragma unroll
        for (int r = 0; r < 4; ++r) s[NT - 1][r] = (r < thr) ? s[NT - 1][r] : -INFINITY;
    }
    // ---- exact row maximum, exponentials against it (raw-score domain: p = exp2(s * c - max * c), c = hd^-0.5 * log2 e)
    float mx = s[0][0];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[nt][r]);
    mx = row_max4(mx);
    const float m_ref = mx * scale_log2e;
    // ---- O^T = V^T P^T in steps of 32 keys; row sums = ones . P^T on the matrix pipe
    float4_t ol = zero;
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) o[dt] = zero;
#pragma unroll
    for (int s2 = 0; s2 < NS; ++s2) {
        const bool two = !(odd_tail && s2 == NS - 1);
        if (PIPE) __builtin_amdgcn_sched_barrier(0);
        if (!PIPE || s2 > 0) read_v(s2, v0, v1);
        if (PIPE) __builtin_amdgcn_sched_barrier(0);
        half8_t pf;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (ABL & 1) {
                pf[j] = (half_t)__builtin_fmaf(s[2 * s2][j], scale_log2e, -m_ref);
                pf[4 + j] = two ? (half_t)__builtin_fmaf(s[two ? 2 * s2 + 1 : 0][j], scale_log2e, -m_ref) : (half_t)0.f;
                continue;
            }
            pf[j] = (half_t)__builtin_amdgcn_exp2f(__builtin_fmaf(s[2 * s2][j], scale_log2e, -m_ref));
            pf[4 + j] = two ? (half_t)__builtin_amdgcn_exp2f(__builtin_fmaf(s[two ? 2 * s2 + 1 : 0][j], scale_log2e, -m_ref)) : (half_t)0.f;
        }
        if (PIPE) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int dt = 0; dt < ((ABL & 4) ? 1 : 4); ++dt) {
            const half8_t vf = {v0[dt][0], v0[dt][1], v0[dt][2], v0[dt][3], v1[dt][0], v1[dt][1], v1[dt][2], v1[dt][3]};
            o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf, o[dt], 0, 0, 0);
        }
        ol = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, pf, ol, 0, 0, 0);
    }
    if (PIPE) __builtin_amdgcn_sched_barrier(0);
    return ol[0];                                              // every d-row of ones . P^T holds the row sum of this lane's query
}

// Normalise and store one query row per lane quartet.  The accumulator layout leaves lane (fr, fg) with d = dt*16 + fg*4 + [0,4)
// for dt = 0..3: stored as it is, a wave instruction writes 8 bytes per lane, 32-byte pieces of 16 different rows (r02m: the
// 155 MB of output then cost 50 us of a 160 us kernel -- no stores 108 us, these stores 160, 64-byte pieces 138, whole rows 137).
// One v_permlane16_swap stage per register pair exchanges the dt parity with the lane's fg parity: lane (fr, fg) then holds the
// octets d0 + [0,8) and 32 + d0 + [0,8), d0 = (fg & 1)*16 + (fg >> 1)*8, i.e. 16 contiguous bytes per lane and 64 contiguous
// bytes per row in each of the two store instructions.
__device__ __forceinline__ void store_row(half_t* row, const float4_t (&o)[4], float inv, int fg) {
    unsigned w[4][2];
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
        const half2_t lo = {(half_t)(o[dt][0] * inv), (half_t)(o[dt][1] * inv)}, hi = {(half_t)(o[dt][2] * inv), (half_t)(o[dt][3] * inv)};
        w[dt][0] = __builtin_bit_cast(unsigned, lo);
        w[dt][1] = __builtin_bit_cast(unsigned, hi);
    }
    typedef unsigned uint4v __attribute__((ext_vector_type(4)));
    uint4v a, b;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        auto p = __builtin_amdgcn_permlane16_swap(w[0][k], w[1][k], false, false);
        auto q = __builtin_amdgcn_permlane16_swap(w[2][k], w[3][k], false, false);
        a[k] = (unsigned)p[0]; a[2 + k] = (unsigned)p[1];
        b[k] = (unsigned)q[0]; b[2 + k] = (unsigned)q[1];
    }
    const int d0 = (fg & 1) * 16 + (fg >> 1) * 8;
    *(uint4v*)(row + d0) = a;
    *(uint4v*)(row + 32 + d0) = b;
}

}  // namespace attn_sp
