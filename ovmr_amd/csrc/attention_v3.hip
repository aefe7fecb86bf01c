// Attention variant 3 (fp16, hd = 64, non-causal, 192 < L <= 208: the ViT-B/16 image tower, L = 197): ONE PASS over the keys.
//
// Variants 0 / 1 are flash-style: 64-key blocks, an online softmax whose running maximum / rescale / row-sum state is carried
// from block to block, one workgroup barrier per block.  At L = 197 that machinery buys nothing -- a 16-row query tile against
// ALL 13 key sub-tiles is only 52 score registers per lane -- and it costs: profiles/r02f_pmc_attn.json shows variant 1 neither
// MFMA- nor VALU-issue-bound (matrix pipe 24 % busy, VALU 57 %) but waiting: 33 % of the wave cycles in s_waitcnt / barriers
// and 37 % in issue stalls of short dependent chains.  Here
//   * PERSISTENT: one 14-wave workgroup per CU walks the (image, head) pairs.  K and V of a head (13 x 16 rows x 128 B each,
//     52 KiB) are brought into LDS once by LDS-DMA (source-side XOR swizzle as in variant 1) into one of TWO buffers: the next
//     head streams in under the current head's arithmetic; ONE barrier per head.  (A first version with one 7-wave workgroup
//     per head, 2 per CU, ran at 154 us against 162 for variant 1: load and compute phases of comparable length, half hidden.)
//   * wave w takes query tile w of the head (13 tiles, the 14th wave only helps staging); per tile: 26 MFMAs give the whole S^T row block (keys on the accumulator rows,
//     the query on the lane: a row's 208 scores sit in four lanes x 52 registers), the EXACT row maximum (no running maximum,
//     no rescale branch), 52 exponentials, P^T packed to fp16 straight into the B operands of the 7 PV steps, V^T through
//     ds_read_b64_tr_b16, the row sum from the matrix pipe (ones . P^T) -- long straight-line code with 13-26 independent
//     chains instead of 4-key-block rounds;
//   * 14 waves per CU at <= 128 VGPRs (3.5 per SIMD).
// Other shapes (text, the CLS-only last block, ViT-L) stay on variants 0 / 1.
#include "common.h"

#include <algorithm>

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef short short4v __attribute__((__vector_size__(8)));

__device__ __forceinline__ half4_t tr_read3(const half_t* p) {
    short4v r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) short4v*)p);
    return __builtin_bit_cast(half4_t, r);
}

// max over the four lanes {l, l^16, l^32, l^48} that hold one query's scores (see attention_v1.hip)
__device__ __forceinline__ float row_max4_v3(float x) {
    const unsigned u = __builtin_bit_cast(unsigned, x);
    auto a = __builtin_amdgcn_permlane16_swap(u, u, false, false);
    x = fmaxf(__builtin_bit_cast(float, (unsigned)a[0]), __builtin_bit_cast(float, (unsigned)a[1]));
    const unsigned w = __builtin_bit_cast(unsigned, x);
    auto b = __builtin_amdgcn_permlane32_swap(w, w, false, false);
    return fmaxf(__builtin_bit_cast(float, (unsigned)b[0]), __builtin_bit_cast(float, (unsigned)b[1]));
}

// LDS-DMA issued from inline asm (M0 = LDS byte address of the wave's 1 KiB piece, saved and restored inside the statement):
// hipcc then does not know an LDS write is in flight.  With the builtin it put an `s_waitcnt vmcnt(0)` in front of the first
// transposing V read of every head -- i.e. it waited for the NEXT head's whole K / V stream before the PV products, which is the
// overlap this kernel exists for.  Ordering is by hand: counted wait + barrier at the top of the head loop.
__device__ __forceinline__ void glds16_asm(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int NT>                                      // key sub-tiles of 16: (NT - 1) * 16 < L <= NT * 16
__global__ __launch_bounds__(896, 4) void attn_f16_v3(const half_t* __restrict__ qkv, half_t* __restrict__ out,
                                                      int L, int H, int nBH, float scale_log2e) {
    constexpr int ROWS = NT * 16, NW = 14, NS = (NT + 1) / 2;    // NS: PV steps of 32 keys
    constexpr int HEAD = 2 * ROWS * 64;                         // halves per buffer: K rows, then V rows
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];   // [2 buffers][K | V][ROWS][64]: 128-byte rows, chunk c of row r in slot c ^ (r & 7)
    half_t* smem = (half_t*)smem_raw;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int D = H * 64, ld = 3 * D;
    const int srow = lane >> 3, schunk = ((lane & 7) ^ srow) * 8;
    const bool has_tile = wave * 16 < L;                       // NT tiles, NW >= NT waves: wave w owns query tile w

    auto head_base = [&](int bh) { return qkv + (long)(bh / H) * L * ld + (bh % H) * 64; };
    // 4 * NT LDS-DMA instructions of 8 rows x 128 B per head, dealt round-robin to the waves
    const unsigned lds_base = (unsigned)(uintptr_t)(lptr_t)smem;
    auto stage = [&](int buf, const half_t* base) {
        for (int ins = wave; ins < 4 * NT; ins += NW) {
            const int isv = ins >= 2 * NT, r0 = (isv ? ins - 2 * NT : ins) * 8;
            const int kc = min(r0 + srow, L - 1);              // rows past the last key repeat it: finite values, masked below
            const unsigned dst = __builtin_amdgcn_readfirstlane(lds_base + 2u * (unsigned)(buf * HEAD + isv * (ROWS * 64) + r0 * 64));
            glds16_asm(base + (1 + isv) * D + (long)kc * ld + schunk, dst);
        }
    };
    auto load_q = [&](const half_t* base, half8_t (&q)[2]) {
        const int qc = min(wave * 16 + fr, L - 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) q[ks] = *(const half8_t*)(base + (long)qc * ld + ks * 32 + fg * 8);
    };

    half8_t ones;
#pragma unroll
    for (int j = 0; j < 8; ++j) ones[j] = (half_t)1.f;
    const float4_t zero = {0.f, 0.f, 0.f, 0.f};
    const int c0 = (fg ^ (fr & 7)) << 3, c1 = ((4 + fg) ^ (fr & 7)) << 3;

    int bh = blockIdx.x;
    if (bh >= nBH) return;
    half8_t qf[2], qn[2];
    stage(0, head_base(bh));
    load_q(head_base(bh), qf);
    for (int it = 0; bh < nBH; ++it, bh += gridDim.x) {
        const int cur = it & 1;
        // this head's K / V (and the query rows) have landed for every wave, and every wave is done with the other buffer
        // (the builtin form, so that hipcc's own scoreboard knows the query loads are back: after an asm wait it re-waited
        // vmcnt(0) -- i.e. for the NEXT head's stream -- in front of the first MFMA)
        __builtin_amdgcn_s_waitcnt(0x0F70);                    // vmcnt(0)
        __syncthreads();
        const int nxt = bh + (int)gridDim.x;
        if (nxt < nBH) {                                       // the next head streams in under this head's arithmetic
            stage(cur ^ 1, head_base(nxt));
            load_q(head_base(nxt), qn);
        }
        if (has_tile) {
            const half_t* sK = smem + cur * HEAD;
            const half_t* sV = sK + ROWS * 64;
            // ---- S^T = K Q^T for all keys: lane (fr, fg) holds query fr, keys nt*16 + fg*4 + r
            float4_t s[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                const half8_t kf0 = *(const half8_t*)(sK + (nt * 16 + fr) * 64 + c0);
                const half8_t kf1 = *(const half8_t*)(sK + (nt * 16 + fr) * 64 + c1);
                s[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf0, qf[0], zero, 0, 0, 0);
                s[nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(kf1, qf[1], s[nt], 0, 0, 0);
            }
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
            mx = row_max4_v3(mx);
            const float m_ref = mx * scale_log2e;
#pragma unroll
            for (int nt = 0; nt < NT; ++nt)
#pragma unroll
                for (int r = 0; r < 4; ++r) s[nt][r] = __builtin_amdgcn_exp2f(__builtin_fmaf(s[nt][r], scale_log2e, -m_ref));
            // ---- O^T = V^T P^T in steps of 32 keys; row sums = ones . P^T on the matrix pipe
            float4_t o[4] = {zero, zero, zero, zero}, ol = zero;
#pragma unroll
            for (int s2 = 0; s2 < NS; ++s2) {
                constexpr bool odd_tail = (NT & 1) != 0;
                const bool two = !(odd_tail && s2 == NS - 1);      // the last step of an odd NT holds one sub-tile
                half8_t pf;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    pf[j] = (half_t)s[2 * s2][j];
                    pf[4 + j] = two ? (half_t)s[two ? 2 * s2 + 1 : 0][j] : (half_t)0.f;
                }
                const int kr = s2 * 32 + fg * 4 + (fr >> 2);       // V^T fragment: lane (fr, fg) addresses key row kr, 4 d-columns
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const int c = dt * 2 + ((fr & 3) >> 1);
                    const int off = (((c ^ (kr & 7)) << 3) + (fr & 1) * 4);      // halves; (kr + 16) & 7 == kr & 7
                    const half4_t v0 = tr_read3(sV + kr * 64 + off);
                    half4_t v1 = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
                    if (two) v1 = tr_read3(sV + (kr + 16) * 64 + off);
                    const half8_t vf = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
                    o[dt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(vf, pf, o[dt], 0, 0, 0);
                }
                ol = __builtin_amdgcn_mfma_f32_16x16x32_f16(ones, pf, ol, 0, 0, 0);
            }
            const int qrow = wave * 16 + fr;
            if (qrow < L) {
                const float inv = 1.0f / ol[0];                // every d-row of ones . P^T holds the row sum of this lane's query
                half_t* op = out + ((long)(bh / H) * L + qrow) * D + (bh % H) * 64 + fg * 4;
#pragma unroll
                for (int dt = 0; dt < 4; ++dt) {
                    const half4_t w = {(half_t)(o[dt][0] * inv), (half_t)(o[dt][1] * inv), (half_t)(o[dt][2] * inv), (half_t)(o[dt][3] * inv)};
                    *(half4_t*)(op + dt * 16) = w;
                }
            }
        }
        qf[0] = qn[0];
        qf[1] = qn[1];
    }
}

}  // namespace

// returns -100 when the shape is not this kernel's (the caller falls back to variant 1)
int launch_attention_f16_v3(const half_t* qkv, half_t* out, int B, int L, int Lq, int H, int causal, hipStream_t s) {
    if (causal || Lq != L || L <= 192 || L > 208) return -100;
    constexpr int NT = 13;
    const size_t lds = (size_t)2 * 2 * NT * 16 * 64 * sizeof(half_t);             // two (K | V) buffers: 104 KiB
    static bool attr_set[OVMR_MAX_DEVICES] = {};
    static int n_cu[OVMR_MAX_DEVICES] = {};
    int dev = 0;
    HIP_CHECK_RET(hipGetDevice(&dev));
    if (dev < 0 || dev >= OVMR_MAX_DEVICES) return -100;
    if (!attr_set[dev]) {
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)attn_f16_v3<NT>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_CHECK_RET(hipDeviceGetAttribute(&n_cu[dev], hipDeviceAttributeMultiprocessorCount, dev));
        attr_set[dev] = true;
    }
    const float sl2e = 0.125f * 1.4426950408889634f;
    const int nBH = B * H;
    const int grid = std::min(nBH, std::max(1, n_cu[dev]));     // persistent: one 14-wave workgroup per CU walks the heads
    hipLaunchKernelGGL((attn_f16_v3<NT>), dim3((unsigned)grid), dim3(896), lds, s, qkv, out, L, H, nBH, sl2e);
    return (int)hipGetLastError();
}
