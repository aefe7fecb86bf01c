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
//   * wave w takes query tile w of the head (13 tiles, the 14th wave only helps staging); the tile body is attn_single_pass.h:
//     26 MFMAs give the whole S^T row block, the EXACT row maximum (no running maximum, no rescale branch), 52 exponentials,
//     P^T packed to fp16 straight into the B operands of the 7 PV steps, V^T through ds_read_b64_tr_b16, the row sum from the
//     matrix pipe, output rows exchanged between lanes so that a store instruction writes 64 contiguous bytes per row;
//   * 14 waves per CU at <= 128 VGPRs (3.5 per SIMD).
// What bounds it (r02m, timing-only ablations of variant 4, same tile body; tools/attn_bench.py): the memory system.  At
// 512 x 12 heads x 197 the kernel moves 620 MB (Q, K, V once in, O once out) in 135-141 us = 4.4-4.6 TB/s; without the output
// stores it takes 108-111 us, streaming K / V alone 55 us (5.6 TB/s), and the tile body hardly matters (no exponentials -4 us,
// no LDS reads -18, one PV MFMA per step -11).  Pinned instruction order, free-running waves (variant 4), a third buffer and
// head-major strides all land within +-3 % of this kernel.
// Other shapes (text, the CLS-only last block, ViT-L) stay on variants 0 / 1.
#include "attn_single_pass.h"

#include <algorithm>
#include <cstdlib>

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
// LDS-DMA issued from inline asm (M0 = LDS byte address of the wave's 1 KiB piece, saved and restored inside the statement):
// hipcc then does not know an LDS write is in flight.  With the builtin it put an `s_waitcnt vmcnt(0)` in front of the first
// transposing V read of every head -- i.e. it waited for the NEXT head's whole K / V stream before the PV products, which is the
// overlap this kernel exists for.  Ordering is by hand: counted wait + barrier at the top of the head loop.
template <bool STREAM>
__device__ __forceinline__ void glds16_asm(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    if constexpr (STREAM)      // (experiment build: K / V with the nontemporal policy, so that they do not evict the attention output from the caches)
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off nt\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
    else
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

template <int NT, bool STREAM = false>                 // key sub-tiles of 16: (NT - 1) * 16 < L <= NT * 16
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
            glds16_asm<STREAM>(base + (1 + isv) * D + (long)kc * ld + schunk, dst);
        }
    };
    auto load_q = [&](const half_t* base, half8_t (&q)[2]) {
        const int qc = min(wave * 16 + fr, L - 1);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) q[ks] = *(const half8_t*)(base + (long)qc * ld + ks * 32 + fg * 8);
    };

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
            float4_t o[4];
            const float lsum = attn_sp::tile<NT, 0>(sK, sV, qf, L, scale_log2e, fr, fg, o);
            const int qrow = wave * 16 + fr;
            if (qrow < L) {
                const float inv = 1.0f / lsum;
                attn_sp::store_row(out + ((long)(bh / H) * L + qrow) * D + (bh % H) * 64, o, inv, fg);
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
#ifdef OVMR_EXPERIMENTS
    static const bool stream_kv = getenv("OVMR_ATTN3_NT") && atoi(getenv("OVMR_ATTN3_NT"));
    if (stream_kv) {
        static bool set2[OVMR_MAX_DEVICES] = {};
        if (!set2[dev]) {
            HIP_CHECK_RET(hipFuncSetAttribute((const void*)attn_f16_v3<NT, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            set2[dev] = true;
        }
        hipLaunchKernelGGL((attn_f16_v3<NT, true>), dim3((unsigned)grid), dim3(896), lds, s, qkv, out, L, H, nBH, sl2e);
        return (int)hipGetLastError();
    }
#endif
    hipLaunchKernelGGL((attn_f16_v3<NT>), dim3((unsigned)grid), dim3(896), lds, s, qkv, out, L, H, nBH, sl2e);
    return (int)hipGetLastError();
}

