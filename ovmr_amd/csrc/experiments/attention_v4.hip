// Attention variant 4 (fp16, hd = 64, non-causal, 192 < L <= 208): variant 3's single pass over the keys with the workgroup
// barrier taken out.
//
// Variant 3 gives wave w query tile w of the head and meets at ONE barrier per head.  13 tiles over 4 SIMDs is 4 + 3 + 3 + 3, so
// the SIMD holding four tile waves sets the pace of every head while the other three idle at the barrier.  This variant was
// written to remove that -- and to find out what bounds the kernel once it is gone.  Answer (r02m / r02n, the timing-only modes
// at the bottom of this file): the memory system.  Balanced, barrier-free waves run at the same 135-140 us as
// variant 3; with the output stores removed 108-115 us; with the tile arithmetic removed as well 82 us.  It is kept as an
// alternative ("attn" = 4) with identical results, and as the harness of those ablations.
// Here the waves of the persistent workgroup never meet:
//   * 12 CONSUMER waves, three per SIMD, take the CU's (head, query tile) list round-robin -- tile t = c, c + 12, c + 24, ... of
//     13 x heads-per-CU tiles -- so every SIMD carries the same load whatever 13 mod 4 is, and the waves drift apart in phase,
//     which lets one wave's exponentials issue under another wave's MFMAs;
//   * 1 PRODUCER wave streams K and V of a head (52 KiB) into a ring of THREE LDS buffers (156 KiB) by LDS-DMA: twelve
//     consecutive tiles touch at most two heads, so two buffers are being read while the third fills;
//   * synchronisation is two LDS words per buffer: `ready` = 1 + index of the head that has landed (producer: s_waitcnt vmcnt(0),
//     then the store -- the LDS executes both in order), `done` = tiles finished on this buffer since the start (consumer:
//     s_waitcnt lgkmcnt(0) after its last K / V read, then ds_add).  The producer refills a buffer when `done` reaches 13 per
//     head that has lived in it.  No s_barrier after the flags are initialised.
// The per-tile arithmetic is variant 3's: 26 MFMAs for the S^T row block, exact row maximum, 52 exponentials, P^T packed into the
// B operands of 7 PV steps, row sums from the matrix pipe.
// Compiled into the EXPERIMENT build only (python -m ovmr_amd.build --experiments): same speed as variant 3, kept as the vehicle of the ablations.
#ifdef OVMR_EXPERIMENTS
#include "../attn_single_pass.h"

#include <algorithm>

namespace {

typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
// LDS-DMA from inline asm (see attention_v3.hip: with the builtin hipcc orders every LDS read behind vmcnt(0)).  Scalar base +
// 32-bit lane offset: the producer's per-instruction work is three scalar adds, no vector address arithmetic (r02m: with a
// 64-bit vector address per instruction the lone producer wave, competing with three busy consumers for its SIMD's vector port,
// needed ~3 us to ISSUE a head and 6.9 us per head in all -- the kernel ran at the producer's pace, 165 us).
__device__ __forceinline__ void glds16_s(unsigned voff, const void* sbase, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %3\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(voff), "s"(sbase), "s"(lds_dst) : "memory");
}

typedef __attribute__((address_space(3))) int lds_int_t;
__device__ __forceinline__ int lds_flag(const lds_int_t* p) {     // wave-uniform value of one LDS word (ds_read, no flat access)
    return __builtin_amdgcn_readfirstlane(*(const volatile lds_int_t*)p);
}

constexpr int NC4 = 12, NW4 = NC4 + 1, NBUF4 = 3;

template <int NT, int MODE>                            // key sub-tiles of 16: (NT - 1) * 16 < L <= NT * 16
__global__ __launch_bounds__(NW4 * 64) void attn_f16_v4(const half_t* __restrict__ qkv, half_t* __restrict__ out,
                                                        int L, int H, int nBH, float scale_log2e) {
    constexpr int ROWS = NT * 16;
    constexpr int HEAD = 2 * ROWS * 64;                         // halves per buffer: K rows, then V rows
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];   // [3 buffers][K | V][ROWS][64], then the flags
    half_t* smem = (half_t*)smem_raw;
    lds_int_t* ready = (lds_int_t*)(lptr_t)(smem + NBUF4 * HEAD);
    lds_int_t* done = ready + NBUF4;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int fr = lane & 15, fg = lane >> 4;
    const int D = H * 64, ld = 3 * D;

    if ((int)blockIdx.x >= nBH) return;
    const int nh = (nBH - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;   // heads of this workgroup
    auto head_base = [&](int hi) {
        const int bh = blockIdx.x + hi * gridDim.x;
        return qkv + (long)(bh / H) * L * ld + (bh % H) * 64;
    };
    if (tid < 2 * NBUF4) ready[tid] = 0;
    __syncthreads();                                            // the only barrier: flags are zero before anyone polls

    if (wave == NC4) {
        // ------------------------------------------------------------------------------------------------ producer
        __builtin_amdgcn_s_setprio(3);
        // one instruction = 8 rows x 128 B: lane -> (row srow of the group, 16-byte slot lane & 7 holding source chunk slot ^ srow).
        // Rows past the last key (only in the last two groups of K and of V) repeat row L - 1: finite values, masked by the consumers.
        const int srow = lane >> 3;
        const unsigned chunk_b = (unsigned)(((lane & 7) ^ srow) * 16), row_b = (unsigned)ld * 2u;
        auto lane_off = [&](int r0) { return (unsigned)(min(r0 + srow, L - 1) - min(r0, L - 1)) * row_b + chunk_b; };
        const unsigned voff = (unsigned)srow * row_b + chunk_b;
        const unsigned voff_a = lane_off((2 * NT - 2) * 8), voff_b = lane_off((2 * NT - 1) * 8);
        const unsigned lds_base = (unsigned)(uintptr_t)(lptr_t)smem;
        for (int hi = 0; hi < nh; ++hi) {
            const int buf = hi % NBUF4;
            // every tile of the heads that lived in this buffer before is finished
            const int need = __builtin_amdgcn_readfirstlane(NT * (hi / NBUF4));
            while (lds_flag(done + buf) < need) __builtin_amdgcn_s_sleep(2);
            asm volatile("" ::: "memory");
            const uint64_t b64 = (uint64_t)head_base(hi);       // wave-uniform; readfirstlane so that it provably lives in SGPRs
            const char* base = (const char*)(((uint64_t)(unsigned)__builtin_amdgcn_readfirstlane((int)(b64 >> 32)) << 32) |
                                             (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b64));
            const unsigned dst0 = __builtin_amdgcn_readfirstlane(lds_base + 2u * (unsigned)(buf * HEAD));
#pragma unroll
            for (int ins = 0; ins < 4 * NT; ++ins) {
                const int isv = ins >= 2 * NT, g = isv ? ins - 2 * NT : ins, r0 = g * 8;
                const char* src = base + (long)(1 + isv) * D * 2 + (long)min(r0, L - 1) * row_b;
                const unsigned vo = g == 2 * NT - 2 ? voff_a : g == 2 * NT - 1 ? voff_b : voff;
                glds16_s(vo, src, dst0 + 2u * (unsigned)(isv * (ROWS * 64) + r0 * 64));
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) *(volatile lds_int_t*)(ready + buf) = hi + 1;
        }
        return;
    }

    // ---------------------------------------------------------------------------------------------------- consumers
    const int ntiles = NT * nh;
    auto load_q = [&](int t, half8_t (&q)[2]) {
        const int hi = t / NT, ti = t - hi * NT;
        const int qc = min(ti * 16 + fr, L - 1);
        const half_t* base = head_base(hi);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) q[ks] = *(const half8_t*)(base + (long)qc * ld + ks * 32 + fg * 8);
    };
    int t = wave;
    if (t >= ntiles) return;
    // VMEM traffic of a consumer is the same straight line every tile -- two query loads for the NEXT tile at the top, two
    // stores at the end -- so the wait hipcc inserts in front of `qf = qn` is a counted vmcnt(2): the stores stay in
    // flight.  (Conditional loads / stores made hipcc wait vmcnt(0) there: query latency exposed on every tile.)
    half8_t qf[2], qn[2];
    load_q(t, qf);
    __builtin_amdgcn_s_waitcnt(0x0F70);                         // vmcnt(0), the builtin form: hipcc's scoreboard sees the loop entered clean
    for (; t < ntiles; t += NC4) {
        const int hi = t / NT, ti = t - hi * NT;
        const int buf = hi % NBUF4;
        load_q(min(t + NC4, ntiles - 1), qn);                   // the next tile's query rows travel under this tile
        if (!(MODE & 4)) while (lds_flag(ready + buf) != hi + 1) __builtin_amdgcn_s_sleep(1);   // MODE 4 / 2: timing-only ablations
        asm volatile("" ::: "memory");
        if (MODE & 2) {
            if (lane == 0) __hip_atomic_fetch_add(done + buf, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            continue;
        }
        const half_t* sK = smem + buf * HEAD;
        const half_t* sV = sK + ROWS * 64;
        float4_t o[4];
        const float lsum = attn_sp::tile<NT, (MODE & 1) | ((MODE >> 3) << 1)>(sK, sV, qf, L, scale_log2e, fr, fg, o);
        // every K / V fragment of this tile is in registers: hand the tile back to the producer
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) __hip_atomic_fetch_add(done + buf, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        {
            // rows past L (last tile) were computed on query row L - 1: they store that row's values again, unconditionally
            const int qrow = min(ti * 16 + fr, L - 1);
            const int bh = blockIdx.x + hi * gridDim.x;
            const float inv = 1.0f / lsum;
            half_t* row = out + ((long)(bh / H) * L + qrow) * D + (bh % H) * 64;
            if (!(MODE & 64) || lsum == 12345.678f) attn_sp::store_row(row, o, inv, fg);     // MODE 64: timing-only, no stores
        }
        qf[0] = qn[0];
        qf[1] = qn[1];
    }
}

template <int MODE>
int launch_v4_mode(const half_t* qkv, half_t* out, int B, int L, int H, hipStream_t s) {
    constexpr int NT = 13;
    const size_t lds = (size_t)NBUF4 * 2 * NT * 16 * 64 * sizeof(half_t) + 64;   // three (K | V) buffers: 156 KiB, + flags
    static bool attr_set[OVMR_MAX_DEVICES] = {};
    static int n_cu[OVMR_MAX_DEVICES] = {};
    int dev = 0;
    HIP_CHECK_RET(hipGetDevice(&dev));
    if (dev < 0 || dev >= OVMR_MAX_DEVICES) return -100;
    if (!attr_set[dev]) {
        HIP_CHECK_RET(hipFuncSetAttribute((const void*)attn_f16_v4<NT, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        HIP_CHECK_RET(hipDeviceGetAttribute(&n_cu[dev], hipDeviceAttributeMultiprocessorCount, dev));
        attr_set[dev] = true;
    }
    const float sl2e = 0.125f * 1.4426950408889634f;
    const int nBH = B * H;
    const int grid = std::min(nBH, std::max(1, n_cu[dev]));     // persistent: one 13-wave workgroup per CU walks the heads
    hipLaunchKernelGGL((attn_f16_v4<NT, MODE>), dim3((unsigned)grid), dim3(NW4 * 64), lds, s, qkv, out, L, H, nBH, sl2e);
    return (int)hipGetLastError();
}

}  // namespace

// returns -100 when the shape is not this kernel's (the caller falls back to variant 1)
int launch_attention_f16_v4(const half_t* qkv, half_t* out, int B, int L, int Lq, int H, int causal, int mode, hipStream_t s) {
    if (causal || Lq != L || L <= 192 || L > 208) return -100;
#ifdef OVMR_EXPERIMENTS
    // timing-only ablations (outputs wrong by construction), tools/attn_bench.py --variants 4 402 404 464 ...:
    //   1 pinned instruction order in the tile body, 2 consumers skip the arithmetic (streaming alone), 4 consumers do not wait
    //   for the producer, 8 / 16 / 32 tile body without exponentials / without LDS reads / with one PV MFMA per step, 64 no stores
    switch (mode) {
        case 1: return launch_v4_mode<1>(qkv, out, B, L, H, s);
        case 2: return launch_v4_mode<2>(qkv, out, B, L, H, s);
        case 4: return launch_v4_mode<4>(qkv, out, B, L, H, s);
        case 12: return launch_v4_mode<4 | 8>(qkv, out, B, L, H, s);
        case 20: return launch_v4_mode<4 | 16>(qkv, out, B, L, H, s);
        case 36: return launch_v4_mode<4 | 32>(qkv, out, B, L, H, s);
        case 60: return launch_v4_mode<4 | 8 | 16 | 32>(qkv, out, B, L, H, s);
        case 64: return launch_v4_mode<64>(qkv, out, B, L, H, s);
        case 68: return launch_v4_mode<4 | 64>(qkv, out, B, L, H, s);
        case 124: return launch_v4_mode<4 | 8 | 16 | 32 | 64>(qkv, out, B, L, H, s);
        default: break;
    }
#endif
    if (mode != 0) return -5;
    return launch_v4_mode<0>(qkv, out, B, L, H, s);
}

#endif  // OVMR_EXPERIMENTS
