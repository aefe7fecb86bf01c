// One-launch classifier head (trainers/mm_classifier_one_prompt.py:348-363, trainers/zsclip.py:55-60).
//
//   out[b][c] = sum_m w[c][m] * softmax_c( float( h( h(scale * f[b]) . clf_m[c] ) ) )        m = mm, vision, text
//
// The reference rounds (logit_scale * image_features) to fp16 before the matmul and every logit to fp16 after it; `.float()` and
// the softmax follow (:357-363).  Both rounding points are kept.  What is NOT kept are the three fp16 logit tensors in HBM and the
// five launches (scale, three GEMMs, softmax) of the first implementation:
//
//   * a workgroup works on tiles of BM query rows x 128 classes, all (up to three) classifiers at once: the scaled, rounded
//     features of its rows are staged ONCE in LDS and shared by the three products (the shared-A sketch of SURVEY.md section 7-7);
//   * the products are computed transposed, S^T = clf . f^T with v_mfma_f32_32x32x16_f16, so that a lane holds ONE query and 16
//     classes per 32 x 32 tile: the row-wise softmax statistics are in-register maxima / sums plus one v_permlane32_swap (lanes
//     l and l + 32 hold the same query) -- the layout of attention_v5.hip;
//   * phase 1 leaves one (maximum, sum of exponentials) pair per (row, class tile, classifier) in a small workspace (device-coherent
//     stores: no cache-wide write-back / invalidate); phase 2 merges the pairs of a row (online-softmax merge, fixed order:
//     deterministic) and writes the weighted probabilities from the logits the workgroup STILL HOLDS IN REGISTERS.  Up to 16 class
//     tiles (2048 classes) every workgroup merges its own rows' pairs; with more tiles that re-reads O(tiles^2) pairs per row tile
//     (256 x 10 000: 74 us of uncached loads), so the merge becomes a phase of its own: units of four (row, classifier) statistics,
//     one per wave with the lanes striding over the class tiles, handed out by a second ticket and counted like the tiles.
//     tools/head_bench.py, device time per call: 256 x 1000 30 us against 34 for the five launches, 64 x 1000 26 / 34, 128 x 1000 at
//     width 768 35 / 39, 256 x 10 000 101 / 117, 256 x 21 841 255 / 344; 512 x 4096 78 / 73 and 2048 x 1000 63 / 56 lose (each 64-row
//     tile re-reads the classifier matrices), hence the entry point's rule: up to 256 rows, or up to 512 rows x 2048 classes.
//   * Between them every tile must be done -- a device-wide dependency, built so that
//     it cannot starve whatever else occupies the chip: tiles are handed out by an atomic ticket, so the workgroups that ARE
//     resident work through all of them and wait on a count of finished TILES (not of arrived workgroups); a workgroup that took
//     more than one tile keeps the last one in registers and queues the earlier ones for recomputation in phase 2 (normally the
//     grid equals the tile count, every workgroup is resident, takes exactly one tile, and the queue stays empty).  The counters
//     live in device memory owned by the handle; the last workgroup to leave re-arms them -- nothing on the host, replayable from
//     a hipGraph, safe with several handles in flight, each used from ONE stream at a time (the counters and the workspace belong to the
//     handle: two launches on one handle must be stream-ordered -- include/ovmr_hip.h says so for every entry point);
//   * raw mode (zero-shot CLIP, one classifier, fp16 logits out): one pass, no counters.
#include "common.h"

#include <algorithm>
#ifdef OVMR_EXPERIMENTS
#include <cstdlib>
#endif

namespace {

#ifdef OVMR_EXPERIMENTS
inline int exp_env(const char* name) { const char* e = getenv(name); return e ? atoi(e) : 0; }
#endif

typedef float float16_t __attribute__((ext_vector_type(16)));

constexpr int HF_BN = 128;                 // classes per tile: 4 waves x 32
enum { HF_TICKET = 0, HF_DONE = 1, HF_LEFT_N = 2, HF_LEFT_POP = 3, HF_EXIT = 4, HF_DUTY_TICKET = 5, HF_DUTY_DONE = 6, HF_SYNC_INTS = 16 };
constexpr int HF_LOCAL_MERGE_MAX = 16;     // class tiles up to which every workgroup merges its rows' per-tile statistics itself (C <= 2048)

#ifdef OVMR_EXPERIMENTS
// shader-clock stamps of workgroup 0 (tools/head_bench.py --stamps): where a launch spends its time
__device__ long long g_head_stamps[16];
#define HF_STAMP(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) g_head_stamps[i] = __builtin_amdgcn_s_memtime(); } while (0)
#else
#define HF_STAMP(i)
#endif

__device__ __forceinline__ int aload(int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void astore(float* p, float v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ float aloadf(const float* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ int aadd(int* p, int v) { return __hip_atomic_fetch_add(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// "This tile / unit is done" and "everything is done": the orderings of the device-wide phases.  The data they order (per-tile statistics,
// merged statistics, the recompute queue) are written and read with agent-scope atomic accesses, which on gfx950 go to device-coherent
// memory by themselves; the product build orders them against the counters the way the ISA does -- every thread drains its stores
// (s_waitcnt vmcnt(0)), the workgroup meets at a barrier, one thread bumps the counter; readers spin on the counter, meet at a barrier,
// then issue their loads -- with RELAXED atomics, i.e. without the cache-wide write-back / invalidate a release / acquire pair at agent
// scope compiles to (buffer_wbl2 sc1 / buffer_inv sc1: measured 3 us of a 30 us launch at 1000 classes, 33 us at 10 000, round 5).
// ACQREL = true is that formally ordered form (C++ memory model: release on the increments, acquire after the spins); the experiment
// build selects it with OVMR_HEAD_ACQREL=1 so that its cost stays measurable (tools/head_bench.py --acqrel; profiles/r06*_head_acqrel.log)
// and tools/race_screen.py screens the relaxed form the product ships.
template <bool ACQREL>
__device__ __forceinline__ void hf_signal(int* counter) {
    if constexpr (ACQREL) __hip_atomic_fetch_add(counter, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    else aadd(counter, 1);
}
template <bool ACQREL>
__device__ __forceinline__ void hf_wait(int* counter, int target) {
    while (aload(counter) < target) __builtin_amdgcn_s_sleep(2);
    if constexpr (ACQREL) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

// BM: query rows per tile (32 or 64).  RAW: one classifier, fp16 logits out, no softmax.
template <int BM, bool RAW, bool ACQREL = false>
__global__ __launch_bounds__(256, 2) void head_fused_kernel(const half_t* __restrict__ feats, int B, int D, float scale,
                                                         const half_t* __restrict__ c0, const half_t* __restrict__ c1,
                                                         const half_t* __restrict__ c2, int n_mod, int C,
                                                         const float* __restrict__ w, float* __restrict__ out,
                                                         half_t* __restrict__ raw_out, float* __restrict__ partial, float* __restrict__ merged,
                                                         int* __restrict__ leftover, int* sync, int Tc, int n_tiles) {
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    constexpr int QB = BM / 32;                              // 32-query blocks per tile
    const int ldf = D + 8;                                   // LDS row stride in halves: + 16 B, so that the 16 lanes of a ds_read_b128 pass hit 64 distinct banks
    half_t* sf = (half_t*)smem_raw;                          // [BM][ldf] scaled features
    float* red = (float*)(smem_raw + (size_t)BM * ldf * 2);  // [4 waves][3][BM][2] per-wave (max, sum); reused as [3][BM][2] final (max, 1 / sum)
    __shared__ int sh_t;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int j = lane & 31, h = lane >> 5;
    const int rows_pad = (n_tiles / Tc) * BM;
    const half_t* clf[3] = {c0, c1, c2};
    float16_t acc[3][QB];
    int staged_tr = -1;

    // The logits of tile t = (row tile tr, class tile tc) -> acc, rounded to fp16 (classes past C: -inf).
    // Lane (j, h) holds query tr * BM + qb * 32 + j and the classes cw0 + (k & 3) + 8 (k >> 2) + 4 h, k = 0..15.
    auto compute = [&](int t) {
        const int tc = t % Tc, tr = t / Tc, r0 = tr * BM;
        const int cw0 = tc * HF_BN + wave * 32;
        if (tr != staged_tr) {                               // h(scale * f) of the BM rows (zeros past the last row)
            __syncthreads();
            for (int i = tid; i < BM * (D / 8); i += 256) {
                const int row = i / (D / 8), ch = i % (D / 8);
                half8_t v;
                if (r0 + row < B) {
                    v = *(const half8_t*)(feats + (long)(r0 + row) * D + ch * 8);
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = (half_t)(scale * (float)v[k]);     // (logit_scale * image_features) in fp16 (:358-360)
                } else {
#pragma unroll
                    for (int k = 0; k < 8; ++k) v[k] = (half_t)0.f;
                }
                *(half8_t*)(sf + row * ldf + ch * 8) = v;
            }
            __syncthreads();
            staged_tr = tr;
        }
        const int crow = min(cw0 + j, C - 1);                // A operand row (a class); rows past the last class repeat it and are masked below
#pragma unroll
        for (int m = 0; m < 3; ++m)
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                for (int k = 0; k < 16; ++k) acc[m][qb][k] = 0.f;
        // K loop, 64 columns (four k-steps) per trip, all classifiers together: the 16-byte fragments of the NEXT trip (up to 12 loads
        // per lane) are in flight while this trip's products run -- the fragments come straight from global memory (each classifier row
        // is read once per workgroup, 32 bytes per lane pair and k-step) and their latency is what bounds a tile
        const half_t* arow[3];
#pragma unroll
        for (int m = 0; m < 3; ++m) arow[m] = clf[m < n_mod ? m : 0] + (long)crow * D + h * 8;
        constexpr int KU = BM == 32 ? 4 : 2;                 // k-steps per trip (64-row tiles: two, their accumulators take the registers)
        half8_t an[3][KU];
#pragma unroll
        for (int m = 0; m < 3; ++m)
            if (m < n_mod) {
#pragma unroll
                for (int u = 0; u < KU; ++u) an[m][u] = *(const half8_t*)(arow[m] + u * 16);
            }
        for (int k0 = 0; k0 < D; k0 += 16 * KU) {            // (D is a multiple of 64)
            half8_t a[3][KU];
#pragma unroll
            for (int m = 0; m < 3; ++m)
#pragma unroll
                for (int u = 0; u < KU; ++u) a[m][u] = an[m][u];
            if (k0 + 16 * KU < D) {
#pragma unroll
                for (int m = 0; m < 3; ++m)
                    if (m < n_mod) {
#pragma unroll
                        for (int u = 0; u < KU; ++u) an[m][u] = *(const half8_t*)(arow[m] + k0 + 16 * KU + u * 16);
                    }
            }
#pragma unroll
            for (int u = 0; u < KU; ++u)
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    const half8_t b = *(const half8_t*)(sf + (qb * 32 + j) * ldf + k0 + u * 16 + h * 8);
#pragma unroll
                    for (int m = 0; m < 3; ++m)
                        if (m < n_mod) acc[m][qb] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[m][u], b, acc[m][qb], 0, 0, 0);
                }
        }
#pragma unroll
        for (int m = 0; m < 3; ++m) {
            if (m >= n_mod) break;
#pragma unroll
            for (int qb = 0; qb < QB; ++qb)
#pragma unroll
                for (int k = 0; k < 16; ++k) {
                    const int c = cw0 + (k & 3) + 8 * (k >> 2) + 4 * h;
                    const float l = (float)(half_t)acc[m][qb][k];       // the fp16 rounding of every logit
                    acc[m][qb][k] = c < C ? l : -INFINITY;
                }
        }
    };

    if constexpr (RAW) {
        const int t = blockIdx.x;
        compute(t);
        const int r0 = (t / Tc) * BM, cw0 = (t % Tc) * HF_BN + wave * 32;
#pragma unroll
        for (int qb = 0; qb < QB; ++qb) {
            const int row = r0 + qb * 32 + j;
            if (row >= B) continue;
#pragma unroll
            for (int kq = 0; kq < 4; ++kq) {
                const int c = cw0 + 8 * kq + 4 * h;
                half_t* dst = raw_out + (long)row * C + c;
                if (c + 3 < C && (C & 3) == 0) {
                    *(half4_t*)dst = (half4_t){(half_t)acc[0][qb][4 * kq], (half_t)acc[0][qb][4 * kq + 1], (half_t)acc[0][qb][4 * kq + 2],
                                               (half_t)acc[0][qb][4 * kq + 3]};
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (c + e < C) dst[e] = (half_t)acc[0][qb][4 * kq + e];
                }
            }
        }
        return;
    } else {
        // phase 1 of a tile: (maximum, sum of exponentials) per (classifier, query) over the tile's 128 classes -> partial
        auto stats = [&](int t) {
            const int tc = t % Tc, r0 = (t / Tc) * BM;
#pragma unroll
            for (int m = 0; m < 3; ++m) {
                if (m >= n_mod) break;
#pragma unroll
                for (int qb = 0; qb < QB; ++qb) {
                    float mx = acc[m][qb][0];
#pragma unroll
                    for (int k = 1; k < 16; ++k) mx = fmaxf(mx, acc[m][qb][k]);
                    {
                        const unsigned u = __builtin_bit_cast(unsigned, mx);
                        auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                        mx = fmaxf(__builtin_bit_cast(float, (unsigned)sw[0]), __builtin_bit_cast(float, (unsigned)sw[1]));
                    }
                    float sum = 0.f;
                    if (mx > -INFINITY) {
#pragma unroll
                        for (int k = 0; k < 16; ++k) sum += __expf(acc[m][qb][k] - mx);
                    }
                    {
                        const unsigned u = __builtin_bit_cast(unsigned, sum);
                        auto sw = __builtin_amdgcn_permlane32_swap(u, u, false, false);
                        sum = __builtin_bit_cast(float, (unsigned)sw[0]) + __builtin_bit_cast(float, (unsigned)sw[1]);
                    }
                    if (h == 0) {
                        float* p = red + (((wave * 3 + m) * BM) + qb * 32 + j) * 2;
                        p[0] = mx;
                        p[1] = sum;
                    }
                }
            }
            __syncthreads();
            for (int i = tid; i < n_mod * BM; i += 256) {        // the four waves' pairs merged in wave order
                const int m = i / BM, q = i % BM;
                float M = -INFINITY, S = 0.f;
#pragma unroll
                for (int wv = 0; wv < 4; ++wv) {
                    const float mx = red[(((wv * 3 + m) * BM) + q) * 2], sm = red[(((wv * 3 + m) * BM) + q) * 2 + 1];
                    const float Mn = fmaxf(M, mx);
                    if (Mn > -INFINITY) S = S * __expf(M - Mn) + sm * __expf(mx - Mn);
                    M = Mn;
                }
                float* p = partial + ((((long)m * rows_pad + r0 + q) * Tc) + tc) * 2;
                astore(p, M);                                     // device-coherent stores (they pass the XCD's L2): no cache-wide write-back
                astore(p + 1, S);
            }
        };
        // phase 2 of a tile: the rows' statistics over all class tiles (tile order), then the weighted probabilities of acc
        auto emit = [&](int t) {
            const int tc = t % Tc, r0 = (t / Tc) * BM, cw0 = tc * HF_BN + wave * 32;
            __syncthreads();
            if (Tc > HF_LOCAL_MERGE_MAX) {                        // merged once for everybody by the duty phase below
                for (int i = tid; i < n_mod * BM; i += 256) {
                    const int m = i / BM, q = i % BM;
                    const float* p = merged + ((long)m * rows_pad + r0 + q) * 2;
                    red[(m * BM + q) * 2] = aloadf(p);
                    red[(m * BM + q) * 2 + 1] = aloadf(p + 1);
                }
            } else
            for (int i = tid; i < n_mod * BM; i += 256) {
                const int m = i / BM, q = i % BM;
                const float* p = partial + (((long)m * rows_pad + r0 + q) * Tc) * 2;
                float M = -INFINITY, S = 0.f;
                for (int t0 = 0; t0 < Tc; t0 += 8) {             // eight pairs in flight, merged in tile order
                    float2_t pr[8];
#pragma unroll
                    for (int u = 0; u < 8; ++u) {                 // device-coherent loads: another XCD wrote most of these
                        const float* pp = p + 2 * min(t0 + u, Tc - 1);
                        pr[u] = (float2_t){aloadf(pp), aloadf(pp + 1)};
                    }
#pragma unroll
                    for (int u = 0; u < 8; ++u) {
                        if (t0 + u >= Tc) break;
                        const float Mn = fmaxf(M, pr[u][0]);
                        if (Mn > -INFINITY) S = S * __expf(M - Mn) + pr[u][1] * __expf(pr[u][0] - Mn);
                        M = Mn;
                    }
                }
                red[(m * BM + q) * 2] = M;
                red[(m * BM + q) * 2 + 1] = 1.0f / S;
            }
            __syncthreads();
#pragma unroll
            for (int qb = 0; qb < QB; ++qb) {
                const int q = qb * 32 + j, row = r0 + q;
                float M[3], inv[3];
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    if (m >= n_mod) break;
                    M[m] = red[(m * BM + q) * 2];
                    inv[m] = red[(m * BM + q) * 2 + 1];
                }
#pragma unroll
                for (int kq = 0; kq < 4; ++kq) {
                    const int c = cw0 + 8 * kq + 4 * h;
                    float4_t o;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        float a = 0.f;
#pragma unroll
                        for (int m = 0; m < 3; ++m) {
                            if (m >= n_mod) break;
                            const float pr = __expf(acc[m][qb][4 * kq + e] - M[m]) * inv[m];
                            a += w ? pr * w[(long)min(c + e, C - 1) * 3 + m] : pr;
                        }
                        o[e] = a;
                    }
                    if (row < B) {
                        float* dst = out + (long)row * C + c;
                        if (c + 3 < C && (C & 3) == 0) *(float4_t*)dst = o;
                        else {
#pragma unroll
                            for (int e = 0; e < 4; ++e)
                                if (c + e < C) dst[e] = o[e];
                        }
                    }
                }
            }
        };

        // ---- phase 1: tiles by ticket; the last one taken stays in registers, earlier ones are queued for phase 2
        HF_STAMP(0);
        int held = -1;
        for (;;) {
            __syncthreads();
            if (tid == 0) sh_t = aadd(sync + HF_TICKET, 1);
            __syncthreads();
            const int t = sh_t;
            if (t >= n_tiles) break;
            if (held >= 0 && tid == 0) __hip_atomic_store(leftover + aadd(sync + HF_LEFT_N, 1), held, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            HF_STAMP(1);
            compute(t);
            HF_STAMP(2);
            stats(t);
            HF_STAMP(3);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every thread: its pairs (and the queue entry) have reached device-coherent memory ...
            __syncthreads();
            if (tid == 0) hf_signal<ACQREL>(sync + HF_DONE);          // ... before the tile counts as done
            held = t;
        }
        HF_STAMP(4);
        if (tid == 0) hf_wait<ACQREL>(sync + HF_DONE, n_tiles);
        __syncthreads();                                              // (the pairs are read with device-coherent loads: nothing to invalidate)
        HF_STAMP(5);
        if (Tc > HF_LOCAL_MERGE_MAX) {
            // Many class tiles: every workgroup of a row tile merging ALL of that row tile's pairs itself re-reads O(tiles^2) pairs
            // (256 x 10 000: 74 us of uncached loads).  Instead the (row, classifier) statistics are merged ONCE: units of four of
            // them (one per wave, the wave's lanes striding over the class tiles, then a fixed-order butterfly: deterministic) are
            // handed out by a second ticket, and a second count of finished units orders the merged values before phase 2.
            const int n_pairs = n_mod * rows_pad, n_units = (n_pairs + 3) / 4;
            for (;;) {
                __syncthreads();
                if (tid == 0) sh_t = aadd(sync + HF_DUTY_TICKET, 1);
                __syncthreads();
                const int u = sh_t;
                if (u >= n_units) break;
                const int pr = u * 4 + wave;
                if (pr < n_pairs) {
                    const float* p = partial + (long)pr * Tc * 2;          // pair index = m * rows_pad + row: the layout of `partial`
                    float M = -INFINITY, S = 0.f;
                    for (int t = lane; t < Tc; t += 64) {
                        const float mx = aloadf(p + 2 * t), sm = aloadf(p + 2 * t + 1);
                        const float Mn = fmaxf(M, mx);
                        if (Mn > -INFINITY) S = S * __expf(M - Mn) + sm * __expf(mx - Mn);
                        M = Mn;
                    }
#pragma unroll
                    for (int o = 1; o < 64; o <<= 1) {
                        const float Mo = __shfl_xor(M, o, 64), So = __shfl_xor(S, o, 64);
                        const float Mn = fmaxf(M, Mo);
                        // (both partners compute the same sum: a + b with a, b swapped -- IEEE addition commutes, the lanes stay identical)
                        const float a = Mn > -INFINITY ? S * __expf(M - Mn) : 0.f, b = Mn > -INFINITY ? So * __expf(Mo - Mn) : 0.f;
                        S = (lane & o) ? b + a : a + b;
                        M = Mn;
                    }
                    if (lane == 0) {
                        astore(merged + (long)pr * 2, M);
                        astore(merged + (long)pr * 2 + 1, 1.0f / S);
                    }
                }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) hf_signal<ACQREL>(sync + HF_DUTY_DONE);
            }
            if (tid == 0) hf_wait<ACQREL>(sync + HF_DUTY_DONE, n_units);
            __syncthreads();
        }
        HF_STAMP(6);
        // ---- phase 2
        if (held >= 0) emit(held);
        HF_STAMP(7);
        for (;;) {
            __syncthreads();
            if (tid == 0) {
                const int n = aload(sync + HF_LEFT_N);                // final: every push happened before its workgroup's next `done`
                const int i = n > 0 ? aadd(sync + HF_LEFT_POP, 1) : 0;
                sh_t = (n > 0 && i < n) ? __hip_atomic_load(leftover + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : -1;
            }
            __syncthreads();
            const int t = sh_t;
            if (t < 0) break;
            compute(t);
            emit(t);
        }
        // ---- the last workgroup to leave re-arms the counters for the next launch on this handle
        if (tid == 0 && aadd(sync + HF_EXIT, 1) == (int)gridDim.x - 1) {
#pragma unroll
            for (int i = 0; i < 8; ++i) __hip_atomic_store(sync + i, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        HF_STAMP(8);
    }
}

template <int BM, bool RAW>
int launch_one(const half_t* feats, int B, int D, float scale, const half_t* const* clf, int n_mod, int C, const float* w,
               float* out, half_t* raw_out, float* partial, float* merged, int* leftover, int* sync, int Tc, size_t lds, int max_grid, hipStream_t s) {
    const int Tr = (B + BM - 1) / BM, n_tiles = Tr * Tc;
    auto kern = head_fused_kernel<BM, RAW>;
#ifdef OVMR_EXPERIMENTS
    static const bool acqrel = exp_env("OVMR_HEAD_ACQREL") == 1;      // the release / acquire form of the phase counters (A/B: its cost)
    if (acqrel && !RAW) kern = head_fused_kernel<BM, RAW, true>;
#endif
    static size_t lds_set[OVMR_MAX_DEVICES] = {};            // per (BM, RAW) and device: the largest dynamic LDS size granted so far
    if (lds > 64 * 1024) {
        int dev = 0;
        HIP_CHECK_RET(hipGetDevice(&dev));
        if (dev < 0 || dev >= OVMR_MAX_DEVICES) return -100;
        if (lds > lds_set[dev]) {
            HIP_CHECK_RET(hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
#ifdef OVMR_EXPERIMENTS
            HIP_CHECK_RET(hipFuncSetAttribute((const void*)head_fused_kernel<BM, RAW, !RAW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
#endif
            lds_set[dev] = lds;
        }
    }
    const int grid = RAW ? n_tiles : std::max(1, std::min(n_tiles, max_grid > 0 ? max_grid : n_tiles));
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(256), lds, s, feats, B, D, scale, clf[0], n_mod > 1 ? clf[1] : nullptr,
                       n_mod > 2 ? clf[2] : nullptr, n_mod, C, w, out, raw_out, partial, merged, leftover, sync, Tc, n_tiles);
    return (int)hipGetLastError();
}

size_t head_lds_bytes(int BM, int D) { return (size_t)BM * (D + 8) * 2 + (size_t)4 * 3 * BM * 2 * 4; }

}  // namespace

// Workspace the launch needs for B rows: floats [3][rows padded to 64][Tc][2] (tile pairs), [3][rows][2] (merged), then ints
// [rows / 32 x Tc] (the recompute queue).
size_t head_fused_ws_bytes(int B, int C) {
    const size_t Tc = (C + HF_BN - 1) / HF_BN, rows = (size_t)(B + 63) / 64 * 64;
    return 3 * rows * (Tc + 1) * 2 * sizeof(float) + (rows / 32) * Tc * sizeof(int);
}
int head_fused_sync_ints() { return HF_SYNC_INTS; }

// -100: shape not taken (D not a multiple of 64, a feature row that does not fit LDS): the caller runs the five-launch path.
// `sync`: head_fused_sync_ints() zero-initialised ints owned by the handle.  `ws`: head_fused_ws_bytes(B, C) bytes.  max_grid > 0 caps the grid
// (tests: a grid smaller than the tile count makes workgroups take several tiles and exercises the recompute queue).
int launch_head_fused(const half_t* feats, int B, int D, float scale, const half_t* const* clf, int n_mod, int C, const float* w,
                      float* out, half_t* raw_out, void* ws, int* sync, int n_cu, int max_grid, hipStream_t s) {
    if (B <= 0) return 0;
    if (D % 64 || D > 4096 || n_mod < 1 || n_mod > 3 || C < 1) return -100;
    const int Tc = (C + HF_BN - 1) / HF_BN;
    const bool raw = raw_out != nullptr;
    // 32-row tiles while there is at most one per CU (more, smaller workgroups: C = 1000, B = 256 -> 64 of them), else 64-row tiles
    int BM = ((long)((B + 31) / 32) * Tc <= (long)n_cu) ? 32 : 64;
    if (head_lds_bytes(BM, D) > 160 * 1024) BM = 32;
    if (head_lds_bytes(BM, D) > 160 * 1024) return -100;
    const size_t lds = head_lds_bytes(BM, D);
    const size_t rows = (size_t)(B + 63) / 64 * 64;
    float* partial = (float*)ws;
    float* merged = partial + 3 * rows * Tc * 2;
    int* leftover = (int*)(merged + 3 * rows * 2);
    if (raw) return BM == 32 ? launch_one<32, true>(feats, B, D, scale, clf, 1, C, nullptr, nullptr, raw_out, nullptr, nullptr, nullptr, nullptr, Tc, lds, 0, s)
                             : launch_one<64, true>(feats, B, D, scale, clf, 1, C, nullptr, nullptr, raw_out, nullptr, nullptr, nullptr, nullptr, Tc, lds, 0, s);
    return BM == 32 ? launch_one<32, false>(feats, B, D, scale, clf, n_mod, C, w, out, nullptr, partial, merged, leftover, sync, Tc, lds, max_grid, s)
                    : launch_one<64, false>(feats, B, D, scale, clf, n_mod, C, w, out, nullptr, partial, merged, leftover, sync, Tc, lds, max_grid, s);
}

#ifdef OVMR_EXPERIMENTS
extern "C" int ovmr_debug_head_stamps(long long* host_out) {
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_head_stamps), 16 * sizeof(long long));
}
#endif
