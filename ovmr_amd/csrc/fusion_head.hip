// OVMR classifier-fusion head (trainers/mm_classifier_one_prompt.py:261-274, 348-363).
//   * xval_argmax_counts: K19 -- per exemplar row argmax over the class logits, then the per-class
//     true-positive / prediction histograms that torcheval's multiclass_f1_score(average=None)
//     builds (torcheval==0.0.7, requirements.txt:16).  Ties resolve to the lowest class index
//     (torch.argmax on CPU).
//   * fusion_weights: K20 -- f1 = 2pr/(p+r), NaN -> 0, softmax(tau * [f1_mm, f1_v, f1_t]).
//   * fused_softmax: K21/K22 -- per query row, softmax over classes of each modality's fp16
//     logits in fp32, then sum_n p[b,c,n] * w[c,n] (column order mm, vision, text).
// All three are HBM/L2-bound row kernels; the logits themselves come from the MFMA GEMM.
#include "common.h"

namespace {

// counts[key] += 1 for every lane with valid set, ONE atomic per distinct key and wave: the argmax of exemplar rows concentrates on few
// classes (always with untrained weights, per class with trained ones: S consecutive rows share their label), and same-address atomics
// serialise -- 16 000 rows on a handful of classes took 155 us (r04o trace), most of the cross-validation step.
__device__ __forceinline__ void wave_histogram_add(int* counts, int key, bool valid) {
    unsigned long long todo = __builtin_amdgcn_ballot_w64(valid);
    const int lane = threadIdx.x & 63;
    while (todo) {
        const int leader = __builtin_ctzll(todo);
        const int k = __builtin_amdgcn_readlane(key, leader);
        const unsigned long long same = __builtin_amdgcn_ballot_w64(valid && key == k);
        if (lane == leader) atomicAdd(counts + k, (int)__builtin_popcountll(same));
        todo &= ~same;
    }
}

__global__ __launch_bounds__(256) void xval_argmax_counts(const half_t* __restrict__ logits, int ld,
                                                          const int* __restrict__ labels, int rows, int C,
                                                          int* __restrict__ tp, int* __restrict__ n_pred) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const half_t* lr = logits + (long)row * ld;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = lane; c < C; c += 64) {
        const float v = (float)lr[c];
        if (v > best || bi == 0x7fffffff) { best = v; bi = c; }   // strict >: first index wins inside a lane
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        if (ov > best || (ov == best && oi < bi)) { best = ov; bi = oi; }
    }
    if (lane == 0 && bi < C) {
        atomicAdd(n_pred + bi, 1);
        if (bi == labels[row]) atomicAdd(tp + bi, 1);
    }
}

// Second stage of the fused logits GEMM + row argmax (EPI_SCALE_ARGMAX, gemm_f16_v5.hip): partial[row][tile] = {maximum of
// the tile's columns, lowest column holding it}; the row's argmax is the largest value, ties to the lowest column -- the same
// answer as one argmax over the whole fp16 logits row -- followed by the same two histogram updates.
__global__ __launch_bounds__(256) void xval_argmax_reduce(const float* __restrict__ partial, int tiles,
                                                          const int* __restrict__ labels, int rows, int C,
                                                          int* __restrict__ tp, int* __restrict__ n_pred) {
    const int row = blockIdx.x * blockDim.x + threadIdx.x;
    const bool live = row < rows;                      // (no early return: the wave-wide histogram below needs every lane)
    // pairs are read as two 32-bit integers (value bits, column): hipcc (ROCm 7.2) mis-selected the VALUE register for the
    // column when the pair was loaded as a float2 and its second lane bit-cast to int
    const int* p = (const int*)partial + (long)(live ? row : 0) * tiles * 2;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int t = 0; t < tiles; ++t) {                 // tiles in increasing column order: strict > keeps the lowest column
        const float v = __int_as_float(p[2 * t]);
        const int c = p[2 * t + 1];
        if (v > best || bi == 0x7fffffff) { best = v; bi = c; }
    }
    const bool ok = live && bi < C;
    wave_histogram_add(n_pred, bi, ok);
    wave_histogram_add(tp, bi, ok && bi == labels[live ? row : 0]);
}

// Classification.process of the test loop (Dassl.pytorch/dassl/evaluation/evaluator.py:50-67): pred = mo.max(1)[1] (lowest column on ties;
// a NaN counts as the largest value, the first one wins -- torch's rule), then the three histograms every figure of evaluate() is
// made of (:69-138: accuracy = sum tp / total, per-class accuracy = tp / n_label, F1 from tp, n_pred, n_label).  One wave per output row
// (16-byte loads where the row is aligned), the block's rows are counted by its first wave with one atomic per distinct class.
// counts = int32 [3][C] (tp, n_pred, n_label) + [1]: rows whose label lies outside [0, C) (counted nowhere else).
template <typename T>
__device__ __forceinline__ void eval_row_argmax(const T* __restrict__ row, int C, int lane, float& best, int& bi) {
    best = -INFINITY;
    bi = 0x7fffffff;
    auto take = [&](float v, int c) {
        // strict >: the first (lowest) column of a value wins inside a lane; NaN beats everything that is not NaN
        if (bi == 0x7fffffff || v > best || (v != v && best == best)) { best = v; bi = c; }
    };
    constexpr int V = 16 / (int)sizeof(T);
    if ((((uintptr_t)row) & 15) == 0) {
        const int Cv = C / V * V;
        for (int c = lane * V; c < Cv; c += 64 * V) {
            if constexpr (sizeof(T) == 4) {
                const float4 q = *(const float4*)(row + c);
                take(q.x, c); take(q.y, c + 1); take(q.z, c + 2); take(q.w, c + 3);
            } else {
                const uint4 q = *(const uint4*)(row + c);
                const unsigned u[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    half2_t h2 = *(const half2_t*)&u[i];
                    take((float)h2[0], c + 2 * i); take((float)h2[1], c + 2 * i + 1);
                }
            }
        }
        for (int c = Cv + lane; c < C; c += 64) take((float)row[c], c);
    } else
        for (int c = lane; c < C; c += 64) take((float)row[c], c);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float ov = __shfl_xor(best, o, 64);
        const int oi = __shfl_xor(bi, o, 64);
        const bool on = ov != ov, bn = best != best;
        const bool better = oi != 0x7fffffff && (bi == 0x7fffffff || (on && !bn) || (!bn && ov > best) || ((ov == best || (on && bn)) && oi < bi));
        if (better) { best = ov; bi = oi; }
    }
}

template <typename T>
__global__ __launch_bounds__(256) void eval_counts_kernel(const T* __restrict__ out, long ld, const int64_t* __restrict__ labels,
                                                          int rows, int C, int* __restrict__ counts) {
    __shared__ int pred_s[4], gt_s[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    float best;
    int bi = 0x7fffffff;
    if (row < rows) eval_row_argmax(out + (long)row * ld, C, lane, best, bi);
    if (lane == 0) {
        pred_s[wave] = row < rows ? bi : -1;
        const int64_t g = row < rows ? labels[row] : -1;
        gt_s[wave] = row < rows ? ((g >= 0 && g < C) ? (int)g : -2) : -1;
    }
    __syncthreads();
    if (wave == 0) {
        const int p = lane < 4 ? pred_s[lane] : -1, g = lane < 4 ? gt_s[lane] : -1;
        const bool live = g >= 0 && p >= 0 && p < C;
        wave_histogram_add(counts + 2 * C, g, live);                  // n_label
        wave_histogram_add(counts + C, p, live);                      // n_pred
        wave_histogram_add(counts, p, live && p == g);                // tp
        wave_histogram_add(counts + 3 * C, 0, g == -2);               // labels outside [0, C)
    }
}

// counts: int32 [3][2][C] = {mm, vision, text} x {tp, n_pred}; n_label: int32 [C]
__global__ void fusion_weights_kernel(const int* __restrict__ counts, const int* __restrict__ n_label, int C,
                                      float tau, float* __restrict__ out) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    float f1[3];
#pragma unroll
    for (int m = 0; m < 3; ++m) {
        const float tp = (float)counts[(m * 2 + 0) * C + c];
        const float np = (float)counts[(m * 2 + 1) * C + c];
        const float precision = tp / np;                 // 0/0 -> NaN like torch
        const float recall = tp / (float)n_label[c];
        float f = 2.f * precision * recall / (precision + recall);
        if (f != f) f = 0.f;                             // torch.nan_to_num
        f1[m] = tau * f;
    }
    const float mx = fmaxf(f1[0], fmaxf(f1[1], f1[2]));
    const float e0 = __expf(f1[0] - mx), e1 = __expf(f1[1] - mx), e2 = __expf(f1[2] - mx);
    const float inv = 1.f / (e0 + e1 + e2);
    out[c * 3 + 0] = e0 * inv;
    out[c * 3 + 1] = e1 * inv;
    out[c * 3 + 2] = e2 * inv;
}

__device__ __forceinline__ float block_reduce(float v, bool is_max, float* red) {
    v = is_max ? wave_max(v) : wave_sum(v);
    const int wave = threadIdx.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[wave] = v;
    __syncthreads();
    float r = red[0];
    for (int i = 1; i < (int)(blockDim.x >> 6); ++i) r = is_max ? fmaxf(r, red[i]) : r + red[i];
    return r;
}

// out[b][c] = sum_m softmax_c(float(l_m[b]))[c] * w[c][m]   (w == nullptr: single modality, weight 1)
__global__ __launch_bounds__(256) void fused_softmax_kernel(const half_t* __restrict__ l0, const half_t* __restrict__ l1,
                                                            const half_t* __restrict__ l2, const float* __restrict__ w,
                                                            int n_mod, float* __restrict__ out, int C) {
    __shared__ float red[4];
    const int b = blockIdx.x;
    const half_t* ls[3] = {l0 + (long)b * C, l1 ? l1 + (long)b * C : nullptr, l2 ? l2 + (long)b * C : nullptr};
    float mx[3], inv[3];
    for (int m = 0; m < n_mod; ++m) {
        float v = -INFINITY;
        for (int c = threadIdx.x; c < C; c += blockDim.x) v = fmaxf(v, (float)ls[m][c]);
        mx[m] = block_reduce(v, true, red);
        float sum = 0.f;
        for (int c = threadIdx.x; c < C; c += blockDim.x) sum += __expf((float)ls[m][c] - mx[m]);
        inv[m] = 1.f / block_reduce(sum, false, red);
    }
    for (int c = threadIdx.x; c < C; c += blockDim.x) {
        float acc = 0.f;
        for (int m = 0; m < n_mod; ++m) {
            const float p = __expf((float)ls[m][c] - mx[m]) * inv[m];
            acc += w ? p * w[c * 3 + m] : p;
        }
        out[(long)b * C + c] = acc;
    }
}

}  // namespace

int launch_argmax_counts(const half_t* logits, int ld, const int* labels, int R, int C, int* tp, int* n_pred, hipStream_t s) {
    if (R <= 0) return 0;
    hipLaunchKernelGGL(xval_argmax_counts, dim3((R + 3) / 4), dim3(256), 0, s, logits, ld, labels, R, C, tp, n_pred);
    return (int)hipGetLastError();
}

int launch_argmax_reduce(const float* partial, int tiles, const int* labels, int R, int C, int* tp, int* n_pred, hipStream_t s) {
    if (R <= 0) return 0;
    hipLaunchKernelGGL(xval_argmax_reduce, dim3((R + 255) / 256), dim3(256), 0, s, partial, tiles, labels, R, C, tp, n_pred);
    return (int)hipGetLastError();
}

int launch_eval_counts(const void* out, int out_is_f32, long ld, const int64_t* labels, int B, int C, int* counts, hipStream_t s) {
    if (B <= 0) return 0;
    if (out_is_f32)
        hipLaunchKernelGGL(eval_counts_kernel<float>, dim3((B + 3) / 4), dim3(256), 0, s, (const float*)out, ld, labels, B, C, counts);
    else
        hipLaunchKernelGGL(eval_counts_kernel<half_t>, dim3((B + 3) / 4), dim3(256), 0, s, (const half_t*)out, ld, labels, B, C, counts);
    return (int)hipGetLastError();
}

int launch_fusion_weights(const int* counts, const int* n_label, int C, float tau, float* out, hipStream_t s) {
    if (C <= 0) return 0;
    hipLaunchKernelGGL(fusion_weights_kernel, dim3((C + 255) / 256), dim3(256), 0, s, counts, n_label, C, tau, out);
    return (int)hipGetLastError();
}

int launch_fused_softmax(const half_t* l0, const half_t* l1, const half_t* l2, const float* w, int n_mod,
                         float* out, int B, int C, hipStream_t s) {
    if (B <= 0) return 0;
    hipLaunchKernelGGL(fused_softmax_kernel, dim3(B), dim3(256), 0, s, l0, l1, l2, w, n_mod, out, C);
    return (int)hipGetLastError();
}
