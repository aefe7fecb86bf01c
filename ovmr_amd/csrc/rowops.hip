// HBM-bound data-movement kernels either side of the GEMMs: patch extraction, CLS row, L2
// normalisation, text embedding / positional add, EOS gather, aggregator input/output and
// prompt assembly.  All are coalesced row kernels (one wave or one thread-group per row).
#include "common.h"

namespace {

// ---- K1 front: im2col for conv1 (kernel = stride = P, no bias; clip/model.py:366,412-414) -------
// out[(b*G*G + gy*G + gx)][k], k = c*P*P + ky*P + kx (the conv weight's own [3,P,P] order), zero
// for k >= 3*P*P.  One thread per 8 consecutive k (16-byte store).
template <typename TI>
__global__ void im2col_kernel(const TI* __restrict__ img, half_t* __restrict__ out, int B, int R, int P,
                              int G, int K, int Kpad) {
    const long chunks_per_row = Kpad >> 3;
    const long total = (long)B * G * G * chunks_per_row;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const long row = i / chunks_per_row;
        const int k0 = (int)(i - row * chunks_per_row) << 3;
        const int b = (int)(row / (G * G));
        const int p = (int)(row - (long)b * G * G);
        const int gy = p / G, gx = p - gy * G;
        half8_t v;
        if ((P & 7) == 0 && k0 + 8 <= K) {
            const int c = k0 / (P * P), rem = k0 - c * P * P, ky = rem / P, kx = rem - ky * P;
            const TI* src = img + (((long)b * 3 + c) * R + gy * P + ky) * R + gx * P + kx;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = (half_t)src[j];
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int k = k0 + j;
                if (k < K) {
                    const int c = k / (P * P), rem = k - c * P * P, ky = rem / P, kx = rem - ky * P;
                    v[j] = (half_t)img[(((long)b * 3 + c) * R + gy * P + ky) * R + gx * P + kx];
                } else {
                    v[j] = (half_t)0.f;
                }
            }
        }
        *(half8_t*)(out + row * Kpad + k0) = v;
    }
}

// x[b*L + 0][:] = h(cls16 + pos16[0])   (clip/model.py:415-416), precomputed in cls_pos
__global__ void fill_cls_kernel(half_t* __restrict__ x, const half_t* __restrict__ cls_pos, int B, int L, int W) {
    const long total = (long)B * (W >> 2);
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int b = (int)(i / (W >> 2)), c = (int)(i - (long)b * (W >> 2)) << 2;
        *(half4_t*)(x + (long)b * L * W + c) = *(const half4_t*)(cls_pos + c);
    }
}

// x / x.norm(dim=-1, keepdim=True) on an fp16 tensor: the norm is accumulated in fp32, rounded to
// fp16, and the quotient is rounded to fp16 (trainers/mm_classifier_one_prompt.py:204,244,307).
__global__ __launch_bounds__(256) void l2norm_f16_kernel(half_t* __restrict__ x, int rows, int D) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    half_t* xr = x + (long)row * D;
    float ss = 0.f;
    for (int c = lane * 4; c < D; c += 256) {
        half4_t h = *(const half4_t*)(xr + c);
#pragma unroll
        for (int k = 0; k < 4; ++k) ss += (float)h[k] * (float)h[k];
    }
    const float n = fmaxf((float)(half_t)sqrtf(wave_sum(ss)), 1e-12f);
    for (int c = lane * 4; c < D; c += 256) {
        half4_t h = *(const half4_t*)(xr + c);
#pragma unroll
        for (int k = 0; k < 4; ++k) h[k] = (half_t)((float)h[k] / n);
        *(half4_t*)(xr + c) = h;
    }
}

template <typename TS, typename TD>
__global__ void cast_kernel(const TS* __restrict__ s, TD* __restrict__ d, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) d[i] = (TD)s[i];
}

template <typename TS>
__global__ void transpose_to_f16_kernel(const TS* __restrict__ s, half_t* __restrict__ d, int rows, int cols) {
    const long n = (long)rows * cols;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int c = (int)(i / rows), r = (int)(i - (long)c * rows);   // d[c][r] = s[r][c]
        d[i] = (half_t)s[(long)r * cols + c];
    }
}

__global__ void pad_rows_kernel(const half_t* __restrict__ s, half_t* __restrict__ d, int rows, int cols, int cols_pad) {
    const long n = (long)rows * cols_pad;
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
        const int r = (int)(i / cols_pad), c = (int)(i - (long)r * cols_pad);
        d[i] = c < cols ? s[(long)r * cols + c] : (half_t)0.f;
    }
}

__global__ void add_f16_kernel(const half_t* a, const half_t* b, half_t* o, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        o[i] = (half_t)((float)a[i] + (float)b[i]);
}

__global__ void scale_f16_kernel(const half_t* x, half_t* y, float scale, long n) {
    for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
        y[i] = (half_t)(scale * (float)x[i]);
}

// ---- text front (clip/model.py:821-823): x = h(h(token_embedding[ids]) + h(pos)), index = argmax(ids)
__global__ __launch_bounds__(256) void text_embed_ids_kernel(const int64_t* __restrict__ ids, int ids_stride,
                                                             const float* __restrict__ tok, const half_t* __restrict__ pos16,
                                                             half_t* __restrict__ x, int N, int Lseq, int D) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)N * Lseq) return;
    const int n = (int)(row / Lseq), l = (int)(row - (long)n * Lseq);
    const float* e = tok + ids[(long)n * ids_stride + l] * (long)D;
    for (int c = lane * 4; c < D; c += 256) {
        float4_t f = *(const float4_t*)(e + c);
        half4_t p = *(const half4_t*)(pos16 + (long)l * D + c);
        half4_t o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = (half_t)((float)(half_t)f[k] + (float)p[k]);
        *(half4_t*)(x + row * D + c) = o;
    }
}

// token_embedding(ids).type(fp16) (trainers/mm_classifier_one_prompt.py:129-130)
__global__ __launch_bounds__(256) void embed_gather_kernel(const int64_t* __restrict__ ids, const float* __restrict__ tok,
                                                           half_t* __restrict__ out, long rows, int D) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const float* e = tok + ids[row] * (long)D;
    for (int c = lane * 4; c < D; c += 256) {
        float4_t f = *(const float4_t*)(e + c);
        *(half4_t*)(out + row * D + c) = (half4_t){(half_t)f[0], (half_t)f[1], (half_t)f[2], (half_t)f[3]};
    }
}

__global__ void argmax_ids_kernel(const int64_t* __restrict__ ids, int ids_stride, int Lctx, int* __restrict__ index, int N) {
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= N) return;
    int64_t best = ids[(long)n * ids_stride];
    int bi = 0;
    for (int l = 1; l < Lctx; ++l) {
        const int64_t v = ids[(long)n * ids_stride + l];
        if (v > best) { best = v; bi = l; }
    }
    index[n] = bi;
}

// TextEncoder.forward front (trainers/mm_classifier_one_prompt.py:81): x = prompts.half + pos.half[:L]
__global__ __launch_bounds__(256) void text_add_pos_kernel(const half_t* __restrict__ prompts, int Lctx,
                                                           const half_t* __restrict__ pos16, half_t* __restrict__ x,
                                                           int N, int Lseq, int D) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)N * Lseq) return;
    const int n = (int)(row / Lseq), l = (int)(row - (long)n * Lseq);
    const half_t* pr = prompts + ((long)n * Lctx + l) * D;
    for (int c = lane * 4; c < D; c += 256) {
        half4_t a = *(const half4_t*)(pr + c);
        half4_t p = *(const half4_t*)(pos16 + (long)l * D + c);
        half4_t o;
#pragma unroll
        for (int k = 0; k < 4; ++k) o[k] = (half_t)((float)a[k] + (float)p[k]);
        *(half4_t*)(x + row * D + c) = o;
    }
}

// x[arange(N), index] (trainers/mm_classifier_one_prompt.py:89): LayerNorm is per row, so the rows
// are gathered first and ln_final runs on N rows only.
__global__ __launch_bounds__(256) void gather_rows_kernel(const half_t* __restrict__ x, const int* __restrict__ index,
                                                          half_t* __restrict__ out, int N, int Lseq, int D) {
    const int lane = threadIdx.x & 63;
    const int n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    const int l = min(max(index[n], 0), Lseq - 1);
    const half_t* s = x + ((long)n * Lseq + l) * D;
    for (int c = lane * 4; c < D; c += 256) *(half4_t*)(out + (long)n * D + c) = *(const half4_t*)(s + c);
}

// The sharded job's ONE packed all-gather (ovmr_amd/shard.py pack_block): a rank's block is [bound, K + 2] fp16 with
// K = 3 D + n_ctx D -- row i = mm | vision | text classifier rows and the visual tokens of class labels[i], the class's int32 label as two
// fp16 bit columns; rows i >= n are zero with label -1.  Rows are 4-byte aligned (K + 2 is even), not 16: 4-byte moves.
__global__ __launch_bounds__(256) void pack_rows_kernel(const half_t* __restrict__ mm, const half_t* __restrict__ v, const half_t* __restrict__ t,
                                                        const half_t* __restrict__ tokens, const int64_t* __restrict__ labels, int n, int D,
                                                        int n_ctx, half_t* __restrict__ block) {
    const int i = blockIdx.x, K = (3 + n_ctx) * D;
    unsigned* dst = (unsigned*)(block + (long)i * (K + 2));
    const long c = i < n ? labels[i] : -1;
    const int D2 = D / 2, K2 = K / 2;
    for (int k = threadIdx.x; k < K2; k += 256) {
        unsigned val = 0;
        if (i < n) {
            const int part = k / D2, o = k - part * D2;
            const half_t* src = part == 0 ? mm + c * D : part == 1 ? v + c * D : part == 2 ? t + c * D : tokens + c * (long)n_ctx * D + (long)(part - 3) * D;
            val = ((const unsigned*)src)[o];
        }
        dst[k] = val;
    }
    if (threadIdx.x == 0) dst[K2] = (unsigned)(int)c;
}

// ... and its inverse on the gathered [rows, K + 2] blocks of all ranks: every row with a label in [0, C) lands in the four class-indexed
// arrays, seen[label] counts it (a class must arrive exactly once); labels outside [-1, C) are counted in seen[C].
__global__ __launch_bounds__(256) void unpack_rows_kernel(const half_t* __restrict__ gathered, int C, int D, int n_ctx, half_t* __restrict__ mm,
                                                          half_t* __restrict__ v, half_t* __restrict__ t, half_t* __restrict__ tokens,
                                                          int* __restrict__ seen) {
    const int K = (3 + n_ctx) * D, D2 = D / 2, K2 = K / 2;
    const unsigned* src = (const unsigned*)(gathered + (long)blockIdx.x * (K + 2));
    const int c = (int)src[K2];
    if (c < 0 || c >= C) {
        if (threadIdx.x == 0 && c != -1) atomicAdd(seen + C, 1);
        return;
    }
    if (threadIdx.x == 0) atomicAdd(seen + c, 1);
    for (int k = threadIdx.x; k < K2; k += 256) {
        const int part = k / D2, o = k - part * D2;
        half_t* dst = part == 0 ? mm + (long)c * D : part == 1 ? v + (long)c * D : part == 2 ? t + (long)c * D
                                                                                 : tokens + (long)c * n_ctx * D + (long)(part - 3) * D;
        ((unsigned*)dst)[o] = src[k];
    }
}

// PromptLearner.forward (trainers/mm_classifier_one_prompt.py:167-168): cat([cls_token, feats]) in fp32
__global__ __launch_bounds__(256) void agg_input_kernel(const float* __restrict__ cls, const half_t* __restrict__ feats,
                                                        float* __restrict__ x, int Cb, int S, int n_ctx, int D) {
    const int lane = threadIdx.x & 63;
    const int La = n_ctx + S;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)Cb * La) return;
    const int c = (int)(row / La), t = (int)(row - (long)c * La);
    for (int k = lane * 4; k < D; k += 256) {
        float4_t v;
        if (t < n_ctx) {
            v = *(const float4_t*)(cls + (long)t * D + k);
        } else {
            half4_t h = *(const half4_t*)(feats + ((long)c * S + (t - n_ctx)) * D + k);
            v = (float4_t){(float)h[0], (float)h[1], (float)h[2], (float)h[3]};
        }
        *(float4_t*)(x + row * D + k) = v;
    }
}

// aggregator(...)[0:n_ctx] (:169) -> tokens [Cb, n_ctx, D] fp32
__global__ __launch_bounds__(256) void agg_output_kernel(const float* __restrict__ x, float* __restrict__ tokens,
                                                         int Cb, int La, int n_ctx, int D) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)Cb * n_ctx) return;
    const int c = (int)(row / n_ctx), t = (int)(row - (long)c * n_ctx);
    for (int k = lane * 4; k < D; k += 256)
        *(float4_t*)(tokens + row * D + k) = *(const float4_t*)(x + ((long)c * La + t) * D + k);
}

// PromptLearner.update_prompts (:156-157): cat([P[:, :2], tokens.half, P[:, 2:-n_ctx]], dim=1)
__global__ __launch_bounds__(256) void assemble_prompts_kernel(const half_t* __restrict__ base, const int64_t* __restrict__ labels,
                                                               const float* __restrict__ tokens, half_t* __restrict__ out,
                                                               int Cb, int Lctx, int n_ctx, int D) {
    const int lane = threadIdx.x & 63;
    const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= (long)Cb * Lctx) return;
    const int c = (int)(row / Lctx), l = (int)(row - (long)c * Lctx);
    const long src_class = labels ? labels[c] : 0;
    half_t* o = out + row * D;
    if (l >= 2 && l < 2 + n_ctx) {
        const float* t = tokens + ((long)c * n_ctx + (l - 2)) * D;
        for (int k = lane * 4; k < D; k += 256) {
            float4_t f = *(const float4_t*)(t + k);
            *(half4_t*)(o + k) = (half4_t){(half_t)f[0], (half_t)f[1], (half_t)f[2], (half_t)f[3]};
        }
    } else {
        const int sl = l < 2 ? l : l - n_ctx;
        const half_t* s = base + (src_class * Lctx + sl) * D;
        for (int k = lane * 4; k < D; k += 256) *(half4_t*)(o + k) = *(const half4_t*)(s + k);
    }
}

inline int grid1d(long n, int block = 256) { return (int)((n + block - 1) / block < 1 ? 1 : ((n + block - 1) / block > 65535L * 16 ? 65535L * 16 : (n + block - 1) / block)); }

}  // namespace

int launch_im2col(const void* img, int img_is_f32, half_t* out, int B, int R, int P, int Kpad, hipStream_t s) {
    const int G = R / P, K = 3 * P * P;
    const long n = (long)B * G * G * (Kpad >> 3);
    if (n <= 0) return 0;
    if (img_is_f32) hipLaunchKernelGGL(im2col_kernel<float>, dim3(grid1d(n)), dim3(256), 0, s, (const float*)img, out, B, R, P, G, K, Kpad);
    else hipLaunchKernelGGL(im2col_kernel<half_t>, dim3(grid1d(n)), dim3(256), 0, s, (const half_t*)img, out, B, R, P, G, K, Kpad);
    return (int)hipGetLastError();
}
int launch_fill_cls(half_t* x, const half_t* cls_pos, int B, int L, int W, hipStream_t s) {
    hipLaunchKernelGGL(fill_cls_kernel, dim3(grid1d((long)B * (W >> 2))), dim3(256), 0, s, x, cls_pos, B, L, W);
    return (int)hipGetLastError();
}
int launch_l2norm_f16(half_t* x, int rows, int D, hipStream_t s) {
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(l2norm_f16_kernel, dim3((rows + 3) / 4), dim3(256), 0, s, x, rows, D);
    return (int)hipGetLastError();
}
int launch_cast(const void* src, int src_f32, void* dst, int dst_f32, long n, hipStream_t s) {
    if (n <= 0) return 0;
    const dim3 g(grid1d(n)), b(256);
    if (src_f32 && dst_f32) hipLaunchKernelGGL((cast_kernel<float, float>), g, b, 0, s, (const float*)src, (float*)dst, n);
    else if (src_f32 && !dst_f32) hipLaunchKernelGGL((cast_kernel<float, half_t>), g, b, 0, s, (const float*)src, (half_t*)dst, n);
    else if (!src_f32 && dst_f32) hipLaunchKernelGGL((cast_kernel<half_t, float>), g, b, 0, s, (const half_t*)src, (float*)dst, n);
    else hipLaunchKernelGGL((cast_kernel<half_t, half_t>), g, b, 0, s, (const half_t*)src, (half_t*)dst, n);
    return (int)hipGetLastError();
}
int launch_transpose_to_f16(const void* src, int src_f32, half_t* dst, int rows, int cols, hipStream_t s) {
    const long n = (long)rows * cols;
    if (src_f32) hipLaunchKernelGGL(transpose_to_f16_kernel<float>, dim3(grid1d(n)), dim3(256), 0, s, (const float*)src, dst, rows, cols);
    else hipLaunchKernelGGL(transpose_to_f16_kernel<half_t>, dim3(grid1d(n)), dim3(256), 0, s, (const half_t*)src, dst, rows, cols);
    return (int)hipGetLastError();
}
int launch_pad_rows_f16(const half_t* src, half_t* dst, int rows, int cols, int cols_pad, hipStream_t s) {
    hipLaunchKernelGGL(pad_rows_kernel, dim3(grid1d((long)rows * cols_pad)), dim3(256), 0, s, src, dst, rows, cols, cols_pad);
    return (int)hipGetLastError();
}
int launch_add_f16(const half_t* a, const half_t* b, half_t* out, long n, hipStream_t s) {
    hipLaunchKernelGGL(add_f16_kernel, dim3(grid1d(n)), dim3(256), 0, s, a, b, out, n);
    return (int)hipGetLastError();
}
// ToTensor + Normalize of the test transform (Dassl.pytorch/dassl/data/transforms/transforms.py:495-526) + the fp16 cast of
// trainers/mm_classifier_one_prompt.py:243 on the device: uint8 [B, R, R, 3] (what decode + bicubic resize + centre crop leave on
// the host) -> fp16 [B, 3, R, R], x = ((u / 255) - mean[c]) / std[c] in fp32 with IEEE division, i.e. bit for bit the fp32 tensor
// torchvision builds, then rounded to fp16.  One thread = 8 consecutive pixels of a row: 24 B in, 3 x 16 B out (one per plane).
__global__ void preprocess_u8_kernel(const uint8_t* __restrict__ in, half_t* __restrict__ out, long groups, int R,
                                     float m0, float m1, float m2, float s0, float s1, float s2) {
    const long g = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= groups) return;
    const int gpr = R / 8;                               // groups per row
    const long row = g / gpr;                            // b * R + y
    const int x0 = (int)(g - row * gpr) * 8;
    const long b = row / R;
    const int y = (int)(row - b * R);
    const uint8_t* src = in + (row * R + x0) * 3;
    uint32_t w[6];
#pragma unroll
    for (int k = 0; k < 6; ++k) w[k] = ((const uint32_t*)src)[k];      // rows are R * 3 bytes, x0 * 3 = 24 * n: 4-byte aligned when R % 4 == 0
    const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
    half8_t o[3];
#pragma unroll
    for (int p = 0; p < 8; ++p)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int byte = p * 3 + c;
            const float u = (float)((w[byte >> 2] >> ((byte & 3) * 8)) & 0xff);
            o[c][p] = (half_t)((u / 255.0f - mean[c]) / sd[c]);
        }
#pragma unroll
    for (int c = 0; c < 3; ++c) *(half8_t*)(out + ((b * 3 + c) * R + y) * (long)R + x0) = o[c];
}

int launch_preprocess_u8(const uint8_t* in, half_t* out, int B, int R, const float* mean3, const float* std3, hipStream_t s) {
    if (B <= 0) return 0;
    if (R % 8) return -2;
    const long groups = (long)B * R * (R / 8);
    hipLaunchKernelGGL(preprocess_u8_kernel, dim3((unsigned)((groups + 255) / 256)), dim3(256), 0, s, in, out, groups, R,
                       mean3[0], mean3[1], mean3[2], std3[0], std3[1], std3[2]);
    return (int)hipGetLastError();
}

int launch_scale_f16(const half_t* x, half_t* y, float scale, long n, hipStream_t s) {
    if (n <= 0) return 0;
    hipLaunchKernelGGL(scale_f16_kernel, dim3(grid1d(n)), dim3(256), 0, s, x, y, scale, n);
    return (int)hipGetLastError();
}
int launch_text_embed_ids(const int64_t* ids, int ids_stride, const float* tok_emb, const half_t* pos16,
                          half_t* x, int* index, int N, int Lctx, int Lseq, int D, hipStream_t s) {
    if (N <= 0) return 0;
    const long rows = (long)N * Lseq;
    hipLaunchKernelGGL(text_embed_ids_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, ids, ids_stride, tok_emb, pos16, x, N, Lseq, D);
    hipLaunchKernelGGL(argmax_ids_kernel, dim3((N + 255) / 256), dim3(256), 0, s, ids, ids_stride, Lctx, index, N);
    return (int)hipGetLastError();
}
int launch_embed_gather(const int64_t* ids, const float* tok_emb, half_t* out, long rows, int D, hipStream_t s) {
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(embed_gather_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, ids, tok_emb, out, rows, D);
    return (int)hipGetLastError();
}
int launch_text_add_pos(const half_t* prompts, int Lctx, const half_t* pos16, half_t* x, int N, int Lseq, int D, hipStream_t s) {
    if (N <= 0) return 0;
    const long rows = (long)N * Lseq;
    hipLaunchKernelGGL(text_add_pos_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, prompts, Lctx, pos16, x, N, Lseq, D);
    return (int)hipGetLastError();
}
int launch_gather_rows_f16(const half_t* x, const int* index, half_t* out, int N, int Lseq, int D, hipStream_t s) {
    if (N <= 0) return 0;
    hipLaunchKernelGGL(gather_rows_kernel, dim3((N + 3) / 4), dim3(256), 0, s, x, index, out, N, Lseq, D);
    return (int)hipGetLastError();
}
int launch_pack_rows(const half_t* mm, const half_t* v, const half_t* t, const half_t* tokens, const int64_t* labels, int n, int D, int n_ctx,
                     int bound, half_t* block, hipStream_t s) {
    if (bound <= 0) return 0;
    hipLaunchKernelGGL(pack_rows_kernel, dim3(bound), dim3(256), 0, s, mm, v, t, tokens, labels, n, D, n_ctx, block);
    return (int)hipGetLastError();
}
int launch_unpack_rows(const half_t* gathered, int rows, int C, int D, int n_ctx, half_t* mm, half_t* v, half_t* t, half_t* tokens, int* seen,
                       hipStream_t s) {
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(unpack_rows_kernel, dim3(rows), dim3(256), 0, s, gathered, C, D, n_ctx, mm, v, t, tokens, seen);
    return (int)hipGetLastError();
}
int launch_agg_input(const float* cls_token, const half_t* feats, float* x, int Cb, int S, int n_ctx, int D, hipStream_t s) {
    const long rows = (long)Cb * (S + n_ctx);
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(agg_input_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, cls_token, feats, x, Cb, S, n_ctx, D);
    return (int)hipGetLastError();
}
int launch_agg_output(const float* x, float* tokens, int Cb, int La, int n_ctx, int D, hipStream_t s) {
    const long rows = (long)Cb * n_ctx;
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(agg_output_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, x, tokens, Cb, La, n_ctx, D);
    return (int)hipGetLastError();
}
int launch_assemble_prompts(const half_t* base, const int64_t* labels, const float* tokens, half_t* out,
                            int Cb, int Lctx, int n_ctx, int D, hipStream_t s) {
    const long rows = (long)Cb * Lctx;
    if (rows <= 0) return 0;
    hipLaunchKernelGGL(assemble_prompts_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, base, labels, tokens, out, Cb, Lctx, n_ctx, D);
    return (int)hipGetLastError();
}
