// Shared device/host helpers for the OVMR gfx950 kernels (wave64, MFMA 16x16x32 f16).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float2_t __attribute__((ext_vector_type(2)));

#define OVMR_WAVE 64
#define OVMR_MAX_DEVICES 64

// GEMM epilogues (C = epi(A * W^T)), rounding points follow the reference's fp16 CPU path:
// every nn.Linear output is rounded to fp16 before the next elementwise op.
enum {
    EPI_NONE = 0,        // C = h(acc)
    EPI_BIAS = 1,        // C = h(acc + bias)                        nn.Linear            (clip/model.py:171,173-177)
    EPI_BIAS_QGELU = 2,  // u = h(acc+bias); C = u * sigmoid(1.702u) c_fc + QuickGELU     (clip/model.py:162-164)
    EPI_BIAS_RES = 3,    // C = h(h(acc+bias) + res)                 x + attn / x + mlp   (clip/model.py:192-193)
    EPI_PATCH = 4,       // C[b*Lout+1+p] = h(h(acc) + pos[1+p])     conv1 + pos add      (clip/model.py:412-416)
    EPI_SCALE = 5,       // C = h(h(acc) * scale)                    logit_scale * einsum (trainers/mm_classifier_one_prompt.py:263)
    // LayerNorm folded into the consuming nn.Linear (gemm_f16_v5 only).  A holds the RAW residual stream x, W holds
    // h(gamma (.) W), and the epilogue applies the row statistics:  LN(x) W^T + bias
    //   = rstd[m] * (acc[m,n] - mean[m] * ln_g[n]) + ln_b[n],  ln_g[n] = sum_k W'[n,k],  ln_b[n] = bias[n] + sum_k beta[k] W[n,k].
    // The reference rounds LN(x) to fp16 before the GEMM (clip/model.py:153-159); here the rounding sits on gamma (.) W
    // instead -- same magnitude of error, one full read + write of the activations less per LayerNorm.
    EPI_LN_BIAS = 6,        // C = h(LN-folded acc)                  ln_1 + in_proj       (clip/model.py:192)
    EPI_LN_BIAS_QGELU = 7,  // u = h(LN-folded acc); C = u * sigmoid(1.702u)   ln_2 + c_fc + QuickGELU (clip/model.py:193)
    // Cross-validation logits that are never written (gemm_f16_v5 only): u = h(h(acc) * scale) as EPI_SCALE, then per row
    // the (maximum, lowest column holding it) of the tile's 256 columns -> argmax_out[m][tn] = {max as float, column};
    // launch_argmax_reduce finishes the row argmax over the N tiles and counts.      (trainers/mm_classifier_one_prompt.py:263-270)
    EPI_SCALE_ARGMAX = 8,
};

struct GemmArgs {
    const void* A; int lda;     // [M,K] row-major
    const void* W; int ldw;     // [N,K] row-major (nn.Linear weight layout)
    void* C; int ldc;           // [M,N] row-major
    const void* bias;           // [N] or nullptr
    const void* res; int ldres; // [M,N] residual (may alias C) or nullptr
    const void* pos;            // EPI_PATCH: fp16 [Lout, N]
    int M, N, K;
    int epi;
    int rows_in, rows_out;      // EPI_PATCH: patches per image (G*G) and tokens per image (G*G+1)
    float scale;                // EPI_SCALE
    int n_group;                // N tiles per L2 group (tile order, set by the launcher; 0 = all)
    int a_blocked, w_blocked;   // operand stored as [rows/128][K/64][128][64] (16 KiB contiguous per (row block, K-tile))
    // LayerNorm folding (see EPI_LN_BIAS).  Row statistics travel as per-row PARTIAL sums [M][slots][2] fp32
    // (sum, sum of squares), one slot per 256-column tile of the producing GEMM, summed in slot order by the consumer
    // (deterministic: no atomics).
    const float* ln_stats; int ln_slots;   // EPI_LN_*: statistics of the A rows over K
    int ln_stride;                         // ... slots between the statistics of consecutive A rows (0 = ln_slots: dense; the CLS-row launch strides over tokens)
    const float* ln_g; const float* ln_b;  // EPI_LN_*: [N] fp32 each
    float* stats_out;                      // EPI_BIAS_RES, optional: partial statistics of the stored C rows, [M][N/256][2]
    int nt_store;                          // v5: 0 = auto (streaming stores when C is much larger than the L2s), 1 = never, 2 = always
    float* argmax_out;                     // EPI_SCALE_ARGMAX: [M][ceil(N/256)][2] (value, column index bits)
    int gelu_mode;                         // QuickGELU epilogues: 0 = the reference's three fp16 rounding points, 1 = quick_gelu_f32x2 (set from variant / 100)
    // EPI_PATCH, gemm_f16_v5 only: A is the fp16 IMAGE tensor [B, 3, R, R] itself and the K loop gathers the patch rows
    // (k = c * 256 + ky * 16 + kx, 16 x 16 patches) straight into LDS -- no im2col pass.  0 = A is the [M, K] matrix.
    int im2col_R;
};

__device__ __forceinline__ float quick_gelu_h(float u) {
    // fp16 rounding points of `x * torch.sigmoid(1.702 * x)` on an fp16 tensor
    // (v_exp_f32 / v_rcp_f32 are 1-ulp fp32 approximations; their result is rounded to fp16 right away)
    half_t t = (half_t)(1.702f * u);
    half_t s = (half_t)__builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * (float)t));
    return (float)(half_t)(u * (float)s);
}

// The same rounding points on a PAIR of fp16 values, written so that hipcc selects packed / mixed-precision instructions:
// t = h(1.702 u) (v_fma_mixlo/hi), e = exp2(-log2e * t) with the fp16 -> fp32 conversion folded into v_fma_mix_f32,
// s = h(1 / (1 + e)) as one v_cvt_pk_f16_f32, u * s as one v_pk_mul_f16: ~9.5 issue slots per element instead of 12.5
// (the epilogue's vector instructions are exposed time: nothing overlaps a tile's epilogue, profiles/r01g_gemm_epilogue.md).
__device__ __forceinline__ half2_t quick_gelu_h2(half2_t u) {
    half2_t t;
    t[0] = (half_t)(1.702f * (float)u[0]);
    t[1] = (half_t)(1.702f * (float)u[1]);
    // The epilogue's QuickGELU is bound by the TRANSCENDENTAL unit (2 per element: ~3.2 us of a ~30 us c_fc tile, unchanged
    // when 25 % of the other vector instructions were removed), so the pair shares ONE reciprocal:
    // 1/a = b / (a b), 1/b = a / (a b).  The exponent argument is clamped at 60 (sigmoid < 2^-60 is 0 in fp16 either way),
    // which keeps a * b finite.
    float2_t e;
    e[0] = __builtin_amdgcn_exp2f(fminf(__builtin_fmaf((float)t[0], -1.4426950408889634f, 0.0f), 60.0f));
    e[1] = __builtin_amdgcn_exp2f(fminf(__builtin_fmaf((float)t[1], -1.4426950408889634f, 0.0f), 60.0f));
    const float a = 1.0f + e[0], b = 1.0f + e[1];
    const float rab = __builtin_amdgcn_rcpf(a * b);
    float2_t r;
    r[0] = b * rab;
    r[1] = a * rab;
    const half2_t s = __builtin_convertvector(r, half2_t);
    return u * s;
}

// Four values at a time: the four denominators a b c d share ONE reciprocal, R = 1 / (abcd); 1/(ab) = cd R, 1/(cd) = ab R,
// 1/a = b / (ab) ...: 1.25 transcendental + 1.25 packed-fp32 multiplies per element instead of 1.5 + 1.5 (the transcendental unit
// is what bounds QuickGELU: quarter rate).  Exponent arguments are clamped at 30 (a sigmoid below 2^-25 rounds to fp16 zero either
// way), which keeps abcd below 2^124.
__device__ __forceinline__ half4_t quick_gelu_h4(half4_t u) {
    half4_t t;
    float4_t e;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        t[k] = (half_t)(1.702f * (float)u[k]);
        e[k] = __builtin_amdgcn_exp2f(fminf(__builtin_fmaf((float)t[k], -1.4426950408889634f, 0.0f), 30.0f));
    }
    const float2_t p = (float2_t){e[0], e[2]} + 1.0f, q = (float2_t){e[1], e[3]} + 1.0f;
    const float2_t pq = p * q;                                     // (ab, cd)
    const float R = __builtin_amdgcn_rcpf(pq[0] * pq[1]);
    const float2_t rr = (float2_t){pq[1], pq[0]} * R;              // (1/(ab), 1/(cd))
    const float2_t sp = q * rr, sq = p * rr;                       // (1/a, 1/c), (1/b, 1/d)
    const half4_t s = __builtin_convertvector((float4_t){sp[0], sq[0], sp[1], sq[1]}, half4_t);
    return u * s;
}

// QuickGELU with ONE rounding point instead of the reference's three (ovmr_set_option "gelu_exact" = 0): the GEMM epilogue calls
// this on the fp32 value x = acc + bias BEFORE it is rounded to fp16 and rounds the product once, g = h(x / (1 + 2^(-1.702 log2(e) x)))
// -- v_mul_f32, v_exp_f32, v_add_f32, v_rcp_f32, v_mul_f32 at the full fp32 rate plus half a v_cvt_pk_f16_f32 per element, where
// quick_gelu_h4 takes a quarter-rate v_fma_mixlo_f16, 1.25 transcendentals and 5 more instructions (profiles/r03a_valu_rate.log: per SIMD
// with two waves, v_fma/mul/add_f32 2.6 cycles, other VOP3 4.4, transcendentals / v_fma_mixlo_f16 / v_permlane16_swap 8.3).  It is
// CLOSER to the real function than the reference's fp16 form (no h(u), h(1.702 u), h(sigmoid) in between); against the reference's
// result it differs by at most a few fp16 steps of the result (bound stated in tests/test_hip_kernels.py).  x < -52: e = +inf, s = 0, g = -0.
__device__ __forceinline__ float2_t quick_gelu_f32x2(float2_t x) {
    float2_t g;
#pragma unroll
    for (int k = 0; k < 2; ++k) {
        const float e = __builtin_amdgcn_exp2f(x[k] * -2.4554669595930157f);
        g[k] = x[k] * __builtin_amdgcn_rcpf(1.0f + e);
    }
    return g;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

#define HIP_CHECK_RET(expr)                                   \
    do {                                                      \
        hipError_t _e = (expr);                               \
        if (_e != hipSuccess) return (int)_e;                 \
    } while (0)

// ---- launchers (one per .hip translation unit) -------------------------------------------
int launch_gemm_f16(const GemmArgs& a, int variant, hipStream_t s);
bool gemm_f16_is_small(int M, int N);   // latency-bound shape: variant 8 runs it on the 64 x 64 split-K kernel (gemm_f16_small.hip)
int launch_gemm_f32(const GemmArgs& a, hipStream_t s);
int launch_layernorm(const void* x, void* y, const float* g, const float* b, int rows, int D,
                     long in_row_stride, int is_f32, hipStream_t s);
int launch_row_stats(const half_t* x, float* stats, int rows, int D, int slots, hipStream_t s);
int launch_fold_ln(const half_t* W, const float* gamma, const float* beta, const half_t* bias, half_t* Wf, float* g, float* b,
                   int N, int K, hipStream_t s);
int launch_attention_f16(const half_t* qkv, half_t* out, int B, int L, int H, int causal, int variant, hipStream_t s);
int launch_attention_f16_q(const half_t* qkv, half_t* out, int B, int L, int Lq, int H, int causal, int variant, hipStream_t s);
// up to 4 groups of sequences of at most 32 tokens in one launch (attention_short.hip); -100: not taken
int launch_attention_f16_short(const half_t* qkv, half_t* out, int n_groups, const int* nseq, const int* L, const long* row0, int H,
                               int causal, hipStream_t s);
int launch_attention_f32(const float* qkv, float* out, int B, int L, int H, hipStream_t s);
int launch_im2col(const void* img, int img_is_f32, half_t* out, int B, int R, int P, int Kpad, hipStream_t s);
int launch_fill_cls(half_t* x, const half_t* cls_pos, int B, int L, int W, hipStream_t s);
int launch_l2norm_f16(half_t* x, int rows, int D, hipStream_t s);
int launch_cast(const void* src, int src_f32, void* dst, int dst_f32, long n, hipStream_t s);
int launch_transpose_to_f16(const void* src, int src_f32, half_t* dst, int rows, int cols, hipStream_t s);
int launch_pad_rows_f16(const half_t* src, half_t* dst, int rows, int cols, int cols_pad, hipStream_t s);
int launch_add_f16(const half_t* a, const half_t* b, half_t* out, long n, hipStream_t s);
int launch_text_embed_ids(const int64_t* ids, int ids_stride, const float* tok_emb, const half_t* pos16,
                          half_t* x, int* index, int N, int Lctx, int Lseq, int D, hipStream_t s);
int launch_embed_gather(const int64_t* ids, const float* tok_emb, half_t* out, long rows, int D, hipStream_t s);
int launch_text_add_pos(const half_t* prompts, int Lctx, const half_t* pos16, half_t* x, int N, int Lseq, int D, hipStream_t s);
int launch_gather_rows_f16(const half_t* x, const int* index, half_t* out, int N, int Lseq, int D, hipStream_t s);
int launch_pack_rows(const half_t* mm, const half_t* v, const half_t* t, const half_t* tokens, const int64_t* labels, int n, int D, int n_ctx,
                     int bound, half_t* block, hipStream_t s);
int launch_unpack_rows(const half_t* gathered, int rows, int C, int D, int n_ctx, half_t* mm, half_t* v, half_t* t, half_t* tokens, int* seen,
                       hipStream_t s);
int launch_agg_input(const float* cls_token, const half_t* feats, float* x, int Cb, int S, int n_ctx, int D, hipStream_t s);
int launch_agg_output(const float* x, float* tokens, int Cb, int La, int n_ctx, int D, hipStream_t s);
int launch_assemble_prompts(const half_t* base, const int64_t* labels, const float* tokens, half_t* out,
                            int Cb, int Lctx, int n_ctx, int D, hipStream_t s);
int launch_argmax_counts(const half_t* logits, int ld, const int* labels, int R, int C, int* tp, int* n_pred, hipStream_t s);
int launch_argmax_reduce(const float* partial, int tiles, const int* labels, int R, int C, int* tp, int* n_pred, hipStream_t s);
int launch_fusion_weights(const int* counts, const int* n_label, int C, float tau, float* out, hipStream_t s);
int launch_eval_counts(const void* out, int out_is_f32, long ld, const int64_t* labels, int B, int C, int* counts, hipStream_t s);
int launch_scale_f16(const half_t* x, half_t* y, float scale, long n, hipStream_t s);
struct ovmr_resize_job;
int launch_resize_crop_u8(const uint8_t* pixels, const ovmr_resize_job* jobs, int n, const int32_t* tables, uint8_t* tmp, int max_ny,
                          uint8_t* out, int R, hipStream_t s);
int launch_preprocess_u8(const uint8_t* in, half_t* out, int B, int R, const float* mean3, const float* std3, hipStream_t s);
// one-launch classifier head (head_fused.hip); -100: shape not taken
size_t head_fused_ws_bytes(int B, int C);
int head_fused_sync_ints();
int launch_head_fused(const half_t* feats, int B, int D, float scale, const half_t* const* clf, int n_mod, int C, const float* w,
                      float* out, half_t* raw_out, void* ws, int* sync, int n_cu, int max_grid, hipStream_t s);
int launch_fused_softmax(const half_t* l0, const half_t* l1, const half_t* l2, const float* w, int n_mod,
                         float* out, int B, int C, hipStream_t s);
