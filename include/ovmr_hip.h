/* libovmr_hip.so -- C ABI of the MI355X (gfx950) implementation of OVMR's hot path.
 *
 * The reference (Zehong-Ma/OVMR) has no FFI: its boundary for this path is a set of Python
 * nn.Module methods.  Each entry point below replaces the arithmetic of one of them; the Python
 * layer in ovmr_amd/modules.py keeps the reference's module API on top (INTEGRATION.md shows the
 * ctypes stub a maintainer of the reference would add).  Paths are relative to the reference root.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer owned by the caller unless stated otherwise (PyTorch
 *     allocates, passes tensor.data_ptr()); the library owns only its weight arena and workspace;
 *   - every compute call is asynchronous on the caller's hipStream_t, performs no allocation and
 *     no host synchronisation (graph-capturable), and may be called with any batch size up to
 *     what ovmr_finalize() sized the workspace for (larger batches are processed in chunks);
 *   - return value: 0 = ok, otherwise an OVMR_E_* code or a positive hipError_t;
 *     ovmr_last_error(h) returns a message for the last failure on that handle;
 *   - one handle per (device, model, stream user): thread-compatible, not thread-safe.  Calls on ONE handle must be ordered on ONE
 *     stream (or by events): the workspace and the device-side phase counters of ovmr_fused_logits belong to the handle, and two
 *     launches on it in flight at once would corrupt both.  Concurrency = several handles (CustomCLIP.forward_batches uses two);
 *   - fp16 = IEEE binary16 ("half"), the reference's only working OVMR precision
 *     (configs/trainers/MM_CLS_OP/vit_b16_c4_ep50_imagenet21k_pretrain.yaml:42).
 */
#ifndef OVMR_HIP_H
#define OVMR_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ovmr_handle ovmr_handle;
typedef void* ovmr_stream;              /* hipStream_t */

enum { OVMR_F16 = 0, OVMR_F32 = 1, OVMR_I64 = 2, OVMR_I32 = 3 };
enum { OVMR_MODE_FUSION = 0, OVMR_MODE_TEXT = 1, OVMR_MODE_VISION = 2, OVMR_MODE_MULTIMODAL = 3 };
enum {
    OVMR_OK = 0,
    OVMR_E_ARG = -1,         /* bad argument / shape            */
    OVMR_E_SHAPE = -2,       /* unsupported dimension           */
    OVMR_E_STATE = -4,       /* weights missing / not finalized */
    OVMR_E_NOMEM = -5,
    OVMR_E_NAME = -6         /* unknown weight name             */
};

/* Architecture, exactly the constructor arguments of CLIP (clip/model.py:717-731) that
 * build_model() infers from a state dict (clip/model.py:899-928), plus the prompt learner's
 * n_ctx (trainers/mm_classifier_one_prompt.py:99) and its fixed 4 aggregator layers (:140). */
typedef struct {
    int embed_dim;            /* 512 (ViT-B), 768 (ViT-L)        */
    int image_resolution;     /* 224 / 336                       */
    int vision_layers;
    int vision_width;         /* heads = width / 64              */
    int vision_patch_size;
    int context_length;       /* 77                              */
    int vocab_size;           /* 49408                           */
    int transformer_width;    /* heads = width / 64              */
    int transformer_layers;
    int n_ctx;                /* visual tokens per class         */
    int agg_layers;           /* 4                               */
} ovmr_model_desc;

int  ovmr_create(const ovmr_model_desc* desc, ovmr_handle** out);
void ovmr_destroy(ovmr_handle* h);
const char* ovmr_last_error(const ovmr_handle* h);
const char* ovmr_version(void);

/* Kernel-variant switch used by tests/bench to A/B implementations: key in {"gemm","attn","ln_fold","xval_fused","gelu_exact","fuse_im2col","enc_chunk","last_q_cls","fused_head","head_max_grid"}.
 * "gemm" (default 8): 8 = 256-row LDS-DMA tiles with the ping-pong K loop (half-tile staging, counted waits) and, for latency-bound shapes
 *   (at most one round -- 256 -- of 64 x 64 tiles: the text tower on a few dozen prompts, CLS-row chains), the 64 x 64 kernel that splits K
 *   over its waves; 7 = 8 without that kernel (A/B); 6 = the 256-row tiles with the double-buffered K loop; 9 = the 64 x 64 split-K kernel wherever it takes the shape (tests);
 *   0 = the 128x128 register-staged kernel everywhere (LayerNorm-folding and fused-argmax launches still take the tile kernel).
 * "attn" (default 3): 3 = single-pass persistent kernel where the shape is its own (non-causal, 192 < L <= 208), the 32x32x16 flash
 *   kernel (= 5) for non-causal L >= 256 (ViT-L), else as 1; 5 = that kernel where it applies, else as 1;
 *   1 = flash-style LDS-DMA kernel for L >= 128, else as 0; 0 = the plain flash-style kernel.  Every variant but 0 runs sequences of at
 *   most 32 tokens (truncated text prompts) on the one-wave-per-(sequence, head) kernel, bit-equal to 0.
 * "ln_fold" (default 1): ln_1 / ln_2 of the fp16 towers are folded into the consuming GEMM where the shape allows
 *   (width % 256 == 0 and >= 256 token rows); 0 runs the separate LayerNorm kernel everywhere.
 * "xval_fused" (default 1): ovmr_xval_counts takes the row argmax inside the logits GEMM's epilogue (the [R, C] logits are never
 *   written); 0 materialises fp16 logits in workspace chunks and runs a row-argmax kernel on them.  Identical counters.
 * "gelu_exact" (default 0): QuickGELU of the c_fc epilogue (clip/model.py:162-164).  1 keeps the three fp16 rounding points of the
 *   reference's fp16 tensors (h(1.702 u), h(sigmoid), h(u s)); 0 evaluates x / (1 + exp(-1.702 x)) in fp32 on the unrounded
 *   linear output and rounds once -- closer to the real function, within a few fp16 steps of the reference's value, and 5 % off
 *   the c_fc launch.
 * "fuse_im2col" (default 1): conv1 on fp16 images with 16 x 16 patches -- the patch-embedding GEMM gathers its A rows from the image
 *   tensor inside its K loop; 0 writes the patch matrix out first (what fp32 images and other patch sizes always do).  Bit-identical.
 * "last_q_cls" (default 1): the last vision block (whose output is read at the CLS row only, clip/model.py:423) projects K and V for every
 *   token and Q for the CLS rows alone when a launch sequence holds at least 256 images; 0 projects Q for every token.  Bit-identical.
 * "fused_head" (default 1): ovmr_fused_logits / ovmr_zeroshot_logits as ONE launch (csrc/head_fused.hip: scaled features staged once, the up
 *   to three products on the matrix pipe, both fp16 rounding points of trainers/mm_classifier_one_prompt.py:357-363, softmax and weighted
 *   sum) for up to 256 query rows, or up to 512 rows x 2048 classes; larger calls run scale + GEMMs + softmax as separate launches.
 *   2 = one launch at any size; 0 = never.  The two implementations sum K in different orders: a logit may land on the neighbouring fp16
 *   value (tests/test_hip_parity.py:test_fusion_head_vs_oracle holds both to the oracle).
 * "head_max_grid" (default 0 = one workgroup per tile): caps the one-launch head's grid (tests: workgroups then take several tiles). */
int ovmr_set_option(ovmr_handle* h, const char* key, int value);

/* Weight ingestion -- replaces build_model()/convert_weights()/load_state_dict
 * (clip/model.py:852-873,899-936) and PromptLearner.load_state_dict
 * (trainers/mm_classifier_one_prompt.py:461-493).  `name` is the reference state-dict key
 * ("visual.conv1.weight", "transformer.resblocks.3.attn.in_proj_weight", "text_projection",
 * "logit_scale", ... and, prefixed "prompt_learner.", "cls_token" /
 * "aggregator.resblocks.N.*").  Data is copied (and converted to the reference's dtype policy:
 * Linear/Conv/MHA/proj fp16, LayerNorm + embeddings fp32, aggregator fp32) on `stream`. */
int ovmr_set_weight(ovmr_handle* h, const char* name, const void* data, int dtype,
                    int ndim, const int64_t* shape, ovmr_stream stream);
/* Validates that every tensor is present, builds derived layouts (transposed projections,
 * fp16 positional tables) and sizes the workspace for `max_images` images / `max_prompts`
 * text sequences / `max_classes` classes per call.  Synchronises `stream` once. */
int ovmr_finalize(ovmr_handle* h, int max_images, int max_prompts, int max_classes, ovmr_stream stream);

/* VisionTransformer.forward (clip/model.py:411-428) [+ x / x.norm() when normalize != 0,
 * trainers/mm_classifier_one_prompt.py:244,307].  image: [B,3,R,R] fp16 or fp32 (cast to fp16 on
 * device like image.type(self.dtype), :243/:306); out: [B, embed_dim] fp16. */
int ovmr_encode_image(ovmr_handle* h, const void* image, int image_dtype, int B,
                      void* out_f16, int normalize, ovmr_stream stream);

/* The launch sequences ovmr_encode_image splits a batch of B images into (diagnostics / bench accounting; no reference
 * counterpart: the reference encodes a loader batch in one piece, trainers/mm_classifier_one_prompt.py:243-245): writes up to
 * max_sizes image counts into sizes, returns the number of sequences (-1 before ovmr_finalize). */
int ovmr_encode_plan(const ovmr_handle* h, int B, int* sizes, int max_sizes);

/* TextEncoder.forward (trainers/mm_classifier_one_prompt.py:80-91).  prompts: [N, context_length,
 * transformer_width] fp16 (already embedded); index: [N] int32 row that is projected (EOS for mm
 * prompts, 1+n_ctx for vision prompts, :163-165); seq_len: number of leading positions that must
 * be computed (max(index)+1 <= seq_len <= context_length; the mask is causal so truncation is
 * exact); normalize: 0 = raw, 1 = x/x.norm(), 2 = the double normalisation of get_mm_v_feats
 * (:204,210).  out: [N, embed_dim] fp16. */
int ovmr_encode_text_embedded(ovmr_handle* h, const void* prompts_f16, const int32_t* index, int N,
                              int seq_len, void* out_f16, int normalize, ovmr_stream stream);

/* CLIP.encode_text (clip/model.py:820-833): ids [N, context_length] int64; the projected row is
 * ids.argmax(-1).  normalize as above (the zero-shot text classifier uses 1, :118-126). */
int ovmr_encode_text_ids(ovmr_handle* h, const int64_t* ids, int N, int seq_len,
                         void* out_f16, int normalize, ovmr_stream stream);

/* Several prompt families through the text transformer in ONE pass.  The reference encodes the multimodal prompts, the vision
 * prompts (get_mm_v_feats, trainers/mm_classifier_one_prompt.py:200-212) and the zero-shot text prompts (:118-126) as three
 * separate TextEncoder / encode_text calls; the arithmetic of a sequence does not depend on its neighbours, so here the token
 * rows of all groups share the GEMM launches of every block and only attention, the read-out gather and the projection run per
 * group (each group keeps its own seq_len: truncation to the last needed row is exact under the causal mask, clip/model.py:802-809).
 * A group is EITHER embedded prompts + read-out rows (as ovmr_encode_text_embedded) OR token ids (as ovmr_encode_text_ids).
 * Rows that do not fit the workspace fall back to one pass per group.  Results equal the per-group entry points up to the
 * kernel choice the larger row count implies (1 - cos ~ 1e-6). */
typedef struct {
    const void* prompts_f16;   /* [n, context_length, transformer_width] fp16, or NULL when ids is given */
    const int64_t* ids;        /* [n, context_length] int64, or NULL */
    const int32_t* index;      /* [n] read-out rows (embedded prompts only) */
    int n, seq_len, normalize; /* as the single-group entry points */
    void* out_f16;             /* [n, embed_dim] fp16 */
} ovmr_text_group;
int ovmr_encode_text_groups(ovmr_handle* h, const ovmr_text_group* groups, int n_groups, ovmr_stream stream);

/* token_embedding(ids).type(fp16) (trainers/mm_classifier_one_prompt.py:129-130): ids [N, L] int64
 * -> out [N, L, transformer_width] fp16. */
int ovmr_embed_tokens(ovmr_handle* h, const int64_t* ids, int N, int L, void* out_f16, ovmr_stream stream);

/* PromptLearner.forward, visual-token generator part (trainers/mm_classifier_one_prompt.py:167-169):
 * feats [Cb, S, embed_dim] fp16 -> tokens [Cb, n_ctx, embed_dim] fp32 (aggregator in fp32). */
int ovmr_generate_tokens(ovmr_handle* h, const void* feats_f16, int Cb, int S,
                         float* tokens_f32, ovmr_stream stream);

/* PromptLearner.update_prompts (:156-157): out[c] = cat(base[row(c)][:2], half(tokens[c]),
 * base[row(c)][2:-n_ctx]); row(c) = labels[c] (int64) or 0 when labels == NULL (the "a ." template
 * broadcast, :173).  base: [*, context_length, width] fp16; out: [Cb, context_length, width] fp16. */
int ovmr_assemble_prompts(ovmr_handle* h, const void* base_f16, const int64_t* labels,
                          const float* tokens_f32, int Cb, void* out_f16, ovmr_stream stream);

/* Cross-validation counts (trainers/mm_classifier_one_prompt.py:261-270): for rows r of
 * feats [R, embed_dim] fp16 with int32 labels [R], logits = logit_scale * feats @ clf^T in fp16
 * (clf [C, embed_dim] fp16), pred = argmax (first index on ties); accumulates
 * n_pred[pred] += 1 and tp[label] += (pred == label) into int32 [C] arrays (caller zeroes). */
int ovmr_xval_counts(ovmr_handle* h, const void* feats_f16, const int32_t* labels, int R,
                     const void* clf_f16, int C, int32_t* tp, int32_t* n_pred, ovmr_stream stream);

/* F1 preference (:268-274): counts int32 [3][2][C] = {mm, vision, text} x {tp, n_pred},
 * n_label int32 [C]; out [C,3] fp32 = softmax(tau * f1) with f1 = 2pr/(p+r), NaN -> 0
 * (torcheval==0.0.7 multiclass_f1_score(average=None)). */
int ovmr_fusion_weights(ovmr_handle* h, const int32_t* counts, const int32_t* n_label, int C,
                        float tau, float* out_f32, ovmr_stream stream);

/* CustomCLIP.forward eval branch (:348-363).  feats [B, embed_dim] fp16 (L2-normalised);
 * mm / v / t classifiers [C, embed_dim] fp16 (unused ones may be NULL for single modes);
 * w [C,3] fp32 (fusion only); out [B,C] fp32 probabilities. */
int ovmr_fused_logits(ovmr_handle* h, const void* feats_f16, int B, const void* mm_f16,
                      const void* v_f16, const void* t_f16, const float* w_f32, int C, int mode,
                      float* out_f32, ovmr_stream stream);

/* Which implementation ovmr_fused_logits runs for B query rows and C classes on this handle: 1 = the one-launch head, 0 = scale + GEMMs +
 * softmax (-1: bad arguments / not finalized).  Both keep the reference's rounding points (:357-363) but sum K in different orders, so a
 * logit may land on the neighbouring fp16 value: a caller that splits one batch over several calls and promises the unsplit call's bits
 * (CustomCLIP.forward's two halves) splits only where the plan is the same. */
int ovmr_head_plan(const ovmr_handle* h, int B, int C);

/* ZeroshotCLIP.model_inference (trainers/zsclip.py:55-60): out [B,C] fp16 =
 * exp(logit_scale) * feats @ text_feats^T (raw logits). */
int ovmr_zeroshot_logits(ovmr_handle* h, const void* feats_f16, int B, const void* text_f16, int C,
                         void* out_f16, ovmr_stream stream);

/* Class-sharded generation (no counterpart in the reference, which has no distributed path: trainers/mm_classifier_one_prompt.py:414-419
 * wraps the model in nn.DataParallel): a rank's contribution to the job's ONE all-gather.  block [bound, K + 2] fp16 with
 * K = 3 D + n_ctx D: row i < n = mm[c] | v[c] | t[c] | tokens[c] for c = labels[i] (int64 class ids; mm / v / t are the class-indexed
 * [C, D] fp16 buffers of forward_prompt, :216-225, tokens [C, n_ctx, D] fp16) followed by c as an int32 bit pattern in two fp16 columns;
 * rows n <= i < bound are zero with label -1.  D even.  Needs no handle. */
int ovmr_pack_rows(const void* mm_f16, const void* v_f16, const void* t_f16, const void* tokens_f16, const int64_t* labels, int n,
                   int D, int n_ctx, int bound, void* block_f16, ovmr_stream stream);

/* The inverse, on the gathered blocks of all ranks ([rows, K + 2] fp16, rank-major): every row whose label lies in [0, C) is written to
 * row `label` of mm / v / t [C, D] and tokens [C, n_ctx, D]; seen (int32 [C + 1], zeroed by the caller) counts the rows per class -- the
 * job is complete iff seen[c] == 1 for every class (:259) -- and, in seen[C], labels outside [-1, C). */
int ovmr_unpack_rows(const void* gathered_f16, int rows, int C, int D, int n_ctx, void* mm_f16, void* v_f16, void* t_f16,
                     void* tokens_f16, int32_t* seen, ovmr_stream stream);

/* Classification.process of the test loop (Dassl.pytorch/dassl/evaluation/evaluator.py:50-67): outputs [B, C] (fp32 = what
 * CustomCLIP.forward returns, or fp16 = ZeroshotCLIP's raw logits; row stride ld elements), labels int64 [B] (the batch's LongTensor as
 * it is).  pred = outputs.max(1)[1] -- lowest column on ties, a NaN is the largest value -- then counts[0][pred] += (pred == label)
 * (true positives), counts[1][pred] += 1, counts[2][label] += 1; counts is int32 [3][C] followed by ONE more int32 that counts the rows
 * whose label lies outside [0, C) (such rows touch no histogram; the host raises when it is not zero).  The caller zeroes counts once
 * and accumulates over the batches of a test pass: no host round trip per batch (the reference does .item() / .cpu() per batch, :59-67);
 * accuracy, error rate, macro-F1 and the per-class tables of evaluate() (:69-138) are functions of these 3C integers.  Needs no handle. */
int ovmr_eval_counts(const void* outputs, int dtype, long ld, const int64_t* labels, int B, int C, int32_t* counts, ovmr_stream stream);

/* exp(logit_scale) as held by the handle (set through ovmr_set_weight("logit_scale")). */
float ovmr_logit_scale(const ovmr_handle* h);

/* Device half of the test transform -- replaces ToTensor + Normalize of _build_transform_test
 * (Dassl.pytorch/dassl/data/transforms/transforms.py:495-526; mean / std of configs/trainers/MM_CLS_OP/*.yaml:14-15) and the
 * .type(fp16) of trainers/mm_classifier_one_prompt.py:243,306.  u8_hwc: DEVICE uint8 [B, R, R, 3], the decoded, bicubic-resized and
 * centre-cropped images (the host's share of the transform); out_f16: fp16 [B, 3, R, R] = h(((u / 255) - mean[c]) / std[c]), the
 * fp32 arithmetic of torchvision bit for bit.  mean3 / std3 are HOST pointers to three floats.  R % 8 == 0.  Needs no handle. */
int ovmr_preprocess_u8(const void* u8_hwc, int B, int R, const float* mean3, const float* std3, void* out_f16, ovmr_stream stream);

/* Device half of the test transform, first part -- replaces Resize(max(INPUT.SIZE), INPUT.INTERPOLATION) + CenterCrop(INPUT.SIZE) of
 * _build_transform_test (Dassl.pytorch/dassl/data/transforms/transforms.py:495-526; torchvision on PIL images, i.e. PIL.Image.resize +
 * crop) for bicubic and bilinear interpolation, BIT-EQUAL to PIL: PIL's own two integer passes (22-bit fixed-point weights, uint8
 * intermediate) over window / weight tables the host builds exactly as PIL does (ovmr_amd/resize.py).  One job per image:
 *   in_offset   byte offset of its decoded uint8 [h, w, 3] pixels in `pixels` (device)
 *   y0, ny      the input rows the R x R crop window needs;  tmp_offset: byte offset of its [ny, R, 3] intermediate in `tmp`
 *   table       offset (in int32) of its size's table in `tables` (device):
 *               [bounds_h R x 2 (xmin, n) | k_h R x ksize_h | bounds_v R x 2 (ymin - y0, n) | k_v R x ksize_v]
 *   passthrough 1: `pixels` already holds the R x R crop of this image (resized on the host: nearest, or an image too large for the
 *               upload ring) and is copied
 * jobs: DEVICE array of n jobs; max_ny: the largest ny among them; out_u8: DEVICE uint8 [n, R, R, 3], the input of
 * ovmr_preprocess_u8.  Needs no handle. */
typedef struct ovmr_resize_job {
    int64_t in_offset;
    int64_t tmp_offset;
    int32_t w, h;
    int32_t y0, ny;
    int32_t table;
    int32_t ksize_h, ksize_v;
    int32_t passthrough;
} ovmr_resize_job;
int ovmr_resize_crop_u8(const void* pixels, const ovmr_resize_job* jobs, int n, const int32_t* tables, void* tmp, int max_ny,
                        void* out_u8, int R, ovmr_stream stream);

/* Images per launch sequence of ovmr_encode_image after ovmr_finalize (<= max_images): a batch of MORE than max_images images is
 * encoded in chunks of this many (one that fits the workspace is a single launch sequence), chosen so that the 256-row-tile grids of the block GEMMs are whole rounds of the CUs and no launch has more than 600 row tiles (ViT-B/16 on 256 CUs: 775 of a
 * reserve of 775 or more, 665 of 768); ovmr_set_option(h, "enc_chunk", n) pins it (0 = automatic).  Chunks of at least 256
 * token rows give bit-identical features; smaller ones take the small-matrix kernels (other rounding points, same tolerance).
 * 0 before ovmr_finalize. */
int ovmr_encode_chunk(const ovmr_handle* h);

/* Closed-form FLOP counts.  ovmr_flops_per_image is the ALGORITHMIC count of the reference's
 * VisionTransformer.forward (every block over every token, clip/model.py:411-428: 35.127 GFLOP for ViT-B/16);
 * ovmr_flops_per_image_executed is what ovmr_encode_image launches for a batch that fills the reserve: its last block runs
 * attention / out_proj / MLP for the CLS query row only (identical result), ~6 % fewer, and -- in launch sequences of at least 256
 * images -- the Q projection too.  ovmr_flops_executed(h, B) is the count for ONE call of ovmr_encode_image on B images: the plan
 * actually run (ovmr_encode_plan), the 256-image rule applied per launch sequence (a query batch of 128 projects Q for every token). */
double ovmr_flops_per_image(const ovmr_handle* h);
double ovmr_flops_per_image_executed(const ovmr_handle* h);
double ovmr_flops_executed(const ovmr_handle* h, int B);
double ovmr_flops_per_prompt(const ovmr_handle* h, int seq_len);

/* Unit-test hooks: launch ONE kernel (no handle).  f32 selects the fp32 (aggregator) kernels;
 * `variant` selects the kernel implementation as ovmr_set_option does; `epi` is the epilogue id of
 * ovmr_amd/csrc/common.h.  A [M,K], W [N,K], C [M,ldc]; qkv [B*L, 3*H*64] -> out [B*L, H*64].
 * For the LayerNorm-folding epilogues (epi 6/7) ovmr_debug_gemm reads `bias` as the folded bias fp32 [N], `pos` as the
 * column sums fp32 [N] and `res` as the row statistics fp32 [M][K/256][2]; with epi 8 (fused row argmax) C receives
 * fp32 [M][ceil(N/256)][2] = (tile maximum, bits of the lowest column holding it); with epi 3 a non-NULL `pos` receives the
 * statistics of the stored rows, fp32 [M][N/256][2]. */
int ovmr_debug_gemm(int f32, int variant, const void* A, const void* W, const void* bias, const void* res,
                    const void* pos, void* C, int M, int N, int K, int ldc, int epi, float scale,
                    int rows_in, int rows_out, ovmr_stream stream);
/* kernel-test hook for the LayerNorm-folding GEMM epilogues (ovmr_amd/csrc/common.h, EPI_LN_BIAS):
 * x1 = h(h(A1 W1^T + b1) + res) [M,D] (skipped when A1 is NULL: x1 is then an input), C2 = h(LN(x1) W2^T + b2) [M,N2],
 * QuickGELU on top when qgelu != 0.  Mirrors clip/model.py:191-194 (ln_1 + in_proj, ln_2 + c_fc). */
int ovmr_debug_lnfold(int variant, const void* A1, const void* W1, const void* b1, const void* res, int M, int D, int K1,
                      const void* W2, const float* gamma, const float* beta, const void* b2, int N2, int qgelu,
                      void* x1, void* C2, ovmr_stream stream);
int ovmr_debug_layernorm(int f32, const void* x, void* y, const float* g, const float* b, int rows, int D,
                         long in_stride, ovmr_stream stream);
int ovmr_debug_attention(int f32, int variant, const void* qkv, void* out, int B, int L, int H, int causal,
                         ovmr_stream stream);

#ifdef __cplusplus
}
#endif
#endif /* OVMR_HIP_H */
