// C ABI of libovmr_hip.so (see include/ovmr_hip.h): weight arena, workspace and the launch
// sequences of the OVMR hot path.  No call in the compute entry points allocates, copies to the
// host or synchronises, so every one of them can be captured into a hipGraph by the caller.
#include "../../include/ovmr_hip.h"
#include "common.h"

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#ifdef OVMR_EXPERIMENTS
int launch_qkv_attn_fused(const half_t* x, const half_t* wf, const float* ln_g, const float* ln_b, const float* stats, half_t* out,
                          int B, int L, int W, hipStream_t s);     // csrc/experiments/qkv_attn_fused.hip
#endif

namespace {

struct Buf {
    void* p = nullptr;
    size_t elems = 0;
    int f32 = 0;
    std::vector<int64_t> shape;
};

struct Block {   // one residual attention block; element type depends on the tower
    void *in_w, *in_b, *out_w, *out_b, *fc_w, *fc_b, *pj_w, *pj_b;
    float *ln1_g, *ln1_b, *ln2_g, *ln2_b;
    // LayerNorm folded into in_proj / c_fc (common.h EPI_LN_BIAS): h(gamma (.) W), column sums, folded bias; built by finalize
    half_t *in_wf = nullptr, *fc_wf = nullptr;
    float *in_g = nullptr, *in_bf = nullptr, *fc_g = nullptr, *fc_bf = nullptr;
};

}  // namespace

struct ovmr_handle {
    ovmr_model_desc d;
    std::map<std::string, Buf> w;
    std::vector<void*> owned;     // weight arena (lives until ovmr_destroy)
    std::vector<void*> derived;   // layouts rebuilt by every ovmr_finalize
    std::string err;
    bool finalized = false;
    int gelu_exact = 0;           // 0 (default): one-rounding fp32 QuickGELU in the c_fc epilogue (common.h quick_gelu_f32x2) -- DEVIATES from the
                                  // reference's fp16 rounding points by up to 8e-3 * max(1, |g|) per value, 6e-7 in 1 - cos end to end (INTEGRATION.md,
                                  // DESIGN.md section 3); 1: the reference's three fp16 rounding points
    int last_q_cls = 1;           // last vision block: Q projected for the CLS rows only (launch sequences of >= 256 images)
    int fuse_im2col = 1;          // patch rows gathered by the patch-embedding GEMM itself (fp16 images, 16 x 16 patches)
    int gemm_variant = 8, attn_variant = 3;   // defaults = fastest verified kernels (tools/gemm_bench.py, tools/attn_bench.py); attention 3 falls back to 1 / 0 by shape
    int ln_fold = 1;                          // fold ln_1 / ln_2 of the fp16 towers into the consuming GEMM where the shape allows
    int xval_fused = 1;                       // cross-validation logits: row argmax fused into the GEMM epilogue (never stored)
    int fused_head = 1;                       // ovmr_fused_logits / ovmr_zeroshot_logits as ONE launch (head_fused.hip); 0: scale + GEMMs + softmax
    int head_max_grid = 0;                    // > 0 caps the fused head's grid (tests: workgroups then take several tiles)
    int fuse_qkv_attn = 0;                    // experiment build only: in_proj + attention of the vision blocks as ONE launch per block (csrc/experiments/qkv_attn_fused.hip)
    int* head_sync = nullptr;                 // the fused head's device counters (zero between launches)
    float logit_scale_exp = 100.f;
    bool have_logit_scale = false;

    std::vector<Block> vis, txt, agg;
    half_t *conv_w = nullptr, *pos16_vis = nullptr, *cls_pos16 = nullptr, *proj_t = nullptr;
    half_t *pos16_txt = nullptr, *textproj_t = nullptr;
    float *ln_pre_g = nullptr, *ln_pre_b = nullptr, *ln_post_g = nullptr, *ln_post_b = nullptr;
    float *ln_final_g = nullptr, *ln_final_b = nullptr, *tok_emb = nullptr, *cls_token = nullptr;
    int Kpad = 0, G = 0, L = 0;

    char* ws = nullptr;
    size_t ws_bytes = 0;
    int max_images = 0, max_prompts = 0, max_classes = 0;
    int enc_chunk_forced = 0;     // option "enc_chunk": > 0 pins the chunk, 0 = pick_encode_chunk
    int n_cu = 0;
    int enc_chunk = 0;            // images per launch sequence of ovmr_encode_image (<= max_images; pick_encode_chunk); option "enc_chunk" overrides
    int enc_fold = 0;             // a remainder of at most this many images (one round of the narrowest GEMM grid) joins the last full sequence
    int img_cap = 0;              // images the workspace holds: max_images + enc_fold
    long agg_rows_cap = 0, logit_elems_cap = 0;
};

namespace {

int fail(ovmr_handle* h, int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    if (h) h->err = buf;
    return code;
}

#define CK(expr)                                                                         \
    do {                                                                                 \
        int _rc = (expr);                                                                \
        if (_rc != 0) return fail(h, _rc, "%s failed with %d (%s:%d)", #expr, _rc, __FILE__, __LINE__); \
    } while (0)

bool ends_with(const std::string& s, const char* suf) {
    const size_t n = strlen(suf);
    return s.size() >= n && s.compare(s.size() - n, n, suf) == 0;
}

// dtype policy of convert_weights() (clip/model.py:852-873); the aggregator is never halved.
bool stored_as_f32(const std::string& name) {
    if (name.rfind("prompt_learner.", 0) == 0) return true;
    if (name.find(".ln_") != std::string::npos || name.rfind("ln_final", 0) == 0) return true;
    if (name == "visual.class_embedding" || name == "visual.positional_embedding" ||
        name == "positional_embedding" || name == "token_embedding.weight" || name == "logit_scale")
        return true;
    if (name.rfind("visual.ln_", 0) == 0) return true;
    return false;   // conv1, proj, text_projection, attn.*, mlp.*
}

void* dev_alloc(ovmr_handle* h, size_t bytes) {
    void* p = nullptr;
    if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) return nullptr;
    h->owned.push_back(p);
    return p;
}

const Buf* find(ovmr_handle* h, const std::string& name, size_t elems) {
    auto it = h->w.find(name);
    if (it == h->w.end()) { fail(h, OVMR_E_STATE, "weight '%s' was never set", name.c_str()); return nullptr; }
    if (it->second.elems != elems) {
        fail(h, OVMR_E_SHAPE, "weight '%s' has %zu elements, expected %zu", name.c_str(), it->second.elems, elems);
        return nullptr;
    }
    return &it->second;
}

int bind_blocks(ovmr_handle* h, const std::string& prefix, int layers, int W, std::vector<Block>& out) {
    out.assign(layers, Block());
    const size_t w = (size_t)W;
    for (int i = 0; i < layers; ++i) {
        const std::string p = prefix + std::to_string(i) + ".";
        struct { const char* n; size_t e; void** dst; } items[] = {
            {"attn.in_proj_weight", 3 * w * w, &out[i].in_w}, {"attn.in_proj_bias", 3 * w, &out[i].in_b},
            {"attn.out_proj.weight", w * w, &out[i].out_w},   {"attn.out_proj.bias", w, &out[i].out_b},
            {"mlp.c_fc.weight", 4 * w * w, &out[i].fc_w},     {"mlp.c_fc.bias", 4 * w, &out[i].fc_b},
            {"mlp.c_proj.weight", 4 * w * w, &out[i].pj_w},   {"mlp.c_proj.bias", w, &out[i].pj_b},
            {"ln_1.weight", w, (void**)&out[i].ln1_g},        {"ln_1.bias", w, (void**)&out[i].ln1_b},
            {"ln_2.weight", w, (void**)&out[i].ln2_g},        {"ln_2.bias", w, (void**)&out[i].ln2_b}};
        for (auto& it : items) {
            const Buf* b = find(h, p + it.n, it.e);
            if (!b) return OVMR_E_STATE;
            *it.dst = b->p;
        }
    }
    return 0;
}

GemmArgs gemm(const void* A, int lda, const void* W, int ldw, void* C, int ldc, int M, int N, int K, int epi,
              const void* bias = nullptr, const void* res = nullptr, int ldres = 0) {
    GemmArgs a;
    memset(&a, 0, sizeof a);
    a.A = A; a.lda = lda; a.W = W; a.ldw = ldw; a.C = C; a.ldc = ldc;
    a.M = M; a.N = N; a.K = K; a.epi = epi; a.bias = bias; a.res = res; a.ldres = ldres; a.scale = 1.f;
    return a;
}

inline size_t align_up(size_t v, size_t a = 256) { return (v + a - 1) / a * a; }

struct Carver {
    char* base; size_t off = 0;
    explicit Carver(char* b) : base(b) {}
    template <typename T> T* take(size_t n) { T* p = (T*)(base + off); off = align_up(off + n * sizeof(T)); return p; }
};

size_t image_ws_bytes(const ovmr_handle* h, long B) {
    const long M = B * h->L, W = h->d.vision_width;
    size_t s = 0;
    s += align_up((size_t)B * h->G * h->G * h->Kpad * 2);
    s += align_up((size_t)M * W * 2) * 2;          // x, y
    s += align_up((size_t)M * 3 * W * 2);          // qkv
    s += align_up((size_t)M * 4 * W * 2);          // mlp hidden
    s += align_up((size_t)B * W * 2);              // CLS rows
    s += align_up((size_t)M * ((W + 255) / 256) * 8);   // LayerNorm partial statistics
    return s;
}
size_t text_ws_bytes(const ovmr_handle* h, long N) {
    const long M = N * h->d.context_length, W = h->d.transformer_width;
    return align_up((size_t)M * W * 2) * 2 + align_up((size_t)M * 3 * W * 2) + align_up((size_t)M * 4 * W * 2) +
           align_up((size_t)N * W * 2) + align_up((size_t)N * 4) + align_up((size_t)M * ((W + 255) / 256) * 8);
}
size_t agg_ws_bytes(const ovmr_handle* h, long rows) {
    const long D = h->d.embed_dim;
    return align_up((size_t)rows * D * 4) * 2 + align_up((size_t)rows * 3 * D * 4) + align_up((size_t)rows * 4 * D * 4);
}

// LayerNorm folding applies when the v5 GEMM takes the shape (it then also emits the statistics): see common.h.
// Not for latency-bound passes (gemm_f16_is_small: the N = W launches -- out_proj, c_proj -- then take the 64 x 64 split-K kernel,
// which emits no statistics; the separate LayerNorm kernel costs ~4 us there).
bool can_fold_ln(const ovmr_handle* h, const Block& k, int M, int W) {
    return h->ln_fold && k.in_wf && M >= 256 && (W % 256) == 0 && W / 256 <= 64 && !(h->gemm_variant == 8 && gemm_f16_is_small(M, W));
}

GemmArgs gemm_ln(GemmArgs a, const float* stats, int slots, const float* g, const float* b) {
    a.ln_stats = stats; a.ln_slots = slots; a.ln_g = g; a.ln_b = b;
    return a;
}
GemmArgs gemm_stats(GemmArgs a, float* stats) {
    a.stats_out = stats;
    return a;
}

// One pre-LN residual attention block (clip/model.py:191-194) on fp16 activations.
// stats != nullptr: ln_1 / ln_2 are folded into in_proj / c_fc.  On entry `stats` holds the partial row statistics of x
// (launch_row_stats, or the previous block's c_proj epilogue); on exit those of the new x.
// `groups` (text tower, ovmr_encode_text_groups): the token rows are several groups of sequences with their own length each --
// the GEMMs run over all rows at once, attention once per group; nullptr: nseq sequences of L tokens.
struct SeqGroup { int nseq, L; long row0; };

int run_block_f16(ovmr_handle* h, const Block& k, half_t* x, half_t* y, half_t* qkv, half_t* hid,
                  int nseq, int L, int W, int causal, hipStream_t s, float* stats = nullptr,
                  const std::vector<SeqGroup>* groups = nullptr) {
    int M = nseq * L;
    if (groups) { M = 0; for (auto& g : *groups) M += g.nseq * g.L; }
    const int H = W / 64, slots = W / 256;
#ifdef OVMR_EXPERIMENTS
    if (stats && h->fuse_qkv_attn && !groups && !causal) {             // experiment: qkv never leaves the CU (bit-equal to the two launches below)
        const int rc = launch_qkv_attn_fused(x, k.in_wf, k.in_g, k.in_bf, stats, y, nseq, L, W, s);
        if (rc != -100) {
            CK(rc);
            goto attention_done;
        }
    }
#endif
    if (stats) {
        CK(launch_gemm_f16(gemm_ln(gemm(x, W, k.in_wf, W, qkv, 3 * W, M, 3 * W, W, EPI_LN_BIAS), stats, slots, k.in_g, k.in_bf), h->gemm_variant, s));
    } else {
        CK(launch_layernorm(x, y, k.ln1_g, k.ln1_b, M, W, W, 0, s));
        CK(launch_gemm_f16(gemm(y, W, k.in_w, W, qkv, 3 * W, M, 3 * W, W, EPI_BIAS, k.in_b), h->gemm_variant, s));
    }
    if (groups) {
        // short groups (every prompt family of a classifier head: <= ~20 tokens) share ONE launch; longer ones run group by group
        int rc = -100;
        if (h->attn_variant >= 1 && groups->size() <= 4) {
            int ns[4], Ls[4];
            long r0[4];
            for (size_t i = 0; i < groups->size(); ++i) { ns[i] = (*groups)[i].nseq; Ls[i] = (*groups)[i].L; r0[i] = (*groups)[i].row0; }
            rc = launch_attention_f16_short(qkv, y, (int)groups->size(), ns, Ls, r0, H, causal, s);
            if (rc != -100) CK(rc);
        }
        if (rc == -100)
            for (auto& g : *groups)
                CK(launch_attention_f16(qkv + g.row0 * 3 * W, y + g.row0 * W, g.nseq, g.L, H, causal, h->attn_variant, s));
    } else
        CK(launch_attention_f16(qkv, y, nseq, L, H, causal, h->attn_variant, s));
#ifdef OVMR_EXPERIMENTS
attention_done:
#endif
    CK(launch_gemm_f16(gemm_stats(gemm(y, W, k.out_w, W, x, W, M, W, W, EPI_BIAS_RES, k.out_b, x, W), stats), h->gemm_variant, s));
    if (stats) {
        CK(launch_gemm_f16(gemm_ln(gemm(x, W, k.fc_wf, W, hid, 4 * W, M, 4 * W, W, EPI_LN_BIAS_QGELU), stats, slots, k.fc_g, k.fc_bf), h->gemm_variant + (h->gelu_exact ? 0 : 100), s));
    } else {
        CK(launch_layernorm(x, y, k.ln2_g, k.ln2_b, M, W, W, 0, s));
        CK(launch_gemm_f16(gemm(y, W, k.fc_w, W, hid, 4 * W, M, 4 * W, W, EPI_BIAS_QGELU, k.fc_b), h->gemm_variant + (h->gelu_exact ? 0 : 100), s));
    }
    CK(launch_gemm_f16(gemm_stats(gemm(hid, 4 * W, k.pj_w, 4 * W, x, W, M, W, 4 * W, EPI_BIAS_RES, k.pj_b, x, W), stats), h->gemm_variant, s));
    return 0;
}

// Same block in fp32 (ResidualAttentionBlockWithDropout in eval mode, clip/model.py:248-251).
int run_block_f32(ovmr_handle* h, const Block& k, float* x, float* y, float* qkv, float* hid,
                  int nseq, int L, int W, hipStream_t s) {
    const int M = nseq * L, H = W / 64;
    CK(launch_layernorm(x, y, k.ln1_g, k.ln1_b, M, W, W, 1, s));
    CK(launch_gemm_f32(gemm(y, W, k.in_w, W, qkv, 3 * W, M, 3 * W, W, EPI_BIAS, k.in_b), s));
    CK(launch_attention_f32(qkv, y, nseq, L, H, s));
    CK(launch_gemm_f32(gemm(y, W, k.out_w, W, x, W, M, W, W, EPI_BIAS_RES, k.out_b, x, W), s));
    CK(launch_layernorm(x, y, k.ln2_g, k.ln2_b, M, W, W, 1, s));
    CK(launch_gemm_f32(gemm(y, W, k.fc_w, W, hid, 4 * W, M, 4 * W, W, EPI_BIAS_QGELU, k.fc_b), s));
    CK(launch_gemm_f32(gemm(hid, 4 * W, k.pj_w, 4 * W, x, W, M, W, 4 * W, EPI_BIAS_RES, k.pj_b, x, W), s));
    return 0;
}

int normalize_rows(ovmr_handle* h, half_t* x, int rows, int D, int mode, hipStream_t s) {
    for (int i = 0; i < mode; ++i) CK(launch_l2norm_f16(x, rows, D, s));
    return 0;
}

// One group of prompts of a text-tower pass: N sequences truncated to Ls tokens, read-out row `index` per sequence.
struct TextGroup {
    const half_t* prompts = nullptr;   // [N, context_length, W] embedded prompts, or
    const int64_t* ids = nullptr;      // [N, context_length] token ids (read-out row = argmax, clip/model.py:831)
    const int* index = nullptr;        // embedded prompts: [N]
    int N = 0, Ls = 0, normalize = 0;
    half_t* out = nullptr;             // [N, embed_dim]
};

size_t text_groups_ws_bytes(const ovmr_handle* h, size_t M, size_t N) {
    const size_t W = h->d.transformer_width;
    return align_up(M * W * 2) * 2 + align_up(M * 3 * W * 2) + align_up(M * 4 * W * 2) + align_up(N * W * 2) + align_up(N * 4) +
           align_up(M * ((W + 255) / 256) * 8);
}

// Text tower (clip/model.py:824-831) over all groups in ONE pass: embedding per group, the blocks' GEMMs over all token rows,
// causal attention with each group's own length (truncation to the last needed row is exact under the causal mask; one launch for
// all groups of at most 32 tokens, attention_short.hip, else group by group),
// read-out row gather, ln_final and projection per group.  The caller has checked that the rows fit the workspace.
int run_text_groups(ovmr_handle* h, const std::vector<TextGroup>& gs, hipStream_t s) {
    const ovmr_model_desc& d = h->d;
    const int W = d.transformer_width, Lc = d.context_length, E = d.embed_dim;
    size_t M = 0, N = 0;
    std::vector<SeqGroup> seq;
    for (auto& g : gs) { seq.push_back({g.N, g.Ls, (long)M}); M += (size_t)g.N * g.Ls; N += g.N; }
    Carver c(h->ws);
    half_t* x = c.take<half_t>(M * W);
    half_t* y = c.take<half_t>(M * W);
    half_t* qkv = c.take<half_t>(M * 3 * W);
    half_t* hid = c.take<half_t>(M * 4 * W);
    half_t* rows = c.take<half_t>(N * W);
    int* index = c.take<int>(N);
    float* stats_buf = c.take<float>(M * ((W + 255) / 256) * 2);
    size_t n0 = 0;
    for (size_t i = 0; i < gs.size(); ++i) {
        const TextGroup& g = gs[i];
        half_t* xg = x + seq[i].row0 * W;
        if (g.ids) CK(launch_text_embed_ids(g.ids, Lc, h->tok_emb, h->pos16_txt, xg, index + n0, g.N, Lc, g.Ls, W, s));
        else CK(launch_text_add_pos(g.prompts, Lc, h->pos16_txt, xg, g.N, g.Ls, W, s));
        n0 += g.N;
    }
    float* stats = !h->txt.empty() && can_fold_ln(h, h->txt[0], (int)M, W) ? stats_buf : nullptr;
    if (stats) CK(launch_row_stats(x, stats, (int)M, W, W / 256, s));
    for (auto& k : h->txt) CK(run_block_f16(h, k, x, y, qkv, hid, 0, 0, W, 1, s, stats, &seq));
    n0 = 0;
    for (size_t i = 0; i < gs.size(); ++i) {
        const TextGroup& g = gs[i];
        CK(launch_gather_rows_f16(x + seq[i].row0 * W, g.ids ? index + n0 : g.index, rows + n0 * W, g.N, g.Ls, W, s));
        n0 += g.N;
    }
    CK(launch_layernorm(rows, rows, h->ln_final_g, h->ln_final_b, (int)N, W, W, 0, s));
    n0 = 0;
    for (auto& g : gs) {
        CK(launch_gemm_f16(gemm(rows + n0 * W, W, h->textproj_t, W, g.out, E, g.N, E, W, EPI_NONE), h->gemm_variant, s));
        CK(normalize_rows(h, g.out, g.N, E, g.normalize, s));
        n0 += g.N;
    }
    return 0;
}

// One group, any size: chunks of max_prompts sequences.
int run_text_group_chunked(ovmr_handle* h, const TextGroup& g, hipStream_t s) {
    const int Lc = h->d.context_length, W = h->d.transformer_width, E = h->d.embed_dim;
    for (int n0 = 0; n0 < g.N; n0 += h->max_prompts) {
        TextGroup c = g;
        c.N = std::min(h->max_prompts, g.N - n0);
        if (g.ids) c.ids = g.ids + (size_t)n0 * Lc;
        else { c.prompts = g.prompts + (size_t)n0 * Lc * W; c.index = g.index + n0; }
        c.out = g.out + (size_t)n0 * E;
        CK(run_text_groups(h, {c}, s));
    }
    return 0;
}

// ovmr_fused_logits: the one-launch head (head_fused.hip) or scale + GEMMs + softmax.  Up to 256 query rows at any class count, up to 512
// rows x 2048 classes: every 64-row tile re-reads the classifier matrices, so beyond that the GEMM path's 256-row tiles win
// (tools/head_bench.py); option fused_head = 2 takes the one-launch kernel at any size, 0 never.  The two sum K in different orders: a logit
// may land on the neighbouring fp16 value, so callers that promise bit-equal rows for differently batched calls ask ovmr_head_plan.
bool head_takes_one_launch(const ovmr_handle* h, int B, int C) {
    return h->fused_head && (B <= 256 || (B <= 512 && C <= 2048) || h->fused_head == 2) && head_fused_ws_bytes(B, C) <= h->ws_bytes;
}

int check_text_call(ovmr_handle* h, int seq_len, int normalize) {
    if (normalize < 0 || normalize > 2) return OVMR_E_ARG;
    if (!h->finalized) return fail(h, OVMR_E_STATE, "ovmr_finalize() has not been called");
    if (seq_len < 1 || seq_len > h->d.context_length) return fail(h, OVMR_E_ARG, "seq_len %d outside [1,%d]", seq_len, h->d.context_length);
    return 0;
}

}  // namespace

extern "C" {

const char* ovmr_version(void) { return "ovmr_hip 0.1 (gfx950)"; }

int ovmr_create(const ovmr_model_desc* d, ovmr_handle** out) {
    if (!d || !out) return OVMR_E_ARG;
    if (d->vision_width % 64 || d->transformer_width % 64 || d->embed_dim % 64 || d->vision_patch_size <= 0 ||
        d->image_resolution % d->vision_patch_size || d->n_ctx < 1 || d->context_length < 4 + d->n_ctx)
        return OVMR_E_SHAPE;
    ovmr_handle* h = new ovmr_handle();
    h->d = *d;
    h->G = d->image_resolution / d->vision_patch_size;
    h->L = h->G * h->G + 1;
    h->Kpad = (3 * d->vision_patch_size * d->vision_patch_size + 63) / 64 * 64;
    const size_t sync_bytes = (size_t)head_fused_sync_ints() * sizeof(int);
    if (hipMalloc((void**)&h->head_sync, sync_bytes) != hipSuccess || hipMemset(h->head_sync, 0, sync_bytes) != hipSuccess) {
        delete h;
        return OVMR_E_NOMEM;
    }
    h->owned.push_back(h->head_sync);
    *out = h;
    return 0;
}

void ovmr_destroy(ovmr_handle* h) {
    if (!h) return;
    for (void* p : h->owned) (void)hipFree(p);
    for (void* p : h->derived) (void)hipFree(p);
    if (h->ws) (void)hipFree(h->ws);
    delete h;
}

const char* ovmr_last_error(const ovmr_handle* h) { return h ? h->err.c_str() : "null handle"; }

// Images per launch sequence of the image tower.  Every GEMM of a block runs 256-row tiles, one workgroup per CU at a time, so a launch
// costs WHOLE rounds of the CUs: with t = ceil(B * L / 256) row tiles the N = W launches (out_proj, K = W; c_proj, K = 4W) pay
// ceil(t * W/256 / CUs) rounds, in_proj ceil(3 t W/256 / CUs), c_fc ceil(4 t W/256 / CUs), a round's time being proportional to K.
// The chunk is the B <= max_images with the most images per unit of that cost (ViT-B/16, 256 CUs: 775 images = 597 tiles = 6.996 /
// 20.99 / 27.98 rounds where 768 pays 7 / 21 / 28 for 6.93 / 20.78 / 27.70; ViT-L/14@336px: 170 images = 384 tiles = 6.0 / 18.0 / 24.0)
// -- measured +1.5-2 % end to end.  Reserves of less than two rounds keep max_images.  No more than 600 row
// tiles per chunk: beyond that the A panels of a launch compete for the L2s (ViT-B/16: 998 and 1108 images per chunk ran 4-5 % slower
// than 775, c_fc at 0.384-0.388 of peak instead of 0.41; ViT-L/14@336px: 227 images = 512 tiles slower than 170 = 384).
static int pick_encode_chunk(int max_images, int L, int W, int n_cu) {
    const long nw = std::max(1, W / 256);
    auto tiles = [&](long b) { return (b * L + 255) / 256; };
    auto rounds = [&](long wg) { return (wg + n_cu - 1) / n_cu; };
    if (n_cu < 1 || tiles(max_images) * nw < 2L * n_cu) return max_images;
    auto cost = [&](long b) { const long t = tiles(b); return 5 * rounds(t * nw) + rounds(3 * t * nw) + rounds(4 * t * nw); };
    const int top = (int)std::min<long>(max_images, std::max<long>(1, 600L * 256 / L));
    int best = top;
    double best_rate = (double)top / (double)cost(top);
    for (int b = top - 1; b >= std::max(1, top * 3 / 4); --b) {
        const double r = (double)b / (double)cost(b);
        if (r > best_rate * 1.002) { best_rate = r; best = b; }      // (a smaller chunk has to buy more than launch overheads cost)
    }
    return best;
}

// Launch sequences of a batch of B images: one if it fits the reserve; else chunks of enc_chunk images, a remainder of at most
// enc_fold images (less than one round of tiles on the narrowest grid: as its own sequence it would pay a whole round on every launch
// PLUS ~70 launches of latency, 2.1 ms for 25 ViT-B/16 images where the extra round inside the previous sequence costs 1.8) folded
// into the last full chunk.
static std::vector<int> encode_plan(const ovmr_handle* h, int B) {
    std::vector<int> plan;
    const int chunk = h->enc_chunk_forced > 0 ? h->enc_chunk : (B > h->max_images && h->enc_chunk > 0 ? h->enc_chunk : h->max_images);
    for (int b0 = 0; b0 < B;) {
        int Bc = std::min(chunk, B - b0);
        const int left = B - b0 - Bc;
        if (left > 0 && left <= h->enc_fold && Bc + left <= h->img_cap) Bc += left;
        plan.push_back(Bc);
        b0 += Bc;
    }
    return plan;
}

int ovmr_set_option(ovmr_handle* h, const char* key, int value) {
    if (!h || !key) return OVMR_E_ARG;
    if (!strcmp(key, "gemm")) h->gemm_variant = value;
    else if (!strcmp(key, "attn")) h->attn_variant = value;
    else if (!strcmp(key, "fuse_im2col")) h->fuse_im2col = value != 0;
    else if (!strcmp(key, "last_q_cls")) h->last_q_cls = value != 0;
    else if (!strcmp(key, "enc_chunk")) {
        h->enc_chunk_forced = value > 0 ? value : 0;
        if (h->finalized)
            h->enc_chunk = h->enc_chunk_forced > 0 ? std::min(h->enc_chunk_forced, h->max_images)
                                                   : pick_encode_chunk(h->max_images, h->L, (int)h->d.vision_width, h->n_cu);
    }
    else if (!strcmp(key, "ln_fold")) h->ln_fold = value;
    else if (!strcmp(key, "xval_fused")) h->xval_fused = value;
    else if (!strcmp(key, "fused_head")) h->fused_head = value;
    else if (!strcmp(key, "head_max_grid")) h->head_max_grid = value;
    else if (!strcmp(key, "gelu_exact")) h->gelu_exact = value;
#ifdef OVMR_EXPERIMENTS
    else if (!strcmp(key, "fuse_qkv_attn")) h->fuse_qkv_attn = value;
#endif
    else return fail(h, OVMR_E_NAME, "unknown option '%s'", key);
    return 0;
}

float ovmr_logit_scale(const ovmr_handle* h) { return h ? h->logit_scale_exp : 0.f; }

int ovmr_preprocess_u8(const void* u8_hwc, int B, int R, const float* mean3, const float* std3, void* out_f16, ovmr_stream stream) {
    if (B == 0) return 0;
    if (!u8_hwc || !out_f16 || !mean3 || !std3 || B < 0 || R <= 0) return OVMR_E_ARG;
    if (R % 8) return OVMR_E_SHAPE;
    return launch_preprocess_u8((const uint8_t*)u8_hwc, (half_t*)out_f16, B, R, mean3, std3, (hipStream_t)stream);
}

int ovmr_resize_crop_u8(const void* pixels, const ovmr_resize_job* jobs, int n, const int32_t* tables, void* tmp, int max_ny,
                        void* out_u8, int R, ovmr_stream stream) {
    if (n == 0) return 0;
    if (!pixels || !jobs || !tables || !out_u8 || n < 0 || R <= 0 || max_ny < 0 || (max_ny > 0 && !tmp)) return OVMR_E_ARG;
    return launch_resize_crop_u8((const uint8_t*)pixels, jobs, n, tables, (uint8_t*)tmp, max_ny, (uint8_t*)out_u8, R, (hipStream_t)stream);
}

int ovmr_set_weight(ovmr_handle* h, const char* name_c, const void* data, int dtype, int ndim,
                    const int64_t* shape, ovmr_stream stream) {
    if (!h || !name_c || !data || (dtype != OVMR_F16 && dtype != OVMR_F32) || ndim < 0 || ndim > 4) return OVMR_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    const std::string name(name_c);
    size_t elems = 1;
    for (int i = 0; i < ndim; ++i) elems *= (size_t)shape[i];
    static const char* suffixes[] = {"attn.in_proj_weight", "attn.in_proj_bias", "attn.out_proj.weight", "attn.out_proj.bias",
                                     "mlp.c_fc.weight", "mlp.c_fc.bias", "mlp.c_proj.weight", "mlp.c_proj.bias",
                                     "ln_1.weight", "ln_1.bias", "ln_2.weight", "ln_2.bias"};
    bool known = false;
    if (name.rfind("visual.transformer.resblocks.", 0) == 0 || name.rfind("transformer.resblocks.", 0) == 0 ||
        name.rfind("prompt_learner.aggregator.resblocks.", 0) == 0)
        for (const char* sf : suffixes) known |= ends_with(name, sf);
    static const char* singles[] = {"visual.class_embedding", "visual.positional_embedding", "visual.proj", "visual.conv1.weight",
                                    "visual.ln_pre.weight", "visual.ln_pre.bias", "visual.ln_post.weight", "visual.ln_post.bias",
                                    "token_embedding.weight", "positional_embedding", "ln_final.weight", "ln_final.bias",
                                    "text_projection", "logit_scale", "prompt_learner.cls_token"};
    for (const char* sn : singles) known |= (name == sn);
    if (!known) return fail(h, OVMR_E_NAME, "unknown weight name '%s'", name_c);

    if (name == "logit_scale") {
        if (elems != 1) return fail(h, OVMR_E_SHAPE, "logit_scale must be a scalar");
        float v = 0.f;
        if (dtype == OVMR_F32) {
            HIP_CHECK_RET(hipMemcpyAsync(&v, data, 4, hipMemcpyDeviceToHost, s));
            HIP_CHECK_RET(hipStreamSynchronize(s));
        } else {
            half_t hv;
            HIP_CHECK_RET(hipMemcpyAsync(&hv, data, 2, hipMemcpyDeviceToHost, s));
            HIP_CHECK_RET(hipStreamSynchronize(s));
            v = (float)hv;
        }
        h->logit_scale_exp = expf(v);     // logit_scale.exp(), trainers/mm_classifier_one_prompt.py:238,296
        h->have_logit_scale = true;
        return 0;
    }
    Buf& b = h->w[name];
    b.f32 = stored_as_f32(name) ? 1 : 0;
    if (!b.p || b.elems != elems) {
        b.p = dev_alloc(h, elems * (b.f32 ? 4 : 2));
        if (!b.p) return fail(h, OVMR_E_NOMEM, "hipMalloc failed for '%s'", name_c);
    }
    b.elems = elems;
    b.shape.assign(shape, shape + ndim);
    h->finalized = false;
    return launch_cast(data, dtype == OVMR_F32, b.p, b.f32, (long)elems, s);
}

int ovmr_finalize(ovmr_handle* h, int max_images, int max_prompts, int max_classes, ovmr_stream stream) {
    if (!h || max_images < 1 || max_prompts < 1 || max_classes < 1) return OVMR_E_ARG;
    hipStream_t s = (hipStream_t)stream;
    const ovmr_model_desc& d = h->d;
    const size_t W = d.vision_width, T = d.transformer_width, E = d.embed_dim, P = d.vision_patch_size;
    if (!h->have_logit_scale) return fail(h, OVMR_E_STATE, "weight 'logit_scale' was never set");

    if (int rc = bind_blocks(h, "visual.transformer.resblocks.", d.vision_layers, (int)W, h->vis)) return rc;
    if (int rc = bind_blocks(h, "transformer.resblocks.", d.transformer_layers, (int)T, h->txt)) return rc;
    if (int rc = bind_blocks(h, "prompt_learner.aggregator.resblocks.", d.agg_layers, (int)E, h->agg)) return rc;

    const Buf *conv, *cls, *pos, *proj, *tpos, *tproj, *b;
    if (!(conv = find(h, "visual.conv1.weight", W * 3 * P * P))) return OVMR_E_STATE;
    if (!(cls = find(h, "visual.class_embedding", W))) return OVMR_E_STATE;
    if (!(pos = find(h, "visual.positional_embedding", (size_t)h->L * W))) return OVMR_E_STATE;
    if (!(proj = find(h, "visual.proj", W * E))) return OVMR_E_STATE;
    if (!(tpos = find(h, "positional_embedding", (size_t)d.context_length * T))) return OVMR_E_STATE;
    if (!(tproj = find(h, "text_projection", T * E))) return OVMR_E_STATE;
#define BINDF(field, nm, n) if (!(b = find(h, nm, n))) return OVMR_E_STATE; h->field = (float*)b->p;
    BINDF(ln_pre_g, "visual.ln_pre.weight", W) BINDF(ln_pre_b, "visual.ln_pre.bias", W)
    BINDF(ln_post_g, "visual.ln_post.weight", W) BINDF(ln_post_b, "visual.ln_post.bias", W)
    BINDF(ln_final_g, "ln_final.weight", T) BINDF(ln_final_b, "ln_final.bias", T)
    BINDF(tok_emb, "token_embedding.weight", (size_t)d.vocab_size * T)
    BINDF(cls_token, "prompt_learner.cls_token", (size_t)d.n_ctx * E)
#undef BINDF

    // derived layouts
    // derived layouts of a previous finalize are dropped first (the stream is idle: every finalize ends with a sync)
    HIP_CHECK_RET(hipStreamSynchronize(s));
    for (void* p : h->derived) (void)hipFree(p);
    h->derived.clear();
    auto dalloc = [&](size_t bytes) -> void* {
        void* p = nullptr;
        if (hipMalloc(&p, bytes ? bytes : 16) != hipSuccess) return nullptr;
        h->derived.push_back(p);
        return p;
    };
    h->conv_w = (half_t*)dalloc(W * h->Kpad * 2);
    h->pos16_vis = (half_t*)dalloc((size_t)h->L * W * 2);
    h->cls_pos16 = (half_t*)dalloc(W * 2);
    h->proj_t = (half_t*)dalloc(E * W * 2);
    h->pos16_txt = (half_t*)dalloc((size_t)d.context_length * T * 2);
    h->textproj_t = (half_t*)dalloc(E * T * 2);
    half_t* cls16 = (half_t*)dalloc(W * 2);
    if (!h->conv_w || !h->pos16_vis || !h->cls_pos16 || !h->proj_t || !h->pos16_txt || !h->textproj_t || !cls16)
        return fail(h, OVMR_E_NOMEM, "hipMalloc failed for derived weights");
    CK(launch_pad_rows_f16((const half_t*)conv->p, h->conv_w, (int)W, (int)(3 * P * P), h->Kpad, s));
    CK(launch_cast(pos->p, 1, h->pos16_vis, 0, (long)h->L * W, s));            // positional_embedding.to(fp16), clip/model.py:416
    CK(launch_cast(cls->p, 1, cls16, 0, (long)W, s));                          // class_embedding.to(fp16), :415
    CK(launch_add_f16(cls16, h->pos16_vis, h->cls_pos16, (long)W, s));
    CK(launch_transpose_to_f16(proj->p, 0, h->proj_t, (int)W, (int)E, s));     // x @ proj == x * proj^T^T
    CK(launch_cast(tpos->p, 1, h->pos16_txt, 0, (long)d.context_length * T, s));
    CK(launch_transpose_to_f16(tproj->p, 0, h->textproj_t, (int)T, (int)E, s));

    // LayerNorm folded into in_proj / c_fc of the two fp16 towers (common.h EPI_LN_BIAS)
    for (auto* tower : {&h->vis, &h->txt}) {
        const size_t Wd = tower == &h->vis ? W : T;
        if (Wd % 256) continue;                                                  // the statistics come in 256-column slots
        for (Block& k : *tower) {
            k.in_wf = (half_t*)dalloc(3 * Wd * Wd * 2); k.in_g = (float*)dalloc(3 * Wd * 4); k.in_bf = (float*)dalloc(3 * Wd * 4);
            k.fc_wf = (half_t*)dalloc(4 * Wd * Wd * 2); k.fc_g = (float*)dalloc(4 * Wd * 4); k.fc_bf = (float*)dalloc(4 * Wd * 4);
            if (!k.in_wf || !k.in_g || !k.in_bf || !k.fc_wf || !k.fc_g || !k.fc_bf)
                return fail(h, OVMR_E_NOMEM, "hipMalloc failed for LayerNorm-folded weights");
            CK(launch_fold_ln((const half_t*)k.in_w, k.ln1_g, k.ln1_b, (const half_t*)k.in_b, k.in_wf, k.in_g, k.in_bf, (int)(3 * Wd), (int)Wd, s));
            CK(launch_fold_ln((const half_t*)k.fc_w, k.ln2_g, k.ln2_b, (const half_t*)k.fc_b, k.fc_wf, k.fc_g, k.fc_bf, (int)(4 * Wd), (int)Wd, s));
        }
    }

    // workspace: one arena, re-carved by each entry point (calls on one handle are stream ordered)
    h->max_images = max_images; h->max_prompts = max_prompts; h->max_classes = max_classes;
    {
        int dev = 0;
        HIP_CHECK_RET(hipGetDevice(&dev));
        HIP_CHECK_RET(hipDeviceGetAttribute(&h->n_cu, hipDeviceAttributeMultiprocessorCount, dev));
        h->enc_chunk = h->enc_chunk_forced > 0 ? std::min(h->enc_chunk_forced, max_images) : pick_encode_chunk(max_images, h->L, (int)d.vision_width, h->n_cu);
    }
    {
        // fold slack: one round of the narrowest grid (N = W: W/256 column tiles of 256 rows), for reserves of two rounds or more
        const long nw = std::max<long>(1, (long)W / 256), round_images = (long)h->n_cu * 256 / (nw * h->L);
        const bool big = ((long)max_images * h->L + 255) / 256 * nw >= 2L * h->n_cu;
        h->enc_fold = big ? (int)std::min<long>(round_images, max_images / 4) : 0;
        h->img_cap = max_images + h->enc_fold;
    }
    h->agg_rows_cap = (long)max_classes * (d.n_ctx + 32);
    h->logit_elems_cap = 32L << 20;
    size_t need = image_ws_bytes(h, h->img_cap);
    need = std::max(need, text_ws_bytes(h, max_prompts));
    need = std::max(need, agg_ws_bytes(h, h->agg_rows_cap));
    need = std::max(need, align_up((size_t)h->logit_elems_cap * 2) * 3 + align_up((size_t)65536 * E * 2));
    if (need > h->ws_bytes) {
        if (h->ws) (void)hipFree(h->ws);
        h->ws = nullptr;
        if (hipMalloc((void**)&h->ws, need) != hipSuccess) return fail(h, OVMR_E_NOMEM, "workspace of %zu bytes", need);
        h->ws_bytes = need;
    }
    HIP_CHECK_RET(hipStreamSynchronize(s));
    h->finalized = true;
    return 0;
}

int ovmr_encode_image(ovmr_handle* h, const void* image, int image_dtype, int B, void* out_f16, int normalize,
                      ovmr_stream stream) {
    if (h && B == 0) return 0;   // empty batch: torch hands out a null data_ptr
    if (!h || !image || !out_f16 || B < 0 || (image_dtype != OVMR_F16 && image_dtype != OVMR_F32)) return OVMR_E_ARG;
    if (!h->finalized) return fail(h, OVMR_E_STATE, "ovmr_finalize() has not been called");
    hipStream_t s = (hipStream_t)stream;
    const ovmr_model_desc& d = h->d;
    const int W = d.vision_width, R = d.image_resolution, L = h->L, G2 = h->G * h->G, E = d.embed_dim;
    const size_t px = (size_t)3 * R * R * (image_dtype == OVMR_F32 ? 4 : 2);
    // a batch that fits the workspace is ONE launch sequence; a larger one is split into chunks of enc_chunk images (pick_encode_chunk),
    // unless the option pins the chunk; a sub-round remainder joins the last chunk (encode_plan)
    int b0 = 0;
    for (const int Bc : encode_plan(h, B)) {
        const int M = Bc * L;
        Carver c(h->ws);
        half_t* col = c.take<half_t>((size_t)Bc * G2 * h->Kpad);
        half_t* x = c.take<half_t>((size_t)M * W);
        half_t* y = c.take<half_t>((size_t)M * W);
        half_t* qkv = c.take<half_t>((size_t)M * 3 * W);
        half_t* hid = c.take<half_t>((size_t)M * 4 * W);
        half_t* rows = c.take<half_t>((size_t)Bc * W);
        float* stats_buf = c.take<float>((size_t)M * ((W + 255) / 256) * 2);
        half_t* out = (half_t*)out_f16 + (size_t)b0 * E;
        // K1/K2: conv1 as GEMM over patches, positional add in the epilogue, CLS row, ln_pre
        // fp16 images with 16 x 16 patches: the GEMM's K loop gathers the patch rows from the image itself (gemm_f16_v5.hip, a.im2col_R);
        // otherwise (fp32 images, other patch sizes, a handful of rows, the 128 x 128 kernel) an im2col pass writes them out first
        const bool fused_patches = h->fuse_im2col && image_dtype == OVMR_F16 && d.vision_patch_size == 16 && h->Kpad == 768 && (R & 15) == 0 &&
                                   Bc * G2 >= 256 && W >= 128 && h->gemm_variant >= 1 && ((uintptr_t)image & 15) == 0;
        GemmArgs pe = gemm(col, h->Kpad, h->conv_w, h->Kpad, x, W, Bc * G2, W, h->Kpad, EPI_PATCH);
        if (fused_patches) {
            pe.A = (const char*)image + (size_t)b0 * px;
            pe.im2col_R = R;
        } else
            CK(launch_im2col((const char*)image + (size_t)b0 * px, image_dtype == OVMR_F32, col, Bc, R, d.vision_patch_size, h->Kpad, s));
        pe.pos = h->pos16_vis; pe.rows_in = G2; pe.rows_out = L;
        CK(launch_gemm_f16(pe, h->gemm_variant, s));
        CK(launch_fill_cls(x, h->cls_pos16, Bc, L, W, s));
        CK(launch_layernorm(x, x, h->ln_pre_g, h->ln_pre_b, M, W, W, 0, s));
        float* stats = can_fold_ln(h, h->vis[0], M, W) ? stats_buf : nullptr;
        if (stats) CK(launch_row_stats(x, stats, M, W, W / 256, s));
        for (size_t li = 0; li + 1 < h->vis.size(); ++li) CK(run_block_f16(h, h->vis[li], x, y, qkv, hid, Bc, L, W, 0, s, stats));
        // Last block: only the CLS row reaches ln_post (clip/model.py:423), so after the K/V projection of all
        // tokens, attention / out-proj / MLP run for the CLS query row only -- identical result, ~6 % fewer FLOPs.
        {
            const Block& k = h->vis.back();
            const int H = W / 64;
            half_t* hid_c = hid;                       // [Bc, 4W]
            half_t* yc = y;                            // [Bc, W]
            // ... and of the in-projection itself only K and V are needed for every token: with at least 256 images (a full row tile of
            // CLS rows) the Q third runs as its own launch over the CLS rows (A and C strided by a sequence; bit-identical values),
            // a third of this block's largest GEMM less
            const bool q_cls_only = Bc >= 256 && h->last_q_cls;
            const size_t WW = (size_t)W * W;
            if (stats) {
                if (q_cls_only) {
                    CK(launch_gemm_f16(gemm_ln(gemm(x, W, k.in_wf + WW, W, qkv + W, 3 * W, M, 2 * W, W, EPI_LN_BIAS), stats, W / 256, k.in_g + W, k.in_bf + W), h->gemm_variant, s));
                    GemmArgs q = gemm_ln(gemm(x, L * W, k.in_wf, W, qkv, L * 3 * W, Bc, W, W, EPI_LN_BIAS), stats, W / 256, k.in_g, k.in_bf);
                    q.ln_stride = L * (W / 256);
                    CK(launch_gemm_f16(q, h->gemm_variant, s));
                } else
                    CK(launch_gemm_f16(gemm_ln(gemm(x, W, k.in_wf, W, qkv, 3 * W, M, 3 * W, W, EPI_LN_BIAS), stats, W / 256, k.in_g, k.in_bf), h->gemm_variant, s));
            } else {
                CK(launch_layernorm(x, y, k.ln1_g, k.ln1_b, M, W, W, 0, s));
                if (q_cls_only) {
                    // (both launches take the kernel the all-token launch of last_q_cls = 0 would take -- the split-K kernel, 9, if
                    //  [M, 3W] is a latency-bound shape, else the tile kernels, 7 -- so that the K summation order, and with it every
                    //  bit, is the same either way)
                    const int v = h->gemm_variant != 8 ? h->gemm_variant : (gemm_f16_is_small(M, 3 * W) ? 9 : 7);
                    CK(launch_gemm_f16(gemm(y, W, (const half_t*)k.in_w + WW, W, qkv + W, 3 * W, M, 2 * W, W, EPI_BIAS, (const half_t*)k.in_b + W), v, s));
                    CK(launch_gemm_f16(gemm(y, L * W, k.in_w, W, qkv, L * 3 * W, Bc, W, W, EPI_BIAS, k.in_b), v, s));
                } else
                    CK(launch_gemm_f16(gemm(y, W, k.in_w, W, qkv, 3 * W, M, 3 * W, W, EPI_BIAS, k.in_b), h->gemm_variant, s));
            }
            CK(launch_attention_f16_q(qkv, yc, Bc, L, 1, H, 0, h->attn_variant, s));
            CK(launch_gemm_f16(gemm(yc, W, k.out_w, W, rows, W, Bc, W, W, EPI_BIAS_RES, k.out_b, x, L * W), h->gemm_variant, s));
            CK(launch_layernorm(rows, yc, k.ln2_g, k.ln2_b, Bc, W, W, 0, s));
            CK(launch_gemm_f16(gemm(yc, W, k.fc_w, W, hid_c, 4 * W, Bc, 4 * W, W, EPI_BIAS_QGELU, k.fc_b), h->gemm_variant + (h->gelu_exact ? 0 : 100), s));
            CK(launch_gemm_f16(gemm(hid_c, 4 * W, k.pj_w, 4 * W, rows, W, Bc, W, 4 * W, EPI_BIAS_RES, k.pj_b, rows, W), h->gemm_variant, s));
        }
        // K9: ln_post on the CLS rows, projection; K10: normalise
        CK(launch_layernorm(rows, rows, h->ln_post_g, h->ln_post_b, Bc, W, W, 0, s));
        CK(launch_gemm_f16(gemm(rows, W, h->proj_t, W, out, E, Bc, E, W, EPI_NONE), h->gemm_variant, s));
        CK(normalize_rows(h, out, Bc, E, normalize ? 1 : 0, s));
        b0 += Bc;
    }
    return 0;
}

int ovmr_encode_plan(const ovmr_handle* h, int B, int* sizes, int max_sizes) {
    if (!h || !h->finalized || B < 0) return -1;
    const std::vector<int> plan = encode_plan(h, B);
    for (size_t i = 0; i < plan.size() && (int)i < max_sizes; ++i) sizes[i] = plan[i];
    return (int)plan.size();
}

int ovmr_encode_text_embedded(ovmr_handle* h, const void* prompts_f16, const int32_t* index, int N, int seq_len,
                              void* out_f16, int normalize, ovmr_stream stream) {
    if (h && N == 0) return 0;
    if (!h || !prompts_f16 || !index || !out_f16 || N < 0) return OVMR_E_ARG;
    if (int rc = check_text_call(h, seq_len, normalize)) return rc;
    TextGroup g;
    g.prompts = (const half_t*)prompts_f16; g.index = index; g.N = N; g.Ls = seq_len; g.normalize = normalize; g.out = (half_t*)out_f16;
    return run_text_group_chunked(h, g, (hipStream_t)stream);
}

int ovmr_encode_text_ids(ovmr_handle* h, const int64_t* ids, int N, int seq_len, void* out_f16, int normalize,
                         ovmr_stream stream) {
    if (h && N == 0) return 0;
    if (!h || !ids || !out_f16 || N < 0) return OVMR_E_ARG;
    if (int rc = check_text_call(h, seq_len, normalize)) return rc;
    TextGroup g;
    g.ids = ids; g.N = N; g.Ls = seq_len; g.normalize = normalize; g.out = (half_t*)out_f16;
    return run_text_group_chunked(h, g, (hipStream_t)stream);
}

int ovmr_encode_text_groups(ovmr_handle* h, const ovmr_text_group* groups, int n_groups, ovmr_stream stream) {
    if (!h || (!groups && n_groups > 0) || n_groups < 0) return OVMR_E_ARG;
    std::vector<TextGroup> gs;
    size_t M = 0, N = 0;
    for (int i = 0; i < n_groups; ++i) {
        const ovmr_text_group& u = groups[i];
        if (u.n == 0) continue;
        if (u.n < 0 || !u.out_f16 || (!u.ids && (!u.prompts_f16 || !u.index)) || (u.ids && u.prompts_f16)) return OVMR_E_ARG;
        if (int rc = check_text_call(h, u.seq_len, u.normalize)) return rc;
        TextGroup g;
        g.prompts = (const half_t*)u.prompts_f16; g.ids = u.ids; g.index = u.index; g.N = u.n; g.Ls = u.seq_len;
        g.normalize = u.normalize; g.out = (half_t*)u.out_f16;
        gs.push_back(g);
        M += (size_t)u.n * u.seq_len; N += u.n;
    }
    if (gs.empty()) return 0;
    hipStream_t s = (hipStream_t)stream;
    if (text_groups_ws_bytes(h, M, N) <= h->ws_bytes) return run_text_groups(h, gs, s);      // one pass over all groups
    for (auto& g : gs) CK(run_text_group_chunked(h, g, s));                                    // too many rows for the workspace
    return 0;
}

int ovmr_embed_tokens(ovmr_handle* h, const int64_t* ids, int N, int L, void* out_f16, ovmr_stream stream) {
    if (h && N == 0) return 0;
    if (!h || !ids || !out_f16 || N < 0 || L < 1) return OVMR_E_ARG;
    if (!h->finalized) return fail(h, OVMR_E_STATE, "ovmr_finalize() has not been called");
    return launch_embed_gather(ids, h->tok_emb, (half_t*)out_f16, (long)N * L, h->d.transformer_width, (hipStream_t)stream);
}

int ovmr_generate_tokens(ovmr_handle* h, const void* feats_f16, int Cb, int S, float* tokens_f32, ovmr_stream stream) {
    if (h && Cb == 0) return 0;
    if (!h || !feats_f16 || !tokens_f32 || Cb < 0 || S < 1) return OVMR_E_ARG;
    if (!h->finalized) return fail(h, OVMR_E_STATE, "ovmr_finalize() has not been called");
    const ovmr_model_desc& d = h->d;
    const int D = d.embed_dim, La = d.n_ctx + S;
    if (La > 128) return fail(h, OVMR_E_SHAPE, "n_ctx + shots = %d exceeds 128", La);
    hipStream_t s = (hipStream_t)stream;
    const int chunk = (int)std::max(1L, h->agg_rows_cap / La);
    for (int c0 = 0; c0 < Cb; c0 += chunk) {
        const int Cc = std::min(chunk, Cb - c0);
        const size_t M = (size_t)Cc * La;
        Carver c(h->ws);
        float* x = c.take<float>(M * D);
        float* y = c.take<float>(M * D);
        float* qkv = c.take<float>(M * 3 * D);
        float* hid = c.take<float>(M * 4 * D);
        CK(launch_agg_input(h->cls_token, (const half_t*)feats_f16 + (size_t)c0 * S * D, x, Cc, S, d.n_ctx, D, s));
        for (auto& k : h->agg) CK(run_block_f32(h, k, x, y, qkv, hid, Cc, La, D, s));
        CK(launch_agg_output(x, tokens_f32 + (size_t)c0 * d.n_ctx * D, Cc, La, d.n_ctx, D, s));
    }
    return 0;
}

int ovmr_assemble_prompts(ovmr_handle* h, const void* base_f16, const int64_t* labels, const float* tokens_f32, int Cb,
                          void* out_f16, ovmr_stream stream) {
    if (h && Cb == 0) return 0;
    if (!h || !base_f16 || !tokens_f32 || !out_f16 || Cb < 0) return OVMR_E_ARG;
    if (!h->finalized) return fail(h, OVMR_E_STATE, "ovmr_finalize() has not been called");
    if (h->d.embed_dim != h->d.transformer_width)
        return fail(h, OVMR_E_SHAPE, "visual tokens (embed_dim %d) do not fit the text width %d", h->d.embed_dim, h->d.transformer_width);
    CK(launch_assemble_prompts((const half_t*)base_f16, labels, tokens_f32, (half_t*)out_f16, Cb, h->d.context_length,
                               h->d.n_ctx, h->d.transformer_width, (hipStream_t)stream));
    return 0;
}

int ovmr_xval_counts(ovmr_handle* h, const void* feats_f16, const int32_t* labels, int R, const void* clf_f16, int C,
                     int32_t* tp, int32_t* n_pred, ovmr_stream stream) {
    if (h && R == 0) return 0;
    if (!h || !feats_f16 || !labels || !clf_f16 || !tp || !n_pred || R < 0 || C < 1) return OVMR_E_ARG;
    if (!h->finalized) return fail(h, OVMR_E_STATE, "ovmr_finalize() has not been called");
    hipStream_t s = (hipStream_t)stream;
    const int D = h->d.embed_dim;
    Carver c(h->ws);
    half_t* logits = c.take<half_t>((size_t)h->logit_elems_cap);
    if (h->xval_fused && R >= 256 && C >= 128) {
        // K18 + K19 fused: the [R, C] logits never reach HBM.  The GEMM epilogue leaves one
        // (maximum, lowest column) pair per row and 256-column tile; a row kernel finishes the argmax and counts.
        const int tiles = (C + 255) / 256;
        float* partial = (float*)logits;                                  // 2 floats per (row, tile) in the logits workspace
        const long cap_rows = std::min<long>(std::max<long>(512, h->logit_elems_cap / (4L * tiles)), 1L << 22);
        if (cap_rows * tiles * 8 > h->logit_elems_cap * 2) return fail(h, OVMR_E_SHAPE, "class count %d too large for the logits workspace", C);
        const long n_chunks = (R + cap_rows - 1) / cap_rows;              // balanced chunks: each has > cap_rows / 2 >= 256 rows
        for (long i = 0, r0 = 0; i < n_chunks; ++i) {
            const int Rc = (int)(R / n_chunks + (i < R % n_chunks ? 1 : 0));
            GemmArgs g = gemm((const half_t*)feats_f16 + (size_t)r0 * D, D, clf_f16, D, nullptr, C, Rc, C, D, EPI_SCALE_ARGMAX);
            g.scale = h->logit_scale_exp;
            g.argmax_out = partial;
            CK(launch_gemm_f16(g, h->gemm_variant, s));
            CK(launch_argmax_reduce(partial, tiles, labels + r0, Rc, C, tp, n_pred, s));
            r0 += Rc;
        }
        return 0;
    }
    const int chunk = (int)std::min<long>(std::max<long>(64, h->logit_elems_cap / C), 1L << 20);
    for (int r0 = 0; r0 < R; r0 += chunk) {
        const int Rc = std::min(chunk, R - r0);
        if ((long)Rc * C > h->logit_elems_cap) return fail(h, OVMR_E_SHAPE, "class count %d too large for the logits workspace", C);
        GemmArgs g = gemm((const half_t*)feats_f16 + (size_t)r0 * D, D, clf_f16, D, logits, C, Rc, C, D, EPI_SCALE);
        g.scale = h->logit_scale_exp;
        CK(launch_gemm_f16(g, h->gemm_variant, s));
        CK(launch_argmax_counts(logits, C, labels + r0, Rc, C, tp, n_pred, s));
    }
    return 0;
}

int ovmr_fusion_weights(ovmr_handle* h, const int32_t* counts, const int32_t* n_label, int C, float tau, float* out_f32,
                        ovmr_stream stream) {
    if (!h || !counts || !n_label || !out_f32 || C < 1) return OVMR_E_ARG;
    CK(launch_fusion_weights(counts, n_label, C, tau, out_f32, (hipStream_t)stream));
    return 0;
}

int ovmr_head_plan(const ovmr_handle* h, int B, int C) {
    if (!h || !h->finalized || B < 0 || C < 1) return -1;
    return head_takes_one_launch(h, B, C) && h->d.embed_dim % 64 == 0 ? 1 : 0;
}

int ovmr_pack_rows(const void* mm, const void* v, const void* t, const void* tokens, const int64_t* labels, int n, int D, int n_ctx, int bound,
                   void* block, ovmr_stream stream) {
    if (bound == 0) return 0;
    if (!mm || !v || !t || !tokens || !block || (n > 0 && !labels) || n < 0 || bound < n || D < 2 || (D & 1) || n_ctx < 1) return OVMR_E_ARG;
    return launch_pack_rows((const half_t*)mm, (const half_t*)v, (const half_t*)t, (const half_t*)tokens, labels, n, D, n_ctx, bound,
                            (half_t*)block, (hipStream_t)stream);
}

int ovmr_unpack_rows(const void* gathered, int rows, int C, int D, int n_ctx, void* mm, void* v, void* t, void* tokens, int32_t* seen,
                     ovmr_stream stream) {
    if (rows == 0) return 0;
    if (!gathered || !mm || !v || !t || !tokens || !seen || rows < 0 || C < 1 || D < 2 || (D & 1) || n_ctx < 1) return OVMR_E_ARG;
    return launch_unpack_rows((const half_t*)gathered, rows, C, D, n_ctx, (half_t*)mm, (half_t*)v, (half_t*)t, (half_t*)tokens, seen,
                              (hipStream_t)stream);
}

int ovmr_eval_counts(const void* outputs, int dtype, long ld, const int64_t* labels, int B, int C, int32_t* counts, ovmr_stream stream) {
    if (B == 0) return 0;
    if (!outputs || !labels || !counts || B < 0 || C < 1 || ld < C || (dtype != OVMR_F16 && dtype != OVMR_F32)) return OVMR_E_ARG;
    return launch_eval_counts(outputs, dtype == OVMR_F32, ld, labels, B, C, counts, (hipStream_t)stream);
}

int ovmr_fused_logits(ovmr_handle* h, const void* feats_f16, int B, const void* mm, const void* v, const void* t,
                      const float* w, int C, int mode, float* out_f32, ovmr_stream stream) {
    if (h && B == 0) return 0;
    if (!h || !feats_f16 || !out_f32 || B < 0 || C < 1) return OVMR_E_ARG;
    if (!h->finalized) return fail(h, OVMR_E_STATE, "ovmr_finalize() has not been called");
    hipStream_t s = (hipStream_t)stream;
    const int D = h->d.embed_dim;
    const void* clf[3];
    int n_mod;
    switch (mode) {
        case OVMR_MODE_FUSION: clf[0] = mm; clf[1] = v; clf[2] = t; n_mod = 3; if (!w) return OVMR_E_ARG; break;   // column order mm, v, t (:361)
        case OVMR_MODE_TEXT: clf[0] = t; n_mod = 1; break;
        case OVMR_MODE_VISION: clf[0] = v; n_mod = 1; break;
        case OVMR_MODE_MULTIMODAL: clf[0] = mm; n_mod = 1; break;
        default: return fail(h, OVMR_E_ARG, "unknown eval mode %d", mode);
    }
    for (int m = 0; m < n_mod; ++m) if (!clf[m]) return fail(h, OVMR_E_ARG, "classifier %d is NULL for mode %d", m, mode);
    if (head_takes_one_launch(h, B, C)) {
        // one launch: scaled features staged once, the (up to) three products, both rounding points, softmax and weighted sum (head_fused.hip)
        const half_t* cl[3] = {(const half_t*)clf[0], n_mod > 1 ? (const half_t*)clf[1] : nullptr, n_mod > 2 ? (const half_t*)clf[2] : nullptr};
        const int rc = launch_head_fused((const half_t*)feats_f16, B, D, h->logit_scale_exp, cl, n_mod, C, mode == OVMR_MODE_FUSION ? w : nullptr,
                                         out_f32, nullptr, h->ws, h->head_sync, h->n_cu, h->head_max_grid, s);
        if (rc != -100) { CK(rc); return 0; }
    }
    const int chunk = (int)std::min<long>(std::max<long>(1, h->logit_elems_cap / C), 65536);
    for (int b0 = 0; b0 < B; b0 += chunk) {
        const int Bc = std::min(chunk, B - b0);
        Carver c(h->ws);
        half_t* l[3] = {nullptr, nullptr, nullptr};
        for (int m = 0; m < 3; ++m) l[m] = c.take<half_t>((size_t)h->logit_elems_cap);
        half_t* sf = c.take<half_t>((size_t)65536 * D);
        // (logit_scale * image_features) is rounded to fp16 before the matmul (:358-360)
        CK(launch_scale_f16((const half_t*)feats_f16 + (size_t)b0 * D, sf, h->logit_scale_exp, (long)Bc * D, s));
        for (int m = 0; m < n_mod; ++m)
            CK(launch_gemm_f16(gemm(sf, D, clf[m], D, l[m], C, Bc, C, D, EPI_NONE), h->gemm_variant, s));
        CK(launch_fused_softmax(l[0], n_mod > 1 ? l[1] : nullptr, n_mod > 2 ? l[2] : nullptr,
                                mode == OVMR_MODE_FUSION ? w : nullptr, n_mod, out_f32 + (size_t)b0 * C, Bc, C, s));
    }
    return 0;
}

int ovmr_zeroshot_logits(ovmr_handle* h, const void* feats_f16, int B, const void* text_f16, int C, void* out_f16,
                         ovmr_stream stream) {
    if (h && B == 0) return 0;
    if (!h || !feats_f16 || !text_f16 || !out_f16 || B < 0 || C < 1) return OVMR_E_ARG;
    if (!h->finalized) return fail(h, OVMR_E_STATE, "ovmr_finalize() has not been called");
    hipStream_t s = (hipStream_t)stream;
    const int D = h->d.embed_dim;
    if (h->fused_head) {                           // scale, product and the fp16 rounding of the logits in one launch
        const half_t* cl[3] = {(const half_t*)text_f16, nullptr, nullptr};
        const int rc = launch_head_fused((const half_t*)feats_f16, B, D, h->logit_scale_exp, cl, 1, C, nullptr, nullptr, (half_t*)out_f16,
                                         nullptr, nullptr, h->n_cu, 0, s);
        if (rc != -100) { CK(rc); return 0; }
    }
    for (int b0 = 0; b0 < B; b0 += 65536) {
        const int Bc = std::min(65536, B - b0);
        Carver c(h->ws);
        half_t* sf = c.take<half_t>((size_t)65536 * D);
        CK(launch_scale_f16((const half_t*)feats_f16 + (size_t)b0 * D, sf, h->logit_scale_exp, (long)Bc * D, s));
        CK(launch_gemm_f16(gemm(sf, D, text_f16, D, (half_t*)out_f16 + (size_t)b0 * C, C, Bc, C, D, EPI_NONE), h->gemm_variant, s));
    }
    return 0;
}

// Closed-form FLOPs (2*MAC), SURVEY.md section 2.3: per layer 24*L*W^2 (QKV 6, out 2, MLP 16) + 4*L^2*W
static double tower_flops(double L, double W, double layers) { return layers * (24.0 * L * W * W + 4.0 * L * L * W); }

int ovmr_encode_chunk(const ovmr_handle* h) { return h && h->finalized ? (h->enc_chunk > 0 ? h->enc_chunk : h->max_images) : 0; }

double ovmr_flops_per_image(const ovmr_handle* h) {
    if (!h) return 0;
    const ovmr_model_desc& d = h->d;
    const double G2 = (double)h->G * h->G, W = d.vision_width, K = 3.0 * d.vision_patch_size * d.vision_patch_size;
    return 2.0 * G2 * K * W + tower_flops(h->L, W, d.vision_layers) + 2.0 * W * d.embed_dim;
}
double ovmr_flops_per_image_executed(const ovmr_handle* h) {
    if (!h) return 0;
    const ovmr_model_desc& d = h->d;
    const double L = h->L, W = d.vision_width;
    // last block as launched: K/V projection of all tokens (4 L W^2), then one query row: scores + PV (4 L W), out_proj (2 W^2),
    // MLP (16 W^2).  Q: for the CLS row alone (2 W^2) where the launch sequences hold >= 256 images and last_q_cls is set (what
    // ovmr_encode_image does for a reserve of that size), else for every token (2 L W^2)
    const bool q_cls = h->last_q_cls && (!h->finalized || std::min(h->max_images, h->enc_chunk > 0 ? h->enc_chunk : h->max_images) >= 256);
    const double last_full = 24.0 * L * W * W + 4.0 * L * L * W;
    const double last_run = 4.0 * L * W * W + 4.0 * L * W + 18.0 * W * W + (q_cls ? 2.0 * W * W : 2.0 * L * W * W);
    return ovmr_flops_per_image(h) - last_full + last_run;
}
double ovmr_flops_executed(const ovmr_handle* h, int B) {
    if (!h || !h->finalized || B <= 0) return 0;
    const ovmr_model_desc& d = h->d;
    const double L = h->L, W = d.vision_width;
    const double last_full = 24.0 * L * W * W + 4.0 * L * L * W;
    double total = 0;
    for (const int Bc : encode_plan(h, B)) {               // the launch sequences ovmr_encode_image runs for this batch
        const bool q_cls = h->last_q_cls && Bc >= 256;
        const double last_run = 4.0 * L * W * W + 4.0 * L * W + 18.0 * W * W + (q_cls ? 2.0 * W * W : 2.0 * L * W * W);
        total += Bc * (ovmr_flops_per_image(h) - last_full + last_run);
    }
    return total;
}
double ovmr_flops_per_prompt(const ovmr_handle* h, int seq_len) {
    if (!h) return 0;
    const ovmr_model_desc& d = h->d;
    return tower_flops(seq_len, d.transformer_width, d.transformer_layers) + 2.0 * d.transformer_width * d.embed_dim;
}

}  // extern "C"

// ---- unit-test hooks (tests/test_hip_kernels.py): one kernel per call, no handle state -----------
extern "C" {

int ovmr_debug_gemm(int f32, int variant, const void* A, const void* W, const void* bias, const void* res,
                    const void* pos, void* C, int M, int N, int K, int ldc, int epi, float scale,
                    int rows_in, int rows_out, ovmr_stream stream) {
    GemmArgs a = gemm(A, K, W, K, C, ldc, M, N, K, epi, bias, res, ldc);
#ifdef OVMR_EXPERIMENTS
    if (const char* e = getenv("OVMR_DEBUG_LDPAD")) {   // stride experiment (tools/gemm_bench.py --ldpad): operands have padded rows
        a.lda = K + atoi(e);
        a.ldw = K + atoi(e);
    }
    if (const char* e = getenv("OVMR_DEBUG_BLOCKED")) {   // timing-only layout experiment (operands are random anyway)
        a.a_blocked = atoi(e) & 1;
        a.w_blocked = (atoi(e) >> 1) & 1;
    }
#endif
    a.pos = pos; a.scale = scale; a.rows_in = rows_in; a.rows_out = rows_out;
    if (epi == EPI_LN_BIAS || epi == EPI_LN_BIAS_QGELU) {   // LN-folding epilogues: bias = ln_b fp32 [N], pos = ln_g fp32 [N], res = statistics fp32 [M][K/256][2]
        a.ln_b = (const float*)bias; a.ln_g = (const float*)pos; a.ln_stats = (const float*)res; a.ln_slots = K / 256;
        a.bias = nullptr; a.pos = nullptr; a.res = nullptr;
    }
    if (epi == EPI_SCALE_ARGMAX) {                          // fused row argmax: C receives fp32 [M][ceil(N/256)][2] = (max, column bits)
        a.argmax_out = (float*)C;
        a.C = nullptr;
    }
#ifdef OVMR_EXPERIMENTS
    if (variant == 58) {                                    // shader-clock stamps of the ping-pong K loop: pos = int64 [8 waves][4096] (tools/gemm_stamps.py)
        a.argmax_out = (float*)pos;
        a.pos = nullptr;
    }
#endif
    if (epi == EPI_BIAS_RES && pos) {                       // statistics epilogue: pos = fp32 [M][N/256][2] output
        a.stats_out = (float*)pos;
        a.pos = nullptr;
    }
    return f32 ? launch_gemm_f32(a, (hipStream_t)stream) : launch_gemm_f16(a, variant, (hipStream_t)stream);
}

// x1 = h(h(A1 W1^T + b1) + res) with the statistics epilogue, then C2 = h(LN(x1; gamma, beta) W2^T + b2) [QuickGELU] with
// the LayerNorm folded into the second GEMM.  Scratch is allocated here (debug / kernel-test hook only).
int ovmr_debug_lnfold(int variant, const void* A1, const void* W1, const void* b1, const void* res, int M, int D, int K1,
                      const void* W2, const float* gamma, const float* beta, const void* b2, int N2, int qgelu,
                      void* x1, void* C2, ovmr_stream stream) {
    hipStream_t s = (hipStream_t)stream;
    if (D % 256 || N2 % 64) return OVMR_E_ARG;
    const int slots = D / 256;
    float *stats = nullptr, *g = nullptr, *bf = nullptr;
    half_t* wf = nullptr;
    int rc = 0;
    if (hipMalloc((void**)&stats, (size_t)M * slots * 8) != hipSuccess || hipMalloc((void**)&g, (size_t)N2 * 4) != hipSuccess ||
        hipMalloc((void**)&bf, (size_t)N2 * 4) != hipSuccess || hipMalloc((void**)&wf, (size_t)N2 * D * 2) != hipSuccess)
        rc = OVMR_E_NOMEM;
    if (!rc) rc = launch_fold_ln((const half_t*)W2, gamma, beta, (const half_t*)b2, wf, g, bf, N2, D, s);
    if (!rc && A1) {
        GemmArgs a = gemm(A1, K1, W1, K1, x1, D, M, D, K1, EPI_BIAS_RES, b1, res, D);
        a.stats_out = stats;
        rc = launch_gemm_f16(a, variant, s);
    } else if (!rc) {
        rc = launch_row_stats((const half_t*)x1, stats, M, D, slots, s);     // A1 == NULL: x1 is an input
    }
    if (!rc) {
        GemmArgs a = gemm(x1, D, wf, D, C2, N2, M, N2, D, qgelu ? EPI_LN_BIAS_QGELU : EPI_LN_BIAS);
        a.ln_stats = stats; a.ln_slots = slots; a.ln_g = g; a.ln_b = bf;
        rc = launch_gemm_f16(a, variant, s);
    }
    (void)hipStreamSynchronize(s);
    (void)hipFree(stats); (void)hipFree(g); (void)hipFree(bf); (void)hipFree(wf);
    return rc;
}

int ovmr_debug_layernorm(int f32, const void* x, void* y, const float* g, const float* b, int rows, int D,
                         long in_stride, ovmr_stream stream) {
    return launch_layernorm(x, y, g, b, rows, D, in_stride, f32, (hipStream_t)stream);
}

int ovmr_debug_attention(int f32, int variant, const void* qkv, void* out, int B, int L, int H, int causal,
                         ovmr_stream stream) {
    return f32 ? launch_attention_f32((const float*)qkv, (float*)out, B, L, H, (hipStream_t)stream)
               : launch_attention_f16((const half_t*)qkv, (half_t*)out, B, L, H, causal, variant, (hipStream_t)stream);
}

}  // extern "C"
