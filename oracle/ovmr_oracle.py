"""CPU oracle for the OVMR hot path -- TEST INFRASTRUCTURE ONLY.

A plain torch-CPU restatement of the reference's arithmetic for the one path this
repository accelerates (CLIP ViT image encoder, CLIP text transformer, OVMR visual-token
generator + multimodal classifier fusion).  Every function cites the reference file:line
it follows (paths relative to the upstream repository root).

Rules (tier framing, section 3 of the task statement):
  * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this
    module; the product package `ovmr_amd` never does and fails loudly without its HIP
    library;
  * parity is PINNED: tests/golden/*.npz were produced by tests/golden/gen_golden.py, which
    imports the real reference modules (clip/model.py and
    trainers/mm_classifier_one_prompt.py) in the build container and records their outputs
    on the seeded inputs of ovmr_amd/synth.py; tests/test_oracle_vs_golden.py checks this
    restatement against those vectors.  The one third-party dependency whose arithmetic is
    not under the reference tree, torcheval==0.0.7 `multiclass_f1_score(average=None)`
    (requirements.txt:16, call sites trainers/mm_classifier_one_prompt.py:268-270), is
    restated in `multiclass_f1_per_class` from its published algorithm and pinned by
    hand-computed known answers plus an sklearn cross-check in the tests.

dtype policy ("prec"):
  "fp16"  the reference's only working OVMR precision (configs/.../*.yaml:42): weights of
          Conv/Linear/MHA/proj are fp16, LayerNorm affine / embeddings stay fp32
          (clip/model.py:852-873), activations fp16, LayerNorm computed in fp32
          (clip/model.py:153-159), the aggregator runs fp32
          (trainers/mm_classifier_one_prompt.py:138-143 is built after convert_weights).
  "fp32"  everything fp32 -- the "ideal" answer, used to separate our error from the
          reference's own fp16 rounding.
"""
from __future__ import annotations

import math
from typing import Dict, Optional, Sequence, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor

_HALF_SUFFIXES = ("attn.in_proj_weight", "attn.in_proj_bias", "attn.out_proj.weight", "attn.out_proj.bias",
                  "mlp.c_fc.weight", "mlp.c_fc.bias", "mlp.c_proj.weight", "mlp.c_proj.bias")


def convert_weights(sd: Dict[str, Tensor], prec: str = "fp16") -> Dict[str, Tensor]:
    """clip/model.py:852-873 -- which tensors convert_weights() halves.

    Conv2d / Linear / MultiheadAttention weights+biases and `proj` / `text_projection`
    become fp16; LayerNorm, class/positional embeddings, token_embedding, logit_scale stay fp32.
    """
    out = {}
    for k, v in sd.items():
        v = torch.as_tensor(v).float()
        if prec == "fp16" and (k.endswith(_HALF_SUFFIXES) or k in ("visual.proj", "text_projection", "visual.conv1.weight")):
            v = v.half()
        out[k] = v
    return out


# ----------------------------------------------------------------------------- blocks
def layer_norm(x: Tensor, w: Tensor, b: Tensor, eps: float = 1e-5) -> Tensor:
    """clip/model.py:153-159: upcast to fp32, nn.LayerNorm (eps 1e-5), cast back."""
    return F.layer_norm(x.float(), (x.shape[-1],), w.float(), b.float(), eps).to(x.dtype)


def quick_gelu(x: Tensor) -> Tensor:
    """clip/model.py:162-164."""
    return x * torch.sigmoid(1.702 * x)


def build_attention_mask(n: int) -> Tensor:
    """clip/model.py:802-808: additive causal mask, -inf strictly above the diagonal."""
    return torch.full((n, n), float("-inf")).triu_(1)


def multi_head_attention(x: Tensor, in_w: Tensor, in_b: Tensor, out_w: Tensor, out_b: Tensor,
                         heads: int, mask: Optional[Tensor]) -> Tensor:
    """nn.MultiheadAttention as used at clip/model.py:171,184-188 (self attention, no dropout).

    x: [N, L, D] (batch-major here; the reference permutes to [L, N, D], the math is per
    sequence so the layout is immaterial).  in_proj rows are ordered q,k,v; head h owns
    channels h*hd:(h+1)*hd; scores are scaled by hd**-0.5; mask is additive.
    """
    N, L, D = x.shape
    hd = D // heads
    qkv = F.linear(x, in_w, in_b)                                    # [N, L, 3D]
    q, k, v = qkv.split(D, dim=-1)
    q = q.reshape(N, L, heads, hd).transpose(1, 2).float()
    k = k.reshape(N, L, heads, hd).transpose(1, 2).float()
    v = v.reshape(N, L, heads, hd).transpose(1, 2).float()
    s = (q @ k.transpose(-1, -2)) * (hd ** -0.5)
    if mask is not None:
        s = s + mask.float()
    p = s.softmax(dim=-1)
    o = (p @ v).to(x.dtype).transpose(1, 2).reshape(N, L, D)
    return F.linear(o, out_w, out_b)


def residual_block(x: Tensor, sd: Dict[str, Tensor], p: str, heads: int, mask: Optional[Tensor]) -> Tensor:
    """clip/model.py:191-194 (and :248-251 for the dropout variant in eval mode)."""
    a = multi_head_attention(layer_norm(x, sd[p + "ln_1.weight"], sd[p + "ln_1.bias"]),
                             sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"],
                             sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"], heads, mask)
    x = x + a
    h = layer_norm(x, sd[p + "ln_2.weight"], sd[p + "ln_2.bias"])
    h = quick_gelu(F.linear(h, sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"]))
    return x + F.linear(h, sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"])


def transformer(x: Tensor, sd: Dict[str, Tensor], prefix: str, heads: int, mask: Optional[Tensor],
                taps: Optional[list] = None) -> Tensor:
    """clip/model.py:261-269 / :341-350: nn.Sequential of residual blocks."""
    i = 0
    while f"{prefix}{i}.ln_1.weight" in sd:
        x = residual_block(x, sd, f"{prefix}{i}.", heads, mask)
        if taps is not None:
            taps.append(x)
        i += 1
    return x


# ----------------------------------------------------------------------------- encoders
def encode_image(image: Tensor, sd: Dict[str, Tensor], taps: Optional[dict] = None) -> Tensor:
    """VisionTransformer.forward, clip/model.py:411-428.

    image [B,3,R,R] (cast to the conv weight dtype as CLIP.encode_image does, :814) ->
    [B, embed_dim].  Token order: CLS first, then patches gy*G+gx.
    """
    w = sd["visual.conv1.weight"]
    dt = w.dtype
    x = F.conv2d(image.to(dt), w, stride=w.shape[-1])                   # :412
    x = x.reshape(x.shape[0], x.shape[1], -1).permute(0, 2, 1)         # :413-414
    cls = sd["visual.class_embedding"].to(dt) + torch.zeros(x.shape[0], 1, x.shape[-1], dtype=dt)
    x = torch.cat([cls, x], dim=1)                                      # :415
    x = x + sd["visual.positional_embedding"].to(dt)                    # :416
    if taps is not None:
        taps["tokens"] = x
    x = layer_norm(x, sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"])  # :417
    if taps is not None:
        taps["ln_pre"] = x
    heads = w.shape[0] // 64                                            # clip/model.py:745
    blocks = [] if taps is not None else None
    x = transformer(x, sd, "visual.transformer.resblocks.", heads, None, blocks)  # :419-421
    if taps is not None:
        taps["blocks"] = blocks
    x = layer_norm(x[:, 0, :], sd["visual.ln_post.weight"], sd["visual.ln_post.bias"])  # :423
    return x @ sd["visual.proj"]                                        # :425-426


def _text_tail(x: Tensor, index: Tensor, sd: Dict[str, Tensor]) -> Tensor:
    x = layer_norm(x, sd["ln_final.weight"], sd["ln_final.bias"])
    return x[torch.arange(x.shape[0]), index.long()] @ sd["text_projection"]


def text_encoder_forward(prompts: Tensor, eos_index: Tensor, sd: Dict[str, Tensor],
                         dtype: torch.dtype = torch.float16) -> Tensor:
    """TextEncoder.forward, trainers/mm_classifier_one_prompt.py:80-91.

    prompts [N,77,Dt] are already embedded; `eos_index` selects the row that is projected
    (for the vision prompt this is 1+n_ctx, i.e. the last visual token, not EOS: :165).
    The reference hard-codes dtype fp16 (:70); the fp32 oracle passes float32.
    """
    heads = sd["ln_final.weight"].shape[0] // 64                        # clip/model.py:925
    x = prompts.to(dtype) + sd["positional_embedding"].to(dtype)[:prompts.shape[1]]   # :81
    x = transformer(x, sd, "transformer.resblocks.", heads, build_attention_mask(x.shape[1]))
    return _text_tail(x, eos_index, sd)                                 # :85-89


def encode_text(ids: Tensor, sd: Dict[str, Tensor]) -> Tensor:
    """CLIP.encode_text, clip/model.py:820-833: embed ids, row at ids.argmax() (EOT) is projected."""
    dt = sd["text_projection"].dtype                                    # CLIP.dtype is the conv dtype (:810-812)
    heads = sd["ln_final.weight"].shape[0] // 64
    x = sd["token_embedding.weight"][ids.long()].to(dt)                 # :821
    x = x + sd["positional_embedding"].to(dt)                           # :823
    x = transformer(x, sd, "transformer.resblocks.", heads, build_attention_mask(x.shape[1]))
    return _text_tail(x, ids.argmax(dim=-1), sd)


def l2_normalize(x: Tensor) -> Tensor:
    """x / x.norm(dim=-1, keepdim=True), e.g. trainers/mm_classifier_one_prompt.py:244,307."""
    return x / x.norm(dim=-1, keepdim=True)


# ----------------------------------------------------------------------------- OVMR head
def zero_shot_classifier(tokenized_prompts: Tensor, sd: Dict[str, Tensor]) -> Tensor:
    """PromptLearner.__init__, trainers/mm_classifier_one_prompt.py:118-126.

    Per class: encode_text of its one prompt, mean over that single prompt, F.normalize.
    (The reference loops batch-1; rows are independent so one batch gives the same rows.)
    """
    t = encode_text(tokenized_prompts, sd)
    return F.normalize(t, dim=-1, p=2)


def prompt_embeddings(tokenized: Tensor, sd: Dict[str, Tensor], dtype=torch.float16) -> Tensor:
    """token_embedding(tokenized).type(fp16), trainers/mm_classifier_one_prompt.py:129-130."""
    return sd["token_embedding.weight"][tokenized.long()].to(dtype)


def update_prompts(prompt_tokens: Tensor, ins_tokens: Tensor, n_ctx: int) -> Tensor:
    """PromptLearner.update_prompts, trainers/mm_classifier_one_prompt.py:156-157."""
    return torch.cat([prompt_tokens[:, :2], ins_tokens.to(prompt_tokens.dtype), prompt_tokens[:, 2:-n_ctx]], dim=1)


def prompt_learner_forward(feats: Tensor, label: Tensor, ori_text_len: Tensor, prompt_tokens: Tensor,
                           visual_prompt_temp: Tensor, pl: Dict[str, Tensor], n_ctx: int):
    """PromptLearner.forward, trainers/mm_classifier_one_prompt.py:159-176.

    feats [Cb,S,D] (fp16 in the reference) are concatenated with the fp32 cls_token, which
    promotes the aggregator input to fp32 (:167-168); the aggregator weights are fp32.
    Returns (mm_prompts, mm_lens, v_prompts, v_lens, tokens[Cb,n_ctx,D] fp32).
    """
    Cb, S, D = feats.shape
    heads = D // 64                                                     # :141
    mm_lens = ori_text_len + n_ctx                                      # :163
    v_lens = torch.ones_like(ori_text_len, dtype=torch.int32) + n_ctx   # :165
    cls = pl["cls_token"].float().unsqueeze(0).repeat(Cb, 1, 1)         # :167 (batch-major here)
    agg_in = torch.cat([cls, feats.to(torch.promote_types(feats.dtype, cls.dtype))], dim=1)
    tokens = transformer(agg_in, pl, "aggregator.resblocks.", heads, None)[:, :n_ctx, :]   # :169
    mm_prompts = update_prompts(prompt_tokens[label.long()], tokens, n_ctx)               # :171
    v_prompts = update_prompts(visual_prompt_temp.repeat(Cb, 1, 1), tokens, n_ctx)        # :173
    return mm_prompts, mm_lens, v_prompts, v_lens, tokens


def get_mm_v_feats(mm_prompts: Tensor, mm_lens: Tensor, v_prompts: Tensor, v_lens: Tensor,
                   sd: Dict[str, Tensor], dtype=torch.float16) -> Tuple[Tensor, Tensor]:
    """CustomCLIP.get_mm_v_feats, trainers/mm_classifier_one_prompt.py:200-212 (lists of length 1):
    normalise, mean over one element, normalise again."""
    mm = l2_normalize(text_encoder_forward(mm_prompts, mm_lens, sd, dtype))
    v = l2_normalize(text_encoder_forward(v_prompts, v_lens, sd, dtype))
    mm = F.normalize(mm.unsqueeze(1).mean(dim=1), dim=-1, p=2)
    v = F.normalize(v.unsqueeze(1).mean(dim=1), dim=-1, p=2)
    return mm, v


def multiclass_f1_per_class(logits: Tensor, labels: Tensor, num_classes: int) -> Tensor:
    """torcheval==0.0.7 multiclass_f1_score(input, target, num_classes=C, average=None)
    as called at trainers/mm_classifier_one_prompt.py:268-270 (restated from the published
    algorithm; torcheval is not vendored in the reference):
      pred = input.argmax(dim=1); tp[c] = #(pred==c & target==c); n_label[c]; n_pred[c];
      precision = tp/n_pred; recall = tp/n_label; f1 = 2pr/(p+r); nan_to_num(f1) (NaN -> 0).
    """
    pred = logits.argmax(dim=1)
    labels = labels.long()
    tp = torch.bincount(labels[pred == labels], minlength=num_classes).float()
    n_label = torch.bincount(labels, minlength=num_classes).float()
    n_pred = torch.bincount(pred, minlength=num_classes).float()
    precision = tp / n_pred
    recall = tp / n_label
    return torch.nan_to_num(2 * precision * recall / (precision + recall))


def f1_from_counts(tp: Tensor, n_pred: Tensor, n_label: Tensor) -> Tensor:
    precision = tp.float() / n_pred.float()
    recall = tp.float() / n_label.float()
    return torch.nan_to_num(2 * precision * recall / (precision + recall))


def cross_validation_logits(eval_feats: Tensor, classifier: Tensor, logit_scale: Tensor) -> Tensor:
    """trainers/mm_classifier_one_prompt.py:263-265: logit_scale * einsum("bmc,pc->bmp").flatten(0,1).
    Row r = c*S + s; the einsum result keeps the feature dtype and the 0-dim fp32 scale does
    not promote it, so fp16 features give fp16 logits."""
    return (logit_scale * torch.einsum("bmc,pc->bmp", eval_feats, classifier)).flatten(0, 1)


def fusion_weights(eval_feats: Tensor, mm: Tensor, v: Tensor, t: Tensor, logit_scale: Tensor, tau: float):
    """trainers/mm_classifier_one_prompt.py:261-274: per-class F1 of the three classifiers on the
    exemplars themselves, softmax(tau * [f1_mm, f1_v, f1_t]) (column order mm, vision, text)."""
    C, S, _ = eval_feats.shape
    labels = torch.arange(C).reshape(-1, 1).repeat(1, S).flatten(0, 1)  # :261
    f1s = [multiclass_f1_per_class(cross_validation_logits(eval_feats, w, logit_scale), labels, C)
           for w in (mm, v, t)]
    ce = torch.stack(f1s, dim=-1).float()                                # :272
    return (tau * ce).softmax(dim=-1), ce                                # :273


def get_fusion_weight_coop(eval_feats: Tensor, mm: Tensor, v: Tensor, t: Tensor, logit_scale: Tensor):
    """CustomCLIP.get_fusion_weight of trainers/coop_mm_classifier.py:235-305, after its encode loop: the features are
    permuted to [S, C, out] (:277), the einsum runs over that layout and the result is permuted back before the row
    flatten (:285-287), labels are arange(C) repeated S times (:278), F1 per class (:295-301), softmax(10 * F1) (:304)."""
    C, S, _ = eval_feats.shape
    feats = eval_feats.permute(1, 0, 2)                                                  # :277
    labels = torch.arange(C).reshape(-1, 1).repeat(1, S).flatten(0, 1)                   # :278
    f1s = []
    for w in (mm, v, t):                                                                 # :285-287
        lg = (logit_scale * torch.einsum("bmc,pc->bmp", feats, w)).permute(1, 0, 2).flatten(0, 1)
        f1s.append(multiclass_f1_per_class(lg, labels, C).reshape(C))                   # :295-301
    ce = torch.cat([f.unsqueeze(-1) for f in f1s], dim=-1).float()                       # :303
    return (10 * ce).softmax(dim=-1)                                                     # :304


def forward_prompt(images_by_class: Tensor, labels: Tensor, tokenized_prompts: Tensor,
                   sd: Dict[str, Tensor], pl: Dict[str, Tensor], n_ctx: int, tau: float,
                   classes_per_batch: int, prec: str = "fp16", text_classifier: Optional[Tensor] = None,
                   image_features: Optional[Tensor] = None):
    """CustomCLIP.forward_prompt, trainers/mm_classifier_one_prompt.py:214-292 (hot loop A + K18-K20).

    images_by_class [C*S,3,R,R] with S consecutive rows per class, labels [C*S] (batch contract
    of Dassl's RandomClassSampler).  Returns a dict with the tensors the
    reference writes to mm_classifiers.pt / visual_tokens.pt plus eval_feat4cls.
    image_features [C*S, D]: l2_normalize(encode_image(images)) of THIS oracle, computed by the caller beforehand (a test that already
    ran the tower over the same images does not run it twice: ViT-L on a CPU takes a second per image); None = encode here.
    """
    dt = torch.float16 if prec == "fp16" else torch.float32
    C = tokenized_prompts.shape[0]
    S = images_by_class.shape[0] // (labels.unique().numel())
    D = sd["visual.proj"].shape[1]
    logit_scale = sd["logit_scale"].float().exp()                        # :238
    prompt_tokens = prompt_embeddings(tokenized_prompts, sd, dt)         # :129
    from ovmr_amd.synth import template_token_ids                        # constants only (token ids of "a .")
    vtemp = prompt_embeddings(torch.from_numpy(template_token_ids(tokenized_prompts.shape[1])), sd, dt)  # :130
    if text_classifier is None:
        text_classifier = zero_shot_classifier(tokenized_prompts, sd)    # :118-126
    mm_clf = torch.zeros(C, D, dtype=dt)
    v_clf = torch.zeros(C, D, dtype=dt)
    vis_tokens = torch.zeros(C, n_ctx, D, dtype=dt)
    eval_feats = torch.zeros(C, S, D, dtype=dt)
    step = classes_per_batch * S
    for s0 in range(0, images_by_class.shape[0], step):                  # :226
        img = images_by_class[s0:s0 + step]
        lab = labels[s0:s0 + step]
        ncls = img.shape[0] // S                                         # :237
        ex_label = lab.reshape(ncls, S)[:, 0]                            # :240
        tok = tokenized_prompts[ex_label]                                # :241
        if image_features is None:
            f = l2_normalize(encode_image(img.to(dt), sd)).reshape(ncls, S, -1)   # :243-245
        else:
            f = image_features[s0:s0 + step].to(dt).reshape(ncls, S, -1)
        eval_feats[ex_label] = f                                         # :247
        mm_p, mm_l, v_p, v_l, tokens = prompt_learner_forward(
            f, ex_label, tok.argmax(dim=-1), prompt_tokens, vtemp, pl, n_ctx)      # :248
        mm, v = get_mm_v_feats(mm_p, mm_l, v_p, v_l, sd, dt)            # :249
        mm_clf[ex_label] = mm.to(dt)                                     # :251
        v_clf[ex_label] = v.to(dt)                                       # :252
        vis_tokens[ex_label] = tokens.to(dt)                             # :255
    fw, f1 = fusion_weights(eval_feats, mm_clf, v_clf, text_classifier, logit_scale, tau)
    return {"text_classifier": text_classifier.float(), "vision_classifier": v_clf.float(),
            "mm_classifier": mm_clf.float(), "fusion_weight": fw.float(), "visual_tokens": vis_tokens,
            "eval_feat4cls": eval_feats, "f1": f1}


def inference_logits(image_features: Tensor, mm: Tensor, v: Tensor, t: Tensor, fusion_weight: Tensor,
                     logit_scale: Tensor, mode: str = "fusion") -> Tensor:
    """CustomCLIP.forward eval branch, trainers/mm_classifier_one_prompt.py:348-363.
    image_features are already L2-normalised (:307). Output [B,C] fp32 probabilities."""
    def sm(w):
        return (logit_scale * image_features @ w.t()).float().softmax(dim=-1)
    if mode == "text":
        return sm(t)
    if mode == "vision":
        return sm(v)
    if mode == "multimodal":
        return sm(mm)
    three = torch.stack([sm(mm), sm(v), sm(t)], dim=-1)                  # :361, order mm, v, t
    return torch.einsum("bmn,mn->bmn", three, fusion_weight.float()).sum(-1)   # :362


def zeroshot_logits(image: Tensor, text_features: Tensor, sd: Dict[str, Tensor]) -> Tensor:
    """ZeroshotCLIP.model_inference, trainers/zsclip.py:55-60 (raw logits, no softmax)."""
    f = l2_normalize(encode_image(image, sd))
    return sd["logit_scale"].float().exp() * f @ text_features.t()


def to_torch(sd_np) -> Dict[str, Tensor]:
    return {k: torch.from_numpy(v) if not isinstance(v, torch.Tensor) else v for k, v in sd_np.items()}
