#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by running the REAL reference.

Runs ONLY in the build container (needs /root/reference, which never travels to the GPU
box).  It imports the reference's own modules -- clip/model.py (L1) and
trainers/mm_classifier_one_prompt.py (L2) -- feeds them the seeded synthetic weights and
inputs of ovmr_amd/synth.py, and records their outputs as small .npz fixtures.  Nothing
from the reference (source, bytecode, vocabulary) is copied into this repository: the
fixtures hold only inputs that cannot be regenerated (token ids the reference tokenizer
produced for a few class names) and the reference's numeric outputs.

Harness: the reference imports torchvision / ftfy / torcheval /
dassl, none of which exist offline, and calls .cuda() unconditionally.  We install
name-only stubs for those modules and make .cuda() the identity.  torcheval's
multiclass_f1_score is the one stub that carries arithmetic; it is restated from
torcheval 0.0.7's published algorithm (argmax -> tp/n_pred/n_label -> 2pr/(p+r) ->
nan_to_num), see oracle/ovmr_oracle.py:multiclass_f1_per_class.

Usage:  PYTHONDONTWRITEBYTECODE=1 python tests/golden/gen_golden.py [--only tiny|small|vitb16|tok|hot|c1]
"""
from __future__ import annotations

import argparse
import os
import sys
import tempfile
import types
from types import SimpleNamespace

sys.dont_write_bytecode = True
import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("OVMR_REFERENCE", "/root/reference")
sys.path.insert(0, REPO)

from ovmr_amd import synth  # noqa: E402


# ----------------------------------------------------------------------------- harness
def _f1_stub(input, target, *, num_classes=None, average="micro"):
    assert average is None
    pred = input.argmax(dim=1)
    target = target.long()
    tp = torch.bincount(target[pred == target], minlength=num_classes).float()
    n_label = torch.bincount(target, minlength=num_classes).float()
    n_pred = torch.bincount(pred, minlength=num_classes).float()
    p, r = tp / n_pred, tp / n_label
    return torch.nan_to_num(2 * p * r / (p + r))


def install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    class _Any:
        def __init__(self, *a, **k):
            pass

    tv = mod("torchvision")
    tv.transforms = mod("torchvision.transforms", Compose=_Any, Resize=_Any, CenterCrop=_Any, ToTensor=_Any,
                        Normalize=_Any, InterpolationMode=SimpleNamespace(BICUBIC=3))
    mod("ftfy", fix_text=lambda s: s)
    te = mod("torcheval")
    te.metrics = mod("torcheval.metrics")
    te.metrics.functional = mod("torcheval.metrics.functional", multiclass_f1_score=_f1_stub,
                                multiclass_precision=None, multiclass_recall=None)

    class _Registry:
        def register(self):
            return lambda cls: cls

    class TrainerX:
        pass

    d = mod("dassl")
    d.engine = mod("dassl.engine", TRAINER_REGISTRY=_Registry(), TrainerX=TrainerX)
    d.metrics = mod("dassl.metrics", compute_accuracy=None)
    d.utils = mod("dassl.utils", load_pretrained_weights=None, load_checkpoint=None)
    d.optim = mod("dassl.optim", build_optimizer=None, build_lr_scheduler=None)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self


def import_reference():
    install_stubs()
    sys.path.insert(0, REF)
    import clip.model as ref_model                       # L1
    from clip import clip as ref_clip                    # tokenizer
    import trainers.mm_classifier_one_prompt as ref_l2   # L2
    return ref_model, ref_clip, ref_l2


def make_cfg(n_ctx, shots, tau, mode, outdir, train_n_ins=8):
    return SimpleNamespace(
        TRAINER=SimpleNamespace(COCOOP=SimpleNamespace(N_CTX=n_ctx, PREC="fp16")),
        INPUT=SimpleNamespace(SIZE=(224, 224)),       # only compared with the constant 224 (:103-107)
        DATALOADER=SimpleNamespace(TRAIN_X=SimpleNamespace(BATCH_SIZE=64, N_INS=train_n_ins), K_TRANSFORMS=1),
        DATASET=SimpleNamespace(NUM_SHOTS=shots),
        EVAL_TAU=tau, EVAL_MODE=mode, OUTPUT_DIR=outdir,
        MODEL=SimpleNamespace(BACKBONE=SimpleNamespace(NAME="synthetic")))


def build_ref_clip(ref_model, spec, seed, jitter, fp32=False):
    sd = {k: torch.from_numpy(v) for k, v in synth.clip_state_dict(spec, seed, jitter).items()}
    model = ref_model.build_model(dict(sd))               # convert_weights -> fp16 + load_state_dict
    if fp32:
        model = model.float()
    return model.eval()


# the aligned (`l2a*`) cases per fixture file: tag -> gen_l2_aligned arguments.  Gain, shots and class count were searched per
# model (and per n_ctx) for the contract the function asserts: every cross-validation argmax of the reference clear by > MARGIN.
ALIGNED_CASES = {
    "tiny": {"l2a": dict(gain=6.0, shots=16, C=6, cpb=3, n_ctx=2), "l2a1": dict(gain=6.0, shots=8, C=6, cpb=3, n_ctx=1)},
    "small": {"l2a": dict(gain=3.0), "l2a1": dict(gain=5.0, n_ctx=1)},
    "vitb16": {"l2a": dict(gain=1.5)},
}
CLASSNAMES = ["accordion", "bass guitar", "airplane", "sea_horse", "stop sign", "yin yang"]


# ----------------------------------------------------------------------------- L1 vectors
@torch.no_grad()
def gen_l1(ref_model, spec, seed, n_img, with_taps, out):
    img = torch.from_numpy(synth.images(n_img, spec.image_resolution, seed=1234))
    ids = torch.from_numpy(synth.class_token_ids(6, seed=4321))
    for tag, fp32 in (("fp16", False), ("fp32", True)):
        m = build_ref_clip(ref_model, spec, seed, True, fp32)
        x = img.to(m.dtype)
        if with_taps:
            v = m.visual
            t = v.conv1(x)
            t = t.reshape(t.shape[0], t.shape[1], -1).permute(0, 2, 1)
            t = torch.cat([v.class_embedding.to(t.dtype) + torch.zeros(t.shape[0], 1, t.shape[-1], dtype=t.dtype), t], 1)
            t = t + v.positional_embedding.to(t.dtype)
            out[f"l1_{tag}_tokens"] = t.float().numpy()
            t = v.ln_pre(t)
            out[f"l1_{tag}_ln_pre"] = t.float().numpy()
            t = t.permute(1, 0, 2)
            for i, blk in enumerate(v.transformer.resblocks):
                t = blk(t)
                out[f"l1_{tag}_block{i}"] = t.permute(1, 0, 2).float().numpy()
        out[f"l1_{tag}_image_features"] = m.encode_image(x).float().numpy()
        out[f"l1_{tag}_text_features"] = m.encode_text(ids).float().numpy()
        # zsclip-style raw logits (trainers/zsclip.py:55-60); that file itself is unimportable
        f = m.encode_image(x)
        f = f / f.norm(dim=-1, keepdim=True)
        tf = m.encode_text(ids)
        tf = tf / tf.norm(dim=-1, keepdim=True)
        out[f"l1_{tag}_zs_logits"] = (m.logit_scale.exp() * f @ tf.t()).float().numpy()


# ----------------------------------------------------------------------------- L2 vectors
@torch.no_grad()
def gen_l2(ref_model, ref_clip, ref_l2, spec, seed, n_ctx, shots, tau, classes_per_batch, n_query, out, tag):
    C = len(CLASSNAMES)
    with tempfile.TemporaryDirectory() as outdir:
        cfg = make_cfg(n_ctx, shots, tau, "fusion", outdir)
        clip_model = build_ref_clip(ref_model, spec, seed, True)
        torch.manual_seed(0)
        model = ref_l2.CustomCLIP(cfg, CLASSNAMES, clip_model).eval()
        pl_sd = {k: torch.from_numpy(v) for k, v in synth.prompt_learner_state_dict(spec, n_ctx, seed, True).items()}
        missing = model.prompt_learner.load_state_dict(pl_sd, strict=True)
        model.device = torch.device("cpu")

        tok = model.tokenized_prompts
        out[f"{tag}_tokenized_prompts"] = tok.numpy()
        out[f"{tag}_prompt_tokens"] = model.prompt_learner.prompt_tokens.float().numpy()
        out[f"{tag}_zero_shot_classifier"] = model.zero_shot_classifier.float().numpy()

        # eval-set loader contract: S consecutive rows per class, classes shuffled
        order = [3, 0, 5, 1, 2, 4][:C]
        labels = np.repeat(np.array(order, dtype=np.int64), shots)
        img = synth.images(C * shots, spec.image_resolution, seed=1234, class_ids=labels, class_strength=0.6)
        out[f"{tag}_eval_labels"] = labels
        step = classes_per_batch * shots
        loader = [{"img": torch.from_numpy(img[s:s + step]), "label": torch.from_numpy(labels[s:s + step])}
                  for s in range(0, C * shots, step)]

        # PromptLearner.forward / TextEncoder.forward vectors on the first batch
        b0 = loader[0]
        ex_label = b0["label"].reshape(-1, shots)[:, 0]
        f = model.image_encoder(b0["img"].half())
        f = (f / f.norm(dim=-1, keepdim=True)).reshape(len(ex_label), shots, -1)
        eos = tok[ex_label].argmax(dim=-1)
        mm_p, mm_l, v_p, v_l, tokens = model.prompt_learner(f, ex_label, eos)
        out[f"{tag}_pl_feats"] = f.float().numpy()
        out[f"{tag}_pl_label"] = ex_label.numpy()
        out[f"{tag}_pl_eos"] = eos.numpy()
        out[f"{tag}_pl_tokens"] = tokens.float().numpy()
        out[f"{tag}_pl_mm_prompts"] = mm_p[0].float().numpy()
        out[f"{tag}_pl_v_prompts"] = v_p[0].float().numpy()
        out[f"{tag}_pl_mm_lens"] = mm_l.numpy()
        out[f"{tag}_pl_v_lens"] = v_l.numpy()
        out[f"{tag}_te_mm"] = model.text_encoder(mm_p[0], mm_l).float().numpy()
        out[f"{tag}_te_v"] = model.text_encoder(v_p[0], v_l).float().numpy()

        # full generation + the four inference modes
        qlab = np.arange(n_query, dtype=np.int64) % C
        q = torch.from_numpy(synth.images(n_query, spec.image_resolution, seed=777, class_ids=qlab, class_strength=0.6))
        out[f"{tag}_query_labels"] = qlab
        for mode in ("fusion", "text", "vision", "multimodal"):
            cfg.EVAL_MODE = mode
            out[f"{tag}_logits_{mode}"] = model(q, eval_set_loader=loader).float().numpy()
        qf = model.image_encoder(q.half())
        out[f"{tag}_query_features"] = (qf / qf.norm(dim=-1, keepdim=True)).float().numpy()
        saved = torch.load(os.path.join(outdir, "mm_classifiers.pt"))
        for k, v in saved.items():
            assert v.dtype == torch.float32
            out[f"{tag}_saved_{k}"] = v.numpy()
        vt = torch.load(os.path.join(outdir, "visual_tokens.pt"))["visual_tokens"]
        assert vt.dtype == torch.float16
        out[f"{tag}_saved_visual_tokens"] = vt.float().numpy()
        out[f"{tag}_eval_feat4cls"] = model.eval_feat4cls.float().numpy()
        out[f"{tag}_state_dict_keys"] = np.array(sorted(model.prompt_learner.state_dict().keys()))


# ----------------------------------------------------------------------------- L2 vectors, "aligned" weights
# With plain random weights the classifier rows are unrelated to the image features, every cross-validation argmax
# (trainers/mm_classifier_one_prompt.py:266-270) is decided by noise, and some of them by less than one fp16 step --
# a comparison of fusion_weight with another implementation is then ill-posed.  The "l2a" fixtures use
#   * synth.align_state_dicts (an identity component in the value / output projections of the text tower and the
#     aggregator, so classifier rows point towards their own class's image features, as trained OVMR weights do),
#   * tiled class patterns at strength 0.9 (features of different classes separate),
#   * 12 classes x 8 shots, a few exemplars per class carrying the NEXT class's pattern (clear-margin mistakes, so
#     tp / n_pred / F1 differ from class to class),
#   * class names chosen from NAME_POOL so that the zero-shot text classifier's argmax is clear on every row:
#     one name wins everywhere by > TEXT_GAP, the other eleven are the pool's lowest-scoring names.
# The script asserts that EVERY cross-validation row of the reference has a top-2 margin > MARGIN for all three classifiers,
# so tests compare fusion_weight / logits_fusion with these fixtures unconditionally.
NAME_POOL = ["accordion", "bass guitar", "airplane", "sea_horse", "stop sign", "yin yang", "tench", "goldfish",
             "hammerhead shark", "electric ray", "hen", "ostrich", "bulbul", "jay", "magpie", "water ouzel", "kite",
             "bald eagle", "great grey owl", "fire salamander", "bullfrog", "tree frog", "loggerhead", "mud turtle",
             "banded gecko", "green lizard", "komodo dragon", "african crocodile", "triceratops", "thunder snake",
             "garter snake", "sea snake", "trilobite", "scorpion", "garden spider", "tick", "centipede", "black grouse",
             "peacock", "quail", "macaw", "drake", "goose", "black swan", "wombat", "jellyfish", "sea anemone", "flatworm"]
MARGIN, TEXT_GAP = 0.5, 0.75


def aligned_inputs(spec, C, shots, n_query, strength, tile):
    """Labels, pattern ids (a class's last c % 3 shots carry the next class's pattern) and images of the l2a case."""
    order = np.random.default_rng(5).permutation(C).astype(np.int64)
    labels = np.repeat(order, shots)
    pattern = labels.copy()
    for i, c in enumerate(order):
        m = int(c) % 3
        if m:
            pattern[(i + 1) * shots - m:(i + 1) * shots] = (int(c) + 1) % C
    img = synth.images(C * shots, spec.image_resolution, seed=1234, class_ids=pattern, class_strength=strength, tile=tile)
    qlab = np.arange(n_query, dtype=np.int64) % C
    q = synth.images(n_query, spec.image_resolution, seed=777, class_ids=qlab, class_strength=strength, tile=tile)
    return labels, pattern, img, qlab, q


@torch.no_grad()
def gen_l2_aligned(ref_model, ref_l2, spec, seed, gain, out, tag="l2a", C=12, shots=8, cpb=4, n_query=8,
                   strength=0.9, tile=16, tau=3.0, n_ctx=2, strict=True):
    sd_np = synth.clip_state_dict(spec, seed, jitter=True)
    pl_np = synth.prompt_learner_state_dict(spec, n_ctx, seed, True)
    synth.align_state_dicts(sd_np, pl_np, spec, gain)
    clip_model = ref_model.build_model({k: torch.from_numpy(v) for k, v in sd_np.items()}).eval()
    pl_sd = {k: torch.from_numpy(v) for k, v in pl_np.items()}
    labels, pattern, img, qlab, q = aligned_inputs(spec, C, shots, n_query, strength, tile)
    ls = clip_model.logit_scale.exp()

    # pass 1: choose the class names.  Text rows of the whole pool against the exemplar features.
    with tempfile.TemporaryDirectory() as outdir:
        pool = ref_l2.CustomCLIP(make_cfg(n_ctx, shots, tau, "fusion", outdir), NAME_POOL, clip_model).eval()
        f = pool.image_encoder(torch.from_numpy(img).half())
        f = f / f.norm(dim=-1, keepdim=True)
        lg = (ls * f @ pool.zero_shot_classifier.t()).float().numpy()          # [C*S, pool]
    win = int(lg.mean(0).argmax())
    ok = [j for j in range(len(NAME_POOL)) if j != win and (lg[:, win] - lg[:, j]).min() > TEXT_GAP]
    assert len(ok) >= C - 1, f"only {len(ok)} pool names stay {TEXT_GAP} below the winner on every row"
    ok = sorted(ok, key=lambda j: lg[:, j].max())[:C - 1]
    names = [NAME_POOL[j] for j in sorted(ok)]
    names.insert(7, NAME_POOL[win])
    out[f"{tag}_classnames"] = np.array(names)

    # pass 2: the generation job itself
    with tempfile.TemporaryDirectory() as outdir:
        cfg = make_cfg(n_ctx, shots, tau, "fusion", outdir)
        model = ref_l2.CustomCLIP(cfg, names, clip_model).eval()
        model.prompt_learner.load_state_dict(pl_sd, strict=True)
        model.device = torch.device("cpu")
        out[f"{tag}_tokenized_prompts"] = model.tokenized_prompts.numpy()
        out[f"{tag}_eval_labels"] = labels
        out[f"{tag}_eval_pattern_ids"] = pattern
        out[f"{tag}_query_labels"] = qlab
        step = cpb * shots
        loader = [{"img": torch.from_numpy(img[s:s + step]), "label": torch.from_numpy(labels[s:s + step])}
                  for s in range(0, C * shots, step)]
        qt = torch.from_numpy(q)
        for mode in ("fusion", "text", "vision", "multimodal"):
            cfg.EVAL_MODE = mode
            out[f"{tag}_logits_{mode}"] = model(qt, eval_set_loader=loader).float().numpy()
        saved = torch.load(os.path.join(outdir, "mm_classifiers.pt"))
        for k, v in saved.items():
            out[f"{tag}_saved_{k}"] = v.numpy()
        out[f"{tag}_saved_visual_tokens"] = torch.load(os.path.join(outdir, "visual_tokens.pt"))["visual_tokens"].float().numpy()
        out[f"{tag}_eval_feat4cls"] = model.eval_feat4cls.float().numpy()
    for k, v in (("gain", gain), ("shots", shots), ("classes_per_batch", cpb), ("tau", tau), ("strength", strength),
                 ("tile", tile), ("margin", MARGIN), ("n_ctx", n_ctx)):
        out[f"{tag}_meta_{k}"] = np.array(v)

    # the fixture's contract: every argmax of the reference is clear
    rows = torch.from_numpy(out[f"{tag}_eval_feat4cls"]).half()
    stats = {}
    for k in ("mm_classifier", "vision_classifier", "text_classifier"):
        l3 = (ls * torch.einsum("bmc,pc->bmp", rows, torch.from_numpy(out[f"{tag}_saved_{k}"]).half())).flatten(0, 1).float().numpy()
        srt = np.sort(l3, axis=1)
        margin = srt[:, -1] - srt[:, -2]
        acc = float((l3.argmax(1) == np.repeat(np.arange(C), shots)).mean())
        stats[k] = (float(margin.min()), acc)
        assert not strict or margin.min() > MARGIN, f"{tag} {k}: top-2 margin {margin.min():.3f} <= {MARGIN}"
    qm = {}
    for mode, k in (("multimodal", "mm_classifier"), ("vision", "vision_classifier"), ("text", "text_classifier")):
        qf = out[f"{tag}_logits_{mode}"]
        qm[mode] = float(np.sort(qf, axis=1)[:, -1].min())
    print(f"{tag} {spec.name}: names {names}\n   (min margin, accuracy) {stats}\n   fusion_weight\n{np.round(out[tag + '_saved_fusion_weight'], 3)}")
    return stats


# ----------------------------------------------------------------------------- `hot`: trained-like statistics through the real reference
HOT = dict(spec="ViT-B/16", seed=11, stat_seed=17, n_img=8, img_seed=5, strength=0.7, tile=16)


@torch.no_grad()
def gen_hot(ref_model, out):
    """The REAL clip/model.py (fp16 after convert_weights, and .float()) on ViT-B/16 -- all 12 blocks -- with
    synth.trained_like_statistics applied to the CLIP-init weights: massive-activation channels, skewed LayerNorm gains, peaky
    attention, saturated QuickGELU inputs.  Pins the engine's two default numerics deviations (LayerNorm fold, one-rounding QuickGELU)
    with the reference itself where CLIP-init statistics cannot (tests/test_hip_parity.py::test_encoder_under_trained_like_statistics)."""
    spec = synth.SPECS[HOT["spec"]]
    sd_np = synth.clip_state_dict(spec, HOT["seed"], jitter=True)
    hot = synth.trained_like_statistics(sd_np, spec, HOT["stat_seed"])
    img = torch.from_numpy(synth.images(HOT["n_img"], spec.image_resolution, seed=HOT["img_seed"], class_ids=np.arange(HOT["n_img"]) % 4,
                                        class_strength=HOT["strength"], tile=HOT["tile"]))
    sd = {k: torch.from_numpy(v) for k, v in sd_np.items()}
    for tag, fp32 in (("fp16", False), ("fp32", True)):
        m = ref_model.build_model(dict(sd))
        m = (m.float() if fp32 else m).eval()
        f = m.encode_image(img.to(m.dtype))
        assert bool(torch.isfinite(f).all())
        out[f"hot_{tag}_image_features"] = f.float().numpy()
        # the residual stream in front of ln_post, CLS rows: where the massive channels live (magnitude recorded for the test's report)
        v = m.visual
        t = v.conv1(img.to(m.dtype))
        t = t.reshape(t.shape[0], t.shape[1], -1).permute(0, 2, 1)
        t = torch.cat([v.class_embedding.to(t.dtype) + torch.zeros(t.shape[0], 1, t.shape[-1], dtype=t.dtype), t], 1)
        t = v.ln_pre(t + v.positional_embedding.to(t.dtype)).permute(1, 0, 2)
        t = v.transformer(t).permute(1, 0, 2)
        out[f"hot_{tag}_cls_stream"] = t[:, 0].float().numpy()
    out["hot_channels"] = hot.astype(np.int64)
    for k, v in HOT.items():
        out[f"hot_meta_{k}"] = np.array(v)


# ----------------------------------------------------------------------------- `c1`: BASELINE.json configuration 1 (zero-shot CLIP, 10 classes)
C1_CLASSES = ["accordion", "airplane", "anchor", "ant", "barrel", "bass", "beaver", "binocular", "bonsai", "brain"]   # the first ten Caltech-101
C1 = dict(spec="ViT-B/16", seed=11, n_img=16, img_seed=1234)                                                          # categories, alphabetically


@torch.no_grad()
def gen_c1(ref_model, ref_clip, out):
    """trainers/zsclip.py:32-60 on a 10-class Caltech-101 subset, restated line by line on the real `CLIP` module (the trainer file needs
    dassl to import): prompts = CUSTOM_TEMPLATES["Caltech101"].format(name) tokenised by the REAL clip.tokenize (:42-45), text features
    normalised (:49-50), per batch normalised image features (:56-57) and logit_scale.exp() * f @ t.T (:58-59) -- in the fp16 model (the
    GPU path of the reference) and after .float() (what clip.load does on a CPU device, clip/clip.py:130-131)."""
    spec = synth.SPECS[C1["spec"]]
    prompts = ["a photo of a {}.".format(c.replace("_", " ")) for c in C1_CLASSES]        # :22, :42-43
    ids = torch.cat([ref_clip.tokenize(p) for p in prompts])                              # :45
    img = torch.from_numpy(synth.images(C1["n_img"], spec.image_resolution, seed=C1["img_seed"]))
    out["c1_prompts"] = np.array(prompts)
    out["c1_token_ids"] = ids.numpy()
    for tag, fp32 in (("fp16", False), ("fp32", True)):
        m = build_ref_clip(ref_model, spec, C1["seed"], True, fp32)
        tf = m.encode_text(ids)
        tf = tf / tf.norm(dim=-1, keepdim=True)                                           # :49-50
        f = m.encode_image(img.to(m.dtype))
        f = f / f.norm(dim=-1, keepdim=True)                                              # :56-57
        logits = m.logit_scale.exp() * f @ tf.t()                                         # :58-59
        out[f"c1_{tag}_text_features"] = tf.float().numpy()
        out[f"c1_{tag}_image_features"] = f.float().numpy()
        out[f"c1_{tag}_logits"] = logits.float().numpy()
    for k, v in C1.items():
        out[f"c1_meta_{k}"] = np.array(v)


def gen_tokenizer(ref_clip, out):
    names = ["a .", "a accordion.", "a bass guitar.", "a sea horse.", "a photo of a yin yang.",
             "a great white shark.", "a toilet tissue.", "a hen-of-the-woods.", "a jack-o'-lantern.", "a T-shirt.",
             "a photo of a Boeing 737-800, a type of aircraft.", "a centered satellite photo of annual crop land.",
             "a photo of a person doing Apply_Eye_Makeup.".replace("_", " "), "a café au lait.", "a 3D printer's nozzle!!",
             "a   tab\tand  spaces .", "a &amp; b &lt;tag&gt;.", "a don't we'll they're I'm he'd.", "a naïve façade — déjà vu.",
             "a 日本語 テスト.", "a photo of 12345 67 8.", "a #hashtag @user $100 50%.", "a pneumonoultramicroscopicsilicovolcanoconiosis.",
             "a MiXeD CaSe StRiNg.", "a (parenthesised) [bracketed] {braced}."]
    out["tok_texts"] = np.array(names)
    out["tok_ids"] = torch.cat([ref_clip.tokenize(t) for t in names]).numpy()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="all")
    args = ap.parse_args()
    torch.set_num_threads(os.cpu_count())
    ref_model, ref_clip, ref_l2 = import_reference()

    if args.only in ("all", "tok"):
        out = {}
        gen_tokenizer(ref_clip, out)
        np.savez_compressed(os.path.join(HERE, "tokenizer.npz"), **out)
        print("tokenizer.npz", {k: v.shape for k, v in out.items()})

    for key, fn in (("hot", lambda o: gen_hot(ref_model, o)), ("c1", lambda o: gen_c1(ref_model, ref_clip, o))):
        if args.only in ("all", key):
            out = {}
            fn(out)
            path = os.path.join(HERE, {"hot": "hot.npz", "c1": "c1_zeroshot.npz"}[key])
            np.savez_compressed(path, **out)
            print(path, os.path.getsize(path) // 1024, "KiB", {k: v.shape for k, v in out.items() if v.ndim})

    for name, n_img, taps, shots, cpb, nq in (("tiny", 4, True, 4, 2, 5), ("small", 3, True, 4, 4, 5),
                                               ("ViT-B/16", 8, False, 4, 2, 5)):
        key = {"tiny": "tiny", "small": "small", "ViT-B/16": "vitb16"}[name]
        if args.only not in ("all", key):
            continue
        spec = synth.SPECS[name]
        out = {"meta_spec": np.array(name), "meta_seed": np.array(11), "meta_shots": np.array(shots),
               "meta_classes_per_batch": np.array(cpb), "meta_tau": np.array(10.0)}
        gen_l1(ref_model, spec, 11, n_img, taps, out)
        gen_l2(ref_model, ref_clip, ref_l2, spec, 11, 2, shots, 10.0, cpb, nq, out, "l2")
        if key != "vitb16":
            gen_l2(ref_model, ref_clip, ref_l2, spec, 11, 1, shots, 10.0, cpb, nq, out, "l2n1")
        for tag, kw in ALIGNED_CASES.get(key, {}).items():
            kw = dict(kw)
            gen_l2_aligned(ref_model, ref_l2, spec, 11, kw.pop("gain"), out, tag=tag, **kw)
        # fp16-valued tensors are stored as float16 (lossless, checked); everything else as produced
        store = {}
        for k, v in out.items():
            if v.dtype == np.float32 and v.size > 1024:
                with np.errstate(over="ignore"):
                    h = v.astype(np.float16)
                if np.array_equal(h.astype(np.float32), v):
                    v = h
            store[k] = v
        path = os.path.join(HERE, f"{key}.npz")
        np.savez_compressed(path, **store)
        print(path, os.path.getsize(path) // 1024, "KiB")


if __name__ == "__main__":
    main()
