"""Pins the CPU oracle (oracle/ovmr_oracle.py) against golden vectors recorded from the
real reference by tests/golden/gen_golden.py.  Runs anywhere (no GPU, no /root/reference)."""
import numpy as np
import pytest
import torch

from conftest import aligned_case, assert_cosine, cosine_rows, near_tie_classes
from ovmr_amd import synth
from oracle import ovmr_oracle as O

torch.set_num_threads(4)
SEED = 11


def _sd(spec, prec):
    # The reference's fp32 mode is build_model() (weights rounded to fp16, clip/model.py:934)
    # followed by clip_model.float() (trainers/mm_classifier_one_prompt.py:380-382).
    sd = O.convert_weights(O.to_torch(synth.clip_state_dict(spec, SEED, jitter=True)), "fp16")
    return sd if prec == "fp16" else {k: v.float() for k, v in sd.items()}


def _pl(spec, n_ctx):
    return {k: torch.from_numpy(v) for k, v in synth.prompt_learner_state_dict(spec, n_ctx, SEED, True).items()}


@pytest.mark.parametrize("name,key,n_img", [("tiny", "tiny", 4), ("small", "small", 3)])
@pytest.mark.parametrize("prec", ["fp16", "fp32"])
def test_l1_intermediates(golden, name, key, n_img, prec):
    """G1: per-op intermediates of VisionTransformer.forward + encode_text + zsclip logits."""
    g = golden(key)
    spec = synth.SPECS[name]
    sd = _sd(spec, prec)
    img = torch.from_numpy(synth.images(n_img, spec.image_resolution, seed=1234))
    taps = {}
    with torch.no_grad():
        feats = O.encode_image(img, sd, taps)
        ids = torch.from_numpy(synth.class_token_ids(6, seed=4321))
        tf = O.encode_text(ids, sd)
        zs = O.zeroshot_logits(img, O.l2_normalize(tf), sd)
    tol = 2e-5 if prec == "fp16" else 1e-10
    assert_cosine(taps["tokens"].float().numpy(), g[f"l1_{prec}_tokens"], tol, "tokens")
    assert_cosine(taps["ln_pre"].float().numpy(), g[f"l1_{prec}_ln_pre"], tol, "ln_pre")
    for i, b in enumerate(taps["blocks"]):
        assert_cosine(b.float().numpy(), g[f"l1_{prec}_block{i}"], tol, f"block{i}")
    assert_cosine(feats.float().numpy(), g[f"l1_{prec}_image_features"], tol, "image_features")
    assert_cosine(tf.float().numpy(), g[f"l1_{prec}_text_features"], tol, "text_features")
    np.testing.assert_allclose(zs.float().numpy(), g[f"l1_{prec}_zs_logits"],
                               atol=0.13 if prec == "fp16" else 2e-3)   # fp16 logits ~100 have 0.0625 spacing


@pytest.mark.parametrize("prec", ["fp16", "fp32"])
def test_l1_vitb16(golden, prec):
    """G2/G3/G8 at the real ViT-B/16 size."""
    g = golden("vitb16")
    spec = synth.SPECS["ViT-B/16"]
    sd = _sd(spec, prec)
    img = torch.from_numpy(synth.images(8, 224, seed=1234))
    with torch.no_grad():
        feats = O.encode_image(img, sd)
        tf = O.encode_text(torch.from_numpy(synth.class_token_ids(6, seed=4321)), sd)
    assert_cosine(feats.float().numpy(), g[f"l1_{prec}_image_features"], 2e-5, "image_features")
    assert_cosine(tf.float().numpy(), g[f"l1_{prec}_text_features"], 2e-5, "text_features")


@pytest.mark.parametrize("prec", ["fp16", "fp32"])
def test_c1_zeroshot_ten_prompts(golden, prec):
    """BASELINE.json configuration 1 (trainers/zsclip.py:32-60 on a 10-class Caltech-101 subset): the oracle's encode_text / zeroshot_logits on
    the prompts the REAL clip.tokenize produced for "a photo of a {}." against the real CLIP module's text features and raw logits, in the
    fp16 model and after .float() (clip/clip.py:130-131: what the reference runs on a CPU device).  First 6 of the 16 recorded images."""
    g = golden("c1_zeroshot")
    spec = synth.SPECS[str(g["c1_meta_spec"])]
    sd = _sd(spec, prec)
    ids = torch.from_numpy(g["c1_token_ids"])
    assert ids.shape == (10, 77) and str(g["c1_prompts"][0]) == "a photo of a accordion."
    img = torch.from_numpy(synth.images(int(g["c1_meta_n_img"]), spec.image_resolution, seed=int(g["c1_meta_img_seed"])))[:6]
    with torch.no_grad():
        tf = O.l2_normalize(O.encode_text(ids, sd))
        zs = O.zeroshot_logits(img.to(tf.dtype), tf, sd)
    assert_cosine(tf.float().numpy(), g[f"c1_{prec}_text_features"], 2e-5, "text features")
    assert_cosine(zs.float().numpy(), g[f"c1_{prec}_logits"][:6], 1e-4, "logits")
    np.testing.assert_allclose(zs.float().numpy(), g[f"c1_{prec}_logits"][:6], atol=0.02 if prec == "fp16" else 2e-3)


@pytest.mark.parametrize("prec", ["fp16", "fp32"])
def test_hot_trained_like_statistics(golden, prec):
    """The `hot` case: the real clip/model.py on ViT-B/16 (12 blocks) whose weights carry trained-like statistics
    (synth.trained_like_statistics: massive-activation channels, skewed LayerNorm gains, peaky attention, saturated QuickGELU inputs).
    The oracle must follow the reference there too -- it is what the GPU tests of the big jobs compare with."""
    g = golden("hot")
    spec = synth.SPECS[str(g["hot_meta_spec"])]
    sd_np = synth.clip_state_dict(spec, int(g["hot_meta_seed"]), jitter=True)
    hot = synth.trained_like_statistics(sd_np, spec, int(g["hot_meta_stat_seed"]))
    assert np.array_equal(hot, g["hot_channels"])
    sd = O.convert_weights(O.to_torch(sd_np), "fp16")
    if prec == "fp32":
        sd = {k: v.float() for k, v in sd.items()}
    n = 4
    img = torch.from_numpy(synth.images(int(g["hot_meta_n_img"]), spec.image_resolution, seed=int(g["hot_meta_img_seed"]), class_ids=np.arange(8) % 4,
                                        class_strength=float(g["hot_meta_strength"]), tile=int(g["hot_meta_tile"])))[:n]
    with torch.no_grad():
        f = O.encode_image(img.to(sd["visual.proj"].dtype), sd)
    assert_cosine(f.float().numpy(), g[f"hot_{prec}_image_features"][:n], 2e-5, "image features under trained-like statistics")
    assert np.abs(g[f"hot_{prec}_cls_stream"][:, hot]).min() > 15.0              # the massive channels are there, in the reference's own stream


@pytest.mark.parametrize("key,name,tag,n_ctx", [("tiny", "tiny", "l2", 2), ("tiny", "tiny", "l2n1", 1),
                                                ("small", "small", "l2", 2), ("vitb16", "ViT-B/16", "l2", 2)])
def test_l2_prompt_learner_and_text_encoder(golden, key, name, tag, n_ctx):
    """G3/G4: PromptLearner.forward and TextEncoder.forward on the reference's own inputs."""
    g = golden(key)
    spec = synth.SPECS[name]
    sd = _sd(spec, "fp16")
    tok = torch.from_numpy(g[f"{tag}_tokenized_prompts"])
    ptok = O.prompt_embeddings(tok, sd)
    np.testing.assert_array_equal(ptok.float().numpy(), g[f"{tag}_prompt_tokens"].astype(np.float32))
    vtemp = O.prompt_embeddings(torch.from_numpy(synth.template_token_ids()), sd)
    feats = torch.from_numpy(g[f"{tag}_pl_feats"]).half()
    with torch.no_grad():
        mm_p, mm_l, v_p, v_l, tokens = O.prompt_learner_forward(
            feats, torch.from_numpy(g[f"{tag}_pl_label"]), torch.from_numpy(g[f"{tag}_pl_eos"]),
            ptok, vtemp, _pl(spec, n_ctx), n_ctx)
        te_mm = O.text_encoder_forward(mm_p, mm_l, sd)
        te_v = O.text_encoder_forward(v_p, v_l, sd)
    assert tokens.dtype == torch.float32
    np.testing.assert_array_equal(mm_l.numpy(), g[f"{tag}_pl_mm_lens"])
    np.testing.assert_array_equal(v_l.numpy(), g[f"{tag}_pl_v_lens"])
    np.testing.assert_allclose(tokens.numpy(), g[f"{tag}_pl_tokens"], atol=2e-4, rtol=1e-4)
    assert_cosine(mm_p.float().numpy(), g[f"{tag}_pl_mm_prompts"].astype(np.float32), 1e-6, "mm_prompts")
    assert_cosine(v_p.float().numpy(), g[f"{tag}_pl_v_prompts"].astype(np.float32), 1e-6, "v_prompts")
    assert_cosine(te_mm.float().numpy(), g[f"{tag}_te_mm"], 2e-5, "te_mm")
    assert_cosine(te_v.float().numpy(), g[f"{tag}_te_v"], 2e-5, "te_v")


@pytest.mark.parametrize("key,name,tag,n_ctx", [("tiny", "tiny", "l2", 2), ("tiny", "tiny", "l2n1", 1),
                                                ("small", "small", "l2", 2), ("vitb16", "ViT-B/16", "l2", 2)])
def test_l2_forward_prompt_and_inference(golden, key, name, tag, n_ctx):
    """G5/G6: the tensors of mm_classifiers.pt / visual_tokens.pt and the four EVAL_MODE outputs."""
    g = golden(key)
    spec = synth.SPECS[name]
    sd = _sd(spec, "fp16")
    shots, cpb = int(g["meta_shots"]), int(g["meta_classes_per_batch"])
    labels = g[f"{tag}_eval_labels"]
    img = torch.from_numpy(synth.images(len(labels), spec.image_resolution, seed=1234,
                                        class_ids=labels, class_strength=0.6))
    tok = torch.from_numpy(g[f"{tag}_tokenized_prompts"])
    with torch.no_grad():
        r = O.forward_prompt(img, torch.from_numpy(labels), tok, sd, _pl(spec, n_ctx), n_ctx,
                             float(g["meta_tau"]), cpb, "fp16")
    assert_cosine(r["eval_feat4cls"].float().numpy(), g[f"{tag}_eval_feat4cls"], 2e-5, "eval_feat4cls")
    assert_cosine(r["text_classifier"].numpy(), g[f"{tag}_saved_text_classifier"], 2e-5, "text")
    assert_cosine(r["vision_classifier"].numpy(), g[f"{tag}_saved_vision_classifier"], 2e-5, "vision")
    assert_cosine(r["mm_classifier"].numpy(), g[f"{tag}_saved_mm_classifier"], 2e-5, "mm")
    assert_cosine(r["visual_tokens"].float().numpy(), g[f"{tag}_saved_visual_tokens"], 2e-5, "visual_tokens")

    # F1 -> fusion weights from the reference's own features/classifiers must match exactly
    ls = sd["logit_scale"].float().exp()
    fw, _ = O.fusion_weights(torch.from_numpy(g[f"{tag}_eval_feat4cls"]).half(),
                             torch.from_numpy(g[f"{tag}_saved_mm_classifier"]).half(),
                             torch.from_numpy(g[f"{tag}_saved_vision_classifier"]).half(),
                             torch.from_numpy(g[f"{tag}_saved_text_classifier"]).half(), ls, float(g["meta_tau"]))
    np.testing.assert_allclose(fw.numpy(), g[f"{tag}_saved_fusion_weight"], atol=1e-6)

    qf = torch.from_numpy(g[f"{tag}_query_features"]).half()
    for mode in ("fusion", "text", "vision", "multimodal"):
        out = O.inference_logits(qf, torch.from_numpy(g[f"{tag}_saved_mm_classifier"]).half(),
                                 torch.from_numpy(g[f"{tag}_saved_vision_classifier"]).half(),
                                 torch.from_numpy(g[f"{tag}_saved_text_classifier"]).half(),
                                 torch.from_numpy(g[f"{tag}_saved_fusion_weight"]), ls, mode)
        assert out.dtype == torch.float32
        np.testing.assert_allclose(out.numpy(), g[f"{tag}_logits_{mode}"], atol=1e-6, err_msg=mode)


@pytest.mark.parametrize("key,name,tag", [("tiny", "tiny", "l2a"), ("tiny", "tiny", "l2a1"), ("small", "small", "l2a"), ("small", "small", "l2a1"), ("vitb16", "ViT-B/16", "l2a")])
def test_l2a_fusion_weight_and_fused_output_unconditional(golden, key, name, tag):
    """The `l2a*` fixtures (aligned weights, every cross-validation argmax of the reference clear by more than `*_meta_margin`;
    `l2a` = n_ctx 2, `l2a1` = n_ctx 1; every fixture family -- tiny, small, ViT-B/16 -- has one): the whole generation job of the oracle -- features -> classifiers -> argmax counts -> F1 ->
    fusion_weight -> the four EVAL_MODE outputs -- against the reference's recorded tensors, fusion_weight EXACTLY."""
    g = golden(key)
    spec, sd_np, pl_np, labels, img, qlab, q = aligned_case(g, name, tag)
    n_ctx = int(g[f"{tag}_meta_n_ctx"]) if f"{tag}_meta_n_ctx" in g.files else 2
    C, S, tau = len(g[f"{tag}_classnames"]), int(g[f"{tag}_meta_shots"]), float(g[f"{tag}_meta_tau"])
    sd = O.convert_weights(O.to_torch(sd_np), "fp16")
    tok = torch.from_numpy(g[f"{tag}_tokenized_prompts"])
    ls = sd["logit_scale"].float().exp()
    # the fixture's contract, re-checked here: no near-tie in the reference's own logits
    ref_f = torch.from_numpy(g[f"{tag}_eval_feat4cls"]).half()
    for k in ("mm_classifier", "vision_classifier", "text_classifier"):
        lg = O.cross_validation_logits(ref_f, torch.from_numpy(g[f"{tag}_saved_{k}"]).half(), ls).float().numpy()
        assert not near_tie_classes(lg, float(g[f"{tag}_meta_margin"])), k
    with torch.no_grad():
        r = O.forward_prompt(torch.from_numpy(img), torch.from_numpy(labels), tok, sd, O.to_torch(pl_np), n_ctx, tau,
                             int(g[f"{tag}_meta_classes_per_batch"]), "fp16")
        qf = O.l2_normalize(O.encode_image(torch.from_numpy(q).half(), sd))
    for k in ("text_classifier", "vision_classifier", "mm_classifier"):
        assert_cosine(r[k].numpy(), g[f"{tag}_saved_{k}"], 2e-5, k)
    assert_cosine(r["visual_tokens"].float().numpy(), g[f"{tag}_saved_visual_tokens"], 2e-5, "visual_tokens")
    assert_cosine(r["eval_feat4cls"].float().numpy(), g[f"{tag}_eval_feat4cls"], 2e-5, "eval_feat4cls")
    np.testing.assert_allclose(r["fusion_weight"].numpy(), g[f"{tag}_saved_fusion_weight"], atol=1e-6)
    assert len(np.unique(np.round(g[f"{tag}_saved_fusion_weight"], 3), axis=0)) >= min(4, C // 2), "degenerate fixture"
    for mode in ("fusion", "text", "vision", "multimodal"):
        out = O.inference_logits(qf, r["mm_classifier"].half(), r["vision_classifier"].half(), r["text_classifier"].half(),
                                 r["fusion_weight"], ls, mode)
        # logits ~100 carry fp16 steps of 0.06, which the softmax turns into 6 % steps of a probability; the 128-wide tiny model has the
        # fewest terms to average them out (same allowance as the per-image outputs of the older tiny fixtures)
        assert_cosine(out.numpy(), g[f"{tag}_logits_{mode}"], 5e-3 if name == "tiny" else 1e-3, mode)


def test_state_dict_keys(golden):
    """PromptLearner.state_dict() is cls_token + 4 x 12 aggregator tensors."""
    g = golden("tiny")
    ours = sorted(synth.prompt_learner_state_dict(synth.SPECS["tiny"], 2, SEED).keys())
    assert ours == [str(k) for k in g["l2_state_dict_keys"]]
    assert len(ours) == 49


def test_f1_known_answers():
    """G7: torcheval multiclass_f1_score(average=None) semantics, hand-computed + sklearn."""
    # 3 classes, 2 samples each.  preds: [0,0, 0,1, 1,1]  labels: [0,0,1,1,2,2]
    logits = torch.tensor([[9., 0, 0], [9, 0, 0], [9, 0, 0], [0, 9, 0], [0, 9, 0], [0, 9, 0]])
    labels = torch.tensor([0, 0, 1, 1, 2, 2])
    f1 = O.multiclass_f1_per_class(logits, labels, 3)
    # class0: tp2 pred3 label2 -> p=2/3 r=1 f1=0.8 ; class1: tp1 pred3 label2 -> p=1/3 r=1/2 f1=0.4 ; class2: never predicted -> NaN -> 0
    np.testing.assert_allclose(f1.numpy(), [0.8, 0.4, 0.0], atol=1e-7)
    from sklearn.metrics import f1_score
    rng = np.random.default_rng(0)
    lg = torch.from_numpy(rng.normal(size=(400, 20)).astype(np.float32))
    lb = torch.from_numpy(np.repeat(np.arange(20), 20))
    ours = O.multiclass_f1_per_class(lg, lb, 20).numpy()
    ref = f1_score(lb.numpy(), lg.argmax(1).numpy(), labels=np.arange(20), average=None, zero_division=0)
    np.testing.assert_allclose(ours, ref, atol=1e-6)
    # a class that is never predicted by any classifier gets softmax(0,0,0) = 1/3 each
    w = (10.0 * torch.zeros(1, 3)).softmax(-1)
    np.testing.assert_allclose(w.numpy(), [[1 / 3] * 3], atol=1e-7)
    # argmax ties resolve to the first index (torch.argmax on CPU)
    assert int(torch.tensor([[1.0, 2.0, 2.0]]).argmax(1)) == 1


def test_tokenizer_fixture(golden):
    """G9: ids the reference tokenizer produced; the synthetic-token layout matches them."""
    g = golden("tokenizer")
    ids = g["tok_ids"]
    assert list(ids[0, :4]) == [synth.SOT_ID, synth.TOK_A, synth.TOK_DOT, synth.EOT_ID]
    assert list(ids[1, :5]) == [49406, 320, 48760, 269, 49407]
    assert list(ids[2, :6]) == [49406, 320, 5992, 5084, 269, 49407]
    np.testing.assert_array_equal(ids[0], synth.template_token_ids()[0])
    t = synth.class_token_ids(50)
    eos = t.argmax(-1)
    assert ((eos >= 4) & (eos <= 7)).all() and (t[np.arange(50), eos] == synth.EOT_ID).all()
    assert (t[:, 0] == synth.SOT_ID).all() and (t[:, 1] == synth.TOK_A).all()
    assert (t[np.arange(50), eos - 1] == synth.TOK_DOT).all()


@pytest.mark.parametrize("key,tag", [("tiny", "l2"), ("small", "l2"), ("vitb16", "l2")])
def test_coop_fusion_weight_variant_equals_recorded(golden, key, tag):
    """trainers/coop_mm_classifier.py:235-305 on the reference's own recorded features and classifiers: its permuted
    [S, C, .] layout and fixed tau = 10 must reproduce the fusion weights the reference saved (recorded with tau = 10)."""
    g = golden(key)
    assert float(g["meta_tau"]) == 10.0
    feats = torch.from_numpy(g[f"{tag}_eval_feat4cls"]).half()
    clf = [torch.from_numpy(g[f"{tag}_saved_{k}"]).half() for k in ("mm_classifier", "vision_classifier", "text_classifier")]
    w = O.get_fusion_weight_coop(feats, *clf, torch.tensor(float(np.exp(np.log(100.0)))))
    np.testing.assert_allclose(w.numpy(), g[f"{tag}_saved_fusion_weight"], atol=1e-6)
