"""CPU tests of the rows SURVEY.md section 8f marks "next": tokenizer, checkpoint files, evaluator, runner helpers."""
import gzip
import os

import numpy as np
import pytest
import torch

from conftest import REPO
from ovmr_amd import synth

REF_BPE = "/root/reference/clip/bpe_simple_vocab_16e6.txt.gz"   # present only in the build container


def make_synthetic_bpe(path):
    """A merge table with the right number of lines (ids stay < 49408); merges are arbitrary but well formed."""
    from ovmr_amd.tokenizer import N_MERGES, _byte_alphabet
    al = _byte_alphabet()
    lines, seen = ["#version: synthetic"], set()
    i = 0
    while len(lines) - 1 < N_MERGES:
        a, b = al[33 + (i % 94)], al[33 + ((i // 94) % 94)] + al[33 + ((i // 8836) % 94)] + "</w>"
        i += 1
        if (a, b) in seen:
            continue
        seen.add((a, b))
        lines.append(f"{a} {b}")
    with gzip.open(path, "wt", encoding="utf-8") as f:
        f.write("\n".join(lines) + "\n")


@pytest.mark.skipif(not os.path.exists(REF_BPE), reason="CLIP merge table only available next to the reference")
def test_tokenizer_matches_reference_ids(golden):
    from ovmr_amd.tokenizer import BPETokenizer
    g = golden("tokenizer")
    tk = BPETokenizer(REF_BPE)
    ids = tk.tokenize([str(t) for t in g["tok_texts"]])
    np.testing.assert_array_equal(ids.numpy(), g["tok_ids"])
    assert tk.encode("accordion") == [48760] and tk.encode("a") == [synth.TOK_A] and tk.encode(".") == [synth.TOK_DOT]
    with pytest.raises(RuntimeError, match="too long"):
        tk.tokenize(["word " * 100])
    assert int(tk.tokenize(["word " * 100], truncate=True)[0, -1]) == synth.EOT_ID


def test_tokenizer_synthetic_table(tmp_path):
    from ovmr_amd.tokenizer import BPETokenizer
    p = str(tmp_path / "bpe.txt.gz")
    make_synthetic_bpe(p)
    tk = BPETokenizer(p)
    t = tk.tokenize(["a sea_horse.".replace("_", " "), "a ."])
    assert t.shape == (2, 77) and int(t[0, 0]) == synth.SOT_ID and int(t.max()) == synth.EOT_ID
    assert (t[1].argmax() == (t[1] == synth.EOT_ID).nonzero()[0, 0])
    with pytest.raises(FileNotFoundError):
        BPETokenizer(str(tmp_path / "missing.gz"))


def test_checkpoint_roundtrip(tmp_path):
    from ovmr_amd import checkpoint
    spec = synth.SPECS["tiny"]
    sd = {k: torch.from_numpy(v) for k, v in synth.prompt_learner_state_dict(spec, 2, 3).items()}
    extra = dict(sd, token_prefix=torch.zeros(1), token_suffix=torch.zeros(1))
    extra = {"module." + k: v for k, v in extra.items()}               # DataParallel prefix is stripped
    f = checkpoint.save_prompt_learner_state(extra, str(tmp_path), 30)
    assert f.endswith("prompt_learner/model.pth.tar-30")
    assert open(tmp_path / "prompt_learner" / "checkpoint").read().strip() == "model.pth.tar-30"
    for epoch in (30, None):                                           # explicit epoch / pointer file
        got = checkpoint.load_prompt_learner_state(str(tmp_path), epoch)
        assert sorted(got) == sorted(sd) and all(torch.equal(got[k], sd[k]) for k in sd)
    with pytest.raises(FileNotFoundError, match="Model not found"):
        checkpoint.load_prompt_learner_state(str(tmp_path), 7)
    # CLIP weights saved as a plain state dict (the TorchScript branch needs OpenAI's archive)
    clip_sd = {k: torch.from_numpy(v) for k, v in synth.clip_state_dict(spec, 1).items()}
    clip_sd["input_resolution"] = torch.tensor(32)
    torch.save(clip_sd, tmp_path / "clip.pt")
    got = checkpoint.load_clip_state_dict(str(tmp_path / "clip.pt"))
    assert "input_resolution" not in got and torch.equal(got["visual.proj"], clip_sd["visual.proj"])
    from ovmr_amd.modules import infer_spec
    s = infer_spec(got)
    assert (s.vision_width, s.vision_layers, s.vision_patch_size, s.image_resolution, s.embed_dim,
            s.transformer_width, s.transformer_layers) == (128, 2, 16, 32, 128, 128, 2)


def test_evaluator_matches_sklearn(tmp_path):
    from sklearn.metrics import f1_score
    from ovmr_amd.evaluator import Classification
    rng = np.random.default_rng(0)
    C, n = 7, 500
    gt = rng.integers(0, C - 1, n)                       # class 6 never appears in y_true
    logits = rng.normal(size=(n, C)).astype(np.float32)
    logits[np.arange(n), gt] += 1.0
    ev = Classification(C, device="cpu")
    for s in range(0, n, 64):
        ev.process(torch.from_numpy(logits[s:s + 64]), torch.from_numpy(gt[s:s + 64]))
    res = ev.evaluate(str(tmp_path))
    pred = logits.argmax(1)
    assert res["accuracy"] == pytest.approx(100.0 * (pred == gt).mean())
    assert res["macro_f1"] == pytest.approx(100.0 * f1_score(gt, pred, average="macro", labels=np.unique(gt)))
    rows = open(tmp_path / "f1_per_class.csv").read().strip().split("\n")
    assert rows[0] == "Label,F1" and len(rows) == 1 + len(np.unique(gt))
    per = 100.0 * f1_score(gt, pred, average=None, labels=np.unique(gt))
    assert float(rows[1].split(",")[1]) == pytest.approx(per[0])
    assert open(tmp_path / "acc_per_class.csv").readline().strip() == "Label,Acc"


def test_runner_helpers(tmp_path):
    from PIL import Image
    from ovmr_amd import cli
    rng = np.random.default_rng(1)
    for split, n in (("train", 3), ("val", 2)):
        for folder in ("n02", "n01"):
            d = tmp_path / split / folder
            d.mkdir(parents=True)
            for i in range(n):
                Image.fromarray(rng.integers(0, 255, (40 + 10 * i, 70, 3), dtype=np.uint8)).save(d / f"im{i}.png")
    (tmp_path / "classnames.txt").write_text("n01 tench fish\nn02 goldfish\n")
    folders, items = cli.list_split(str(tmp_path), "train")
    assert folders == ["n01", "n02"] and [l for _, l in items] == [0, 0, 0, 1, 1, 1]
    assert cli.read_classnames(str(tmp_path), folders) == ["tench fish", "goldfish"]
    ex = cli.exemplar_items(items, 2)
    assert [l for _, l in ex] == [0, 0, 1, 1]
    over = cli.exemplar_items(items, 5)                 # more shots than images: filled up with replacement (samplers.py:148-149)
    assert [l for _, l in over] == [0] * 5 + [1] * 5 and set(over) == set(items)
    batches = list(cli.FolderLoader(ex, 2, 32))
    assert len(batches) == 2 and batches[0]["img"].shape == (2, 3, 32, 32) and batches[0]["label"].tolist() == [0, 0]
    x = cli.test_transform(Image.fromarray(np.full((50, 80, 3), 128, dtype=np.uint8)), 32)
    expect = (128 / 255.0 - np.array(cli.PIXEL_MEAN)) / np.array(cli.PIXEL_STD)
    np.testing.assert_allclose(x.mean(dim=(1, 2)).numpy(), expect, atol=1e-5)
    a = cli.parse(["--root", "r", "--clip-weights", "w.pt", "--eval-only", "--load-epoch", "30", "--eval_tau", "5",
                   "DATASET.NUM_SHOTS", "8"])
    assert a.load_epoch == 30 and a.eval_tau == 5.0 and a.opts == ["DATASET.NUM_SHOTS", "8"]
    # the reference's defaults (train.py:183-255), not this runner's old ones
    assert a.eval_mode == "multimodal" and a.n_ctx is None and a.seed == -1 and a.trainer == "" and a.output_dir == ""


# the keys of the reference's own YAML files (configs/trainers/MM_CLS_OP/vit_b16_c4_ep50_imagenet21k_pretrain.yaml,
# configs/datasets/imagenet.yaml): the parser must take such a file as it is
TRAINER_YAML = """
DATALOADER:
  TRAIN_X:
    BATCH_SIZE: 1536
    SAMPLER: "RandomClassSampler"
    N_INS: 8
  TEST:
    BATCH_SIZE: 256
    N_INS: 16
  NUM_WORKERS: 8
  K_TRANSFORMS: 1
INPUT:
  SIZE: (224, 224)
  INTERPOLATION: "bicubic"
  PIXEL_MEAN: [0.48145466, 0.4578275, 0.40821073]
  PIXEL_STD: [0.26862954, 0.26130258, 0.27577711]
  RRCROP_SCALE: (0.25, 1.0)
  TRANSFORMS: ["random_resized_crop", "random_flip", 'colorjitter', 'gaussian_noise', "normalize"]
OPTIM:
  NAME: "adam"
  LR: 0.0002
  MAX_EPOCH: 30
  WARMUP_CONS_LR: 1e-5
TRAIN:
  PRINT_FREQ: 10
TEST:
  NO_TEST: True
MODEL:
  BACKBONE:
    NAME: "ViT-B/16"
TRAINER:
  COCOOP:
    N_CTX: 2
    CTX_INIT: " ?"
    PREC: "fp16"
"""


def test_setup_cfg_layers_as_train_py(tmp_path):
    """ovmr_amd.config.setup_cfg against train.py:134-155: defaults < dataset config file < config file < flags < trailing opts."""
    from types import SimpleNamespace as NS
    from ovmr_amd import config
    (tmp_path / "trainer.yaml").write_text(TRAINER_YAML)
    (tmp_path / "dataset.yaml").write_text('DATASET:\n  NAME: "ImageNet"\n  NUM_SHOTS: 4\nDATALOADER:\n  TEST:\n    BATCH_SIZE: 64\n')
    base = dict(root="", output_dir="", seed=-1, trainer="", backbone="", init_weight="", n_ctx=None, eval_mode="multimodal",
                eval_tau=None, dataset_config_file="", config_file="", opts=[])
    # 1. nothing given: the reference's defaults
    c = config.setup_cfg(NS(**base))
    assert (c.EVAL_MODE, c.EVAL_TAU, c.TRAINER.COCOOP.N_CTX, c.DATALOADER.TEST.BATCH_SIZE, c.DATASET.NUM_SHOTS, c.SEED) == ("multimodal", 10, 16, 32, -1, -1)
    assert c.DATASET.SUBSAMPLE_CLASSES == "all" and c.INPUT.INTERPOLATION == "bilinear" and c.INPUT.TRANSFORMS == ()
    # 2. files: the config file wins over the dataset file, both over the defaults; literal strings are decoded
    c = config.setup_cfg(NS(**{**base, "dataset_config_file": str(tmp_path / "dataset.yaml"), "config_file": str(tmp_path / "trainer.yaml")}))
    assert c.DATALOADER.TEST.BATCH_SIZE == 256 and c.DATASET.NUM_SHOTS == 4 and c.DATASET.NAME == "ImageNet"
    assert c.INPUT.SIZE == (224, 224) and c.INPUT.INTERPOLATION == "bicubic" and "normalize" in c.INPUT.TRANSFORMS
    assert c.TRAINER.COCOOP.N_CTX == 2 and c.MODEL.BACKBONE.NAME == "ViT-B/16" and c.DATALOADER.NUM_WORKERS == 8
    assert c.INPUT.PIXEL_MEAN == [0.48145466, 0.4578275, 0.40821073] and c.DATALOADER.TRAIN_X.N_INS == 8
    # 3. + 4. flags (only truthy ones, reset_cfg), then the trailing opts -- generate_classifier.sh's own command line
    c = config.setup_cfg(NS(**{**base, "config_file": str(tmp_path / "trainer.yaml"), "root": "./data", "seed": 1, "trainer": "MM_CLS_OP",
                               "output_dir": "out", "eval_mode": "fusion", "eval_tau": 10, "n_ctx": 2,
                               "opts": ["DATASET.NUM_SHOTS", "16", "DATASET.SUBSAMPLE_CLASSES", "new", "DATALOADER.TEST.BATCH_SIZE", "128"]}))
    assert (c.DATASET.ROOT, c.SEED, c.TRAINER.NAME, c.OUTPUT_DIR, c.EVAL_MODE) == ("./data", 1, "MM_CLS_OP", "out", "fusion")
    assert (c.DATASET.NUM_SHOTS, c.DATASET.SUBSAMPLE_CLASSES, c.DATALOADER.TEST.BATCH_SIZE) == (16, "new", 128)
    c = config.setup_cfg(NS(**{**base, "config_file": str(tmp_path / "trainer.yaml"), "n_ctx": 0, "eval_tau": 0, "seed": 0}))
    assert c.TRAINER.COCOOP.N_CTX == 2 and c.EVAL_TAU == 10 and c.SEED == -1                     # `if args.x:` -- falsy flags leave the cfg alone
    # keys that do not exist raise (yacs: "Non-existent config key"), as does this runner's old spelling of the batch size
    for bad in (["TEST.BATCH_SIZE", "6"], ["DATASET.SUBSAMPLE", "base"], ["FOO", "1"]):
        with pytest.raises(KeyError):
            config.setup_cfg(NS(**{**base, "opts": bad}))
    with pytest.raises(ValueError):
        config.setup_cfg(NS(**{**base, "opts": ["DATASET.SUBSAMPLE_CLASSES", "novel"]}))
    with pytest.raises(ValueError):
        config.setup_cfg(NS(**{**base, "opts": ["DATASET.NUM_SHOTS", "many"]}))                  # type mismatch with the default
    with pytest.raises(ValueError):
        config.setup_cfg(NS(**{**base, "opts": ["DATASET.NUM_SHOTS"]}))                          # odd override list
    config.setup_cfg(NS(**{**base, "opts": ["OPTIM.LR", "0.1", "TRAIN.PRINT_FREQ", "5"]}))        # known, ignored on this path


def test_subsample_classes_base_new(tmp_path):
    """DATASET.SUBSAMPLE_CLASSES as datasets/oxford_pets.py:141-202 on a 5-class folder: `base` = the first ceil(5/2) = 3 labels,
    `new` = the other 2, relabelled from 0 in sorted order, in the exemplar AND the test items; class names follow the labels."""
    from types import SimpleNamespace as NS
    from PIL import Image
    from ovmr_amd import cli, config
    rng = np.random.default_rng(0)
    names = {"n05": "echidna", "n01": "tench", "n03": "stingray", "n02": "goldfish", "n04": "hen"}
    for split, n in (("train", 4), ("val", 2)):
        for folder in names:
            d = tmp_path / split / folder
            d.mkdir(parents=True)
            for i in range(n):
                Image.fromarray(rng.integers(0, 255, (20, 24, 3), dtype=np.uint8)).save(d / f"im{i}.png")
    (tmp_path / "classnames.txt").write_text("".join(f"{k} {v}\n" for k, v in names.items()))
    base = dict(root=str(tmp_path), output_dir="", seed=1, trainer="MM_CLS_OP", backbone="", init_weight="", n_ctx=2, eval_mode="fusion",
                eval_tau=10, dataset_config_file="", config_file="")
    got = {}
    for sub in ("all", "base", "new"):
        cfg = config.setup_cfg(NS(**base, opts=["DATASET.NUM_SHOTS", "3", "DATASET.SUBSAMPLE_CLASSES", sub]))
        got[sub] = cli.build_splits(cfg)
    ordered = ["tench", "goldfish", "stingray", "hen", "echidna"]                # sorted folders n01..n05
    for sub, want in (("all", ordered), ("base", ordered[:3]), ("new", ordered[3:])):
        classnames, ex, test = got[sub]
        C = len(want)
        assert classnames == want
        assert [l for _, l in ex] == [c for c in range(C) for _ in range(3)]                       # 3 consecutive rows per class, labels 0..C-1
        assert sorted({l for _, l in test}) == list(range(C)) and len(test) == 2 * C
        folders_of = lambda items, lab: {p.split("/")[-2] for p, l in items if l == lab}
        for c, nm in enumerate(want):                                                            # every label points at its own folder, in both splits
            f = [k for k, v in names.items() if v == nm][0]
            assert folders_of(ex, c) == {f} and folders_of(test, c) == {f}
    # the draw precedes the subsampling: `base` / `new` exemplars are the same images the all-classes job draws
    all_ex = {p for p, _ in got["all"][1]}
    assert {p for p, _ in got["base"][1]} | {p for p, _ in got["new"][1]} == all_ex
    # the function itself, on labels with a gap and an extra element per item
    a = [("x", 7, "seven"), ("y", 2, "two"), ("z", 4, "four")]
    b = [("t", 4, "four"), ("u", 7, "seven")]
    ra, rb = config.subsample_classes(a, b, subsample="base")                                    # sorted labels 2, 4 | 7
    assert ra == [("y", 0, "two"), ("z", 1, "four")] and rb == [("t", 1, "four")]
    ra, rb = config.subsample_classes(a, b, subsample="new")
    assert ra == [("x", 0, "seven")] and rb == [("u", 0, "seven")]
    with pytest.raises(AssertionError):
        config.subsample_classes(a, subsample="novel")
    # an exemplar list reproduces a given few-shot set (labels before subsampling)
    lst = tmp_path / "few.txt"
    lst.write_text("".join(f"{p} {l + 3}\n" for p, l in got["new"][1]))
    cfg = config.setup_cfg(NS(**base, opts=["DATASET.NUM_SHOTS", "3", "DATASET.SUBSAMPLE_CLASSES", "all"]))
    with pytest.raises(SystemExit):
        cli.build_splits(cfg, exemplar_list=str(lst))                                           # classes without exemplars
    # ... and is validated before anything is decoded: forward_prompt groups rows by POSITION, so a class with more than NUM_SHOTS rows
    # would shift every later class group; out-of-range labels and missing files fail with the offending line
    full = got["all"][1]
    ok = tmp_path / "ok.txt"
    ok.write_text("".join(f"{p} {l}\n" for p, l in full))
    names_all, ex_all, _ = cli.build_splits(cfg, exemplar_list=str(ok))
    assert names_all == ordered and [l for _, l in ex_all] == [c for c in range(5) for _ in range(3)]
    short = tmp_path / "short.txt"                                                               # two rows for class 1: filled up to three of ITS images
    short.write_text("".join(f"{p} {l}\n" for i, (p, l) in enumerate(full) if i != 4))
    _, ex_short, _ = cli.build_splits(cfg, exemplar_list=str(short))
    assert [l for _, l in ex_short] == [c for c in range(5) for _ in range(3)]
    assert {p.split("/")[-2] for p, l in ex_short if l == 1} == {"n02"}
    for bad_lines, what in ((full + [full[4]], "more than DATASET.NUM_SHOTS"), (full[:-1] + [(full[-1][0], 5)], "outside"),
                            (full[:-1] + [(full[-1][0], -1)], "outside"), (full[:-1] + [(str(tmp_path / "nope.png"), 4)], "does not exist")):
        bad = tmp_path / "bad.txt"
        bad.write_text("".join(f"{p} {l}\n" for p, l in bad_lines))
        with pytest.raises(SystemExit, match=what):
            cli.build_splits(cfg, exemplar_list=str(bad))
    (tmp_path / "bad.txt").write_text("just-a-path-without-label\n")
    with pytest.raises(SystemExit, match="expected"):
        cli.build_splits(cfg, exemplar_list=str(tmp_path / "bad.txt"))
    with pytest.raises(ValueError, match="more than DATASET.NUM_SHOTS"):
        cli.layout_exemplars(full + [full[0]], 3, 1)


def test_loader_list_form_k_transforms():
    """K_TRANSFORMS > 1 hands the model a LIST of image tensors (trainers/mm_classifier_one_prompt.py:229-234):
    unsqueeze(1), cat on dim 1, flatten(0, 1) -> the k views of an image are consecutive rows."""
    from ovmr_amd.modules import CustomCLIP
    a = torch.arange(3 * 2, dtype=torch.float32).reshape(3, 2)          # 3 images, view 0
    b = a + 100                                                          # view 1
    out = CustomCLIP._batch_images({"img": [a, b]}, "cpu")
    assert out.shape == (6, 2)
    assert torch.equal(out, torch.stack([a, b], dim=1).flatten(0, 1))
    assert torch.equal(out[0], a[0]) and torch.equal(out[1], b[0]) and torch.equal(out[2], a[1])
    single = CustomCLIP._batch_images({"img": a}, "cpu")
    assert single is a


CKPT = os.path.join(REPO, "tests", "golden", "ckpt")


def test_reference_written_checkpoint_files():
    """SURVEY 8f-3 on files the REFERENCE's own writers produced (tests/golden/gen_checkpoints.py): a TorchScript archive
    of the reference CLIP module (the format of OpenAI's ViT-B-16.pt) and a Dassl save_checkpoint() directory."""
    from ovmr_amd import checkpoint
    from ovmr_amd.modules import infer_spec
    spec = synth.SPECS["micro"]
    exp = np.load(os.path.join(CKPT, "micro_expected.npz"))
    sd = checkpoint.load_clip_state_dict(os.path.join(CKPT, "micro_clip_jit.pt"))       # the torch.jit.load branch
    assert sorted(sd) == sorted(k for k in exp["clip_keys"] if k not in ("input_resolution", "context_length", "vocab_size"))
    assert {"input_resolution", "context_length", "vocab_size"} <= set(exp["clip_keys"].tolist())
    got = infer_spec(sd)
    for f in ("embed_dim", "image_resolution", "vision_layers", "vision_width", "vision_patch_size", "context_length",
              "vocab_size", "transformer_width", "transformer_heads", "transformer_layers"):
        assert getattr(got, f) == getattr(spec, f), f
    ref = synth.clip_state_dict(spec, 11, jitter=True)
    halved = 0
    for k, v in ref.items():
        t = sd[k]
        if t.dtype == torch.float16:                         # convert_weights() halves conv / linear / attention / projections
            halved += 1
            assert torch.equal(t, torch.from_numpy(v).half()), k
        else:
            assert t.dtype == torch.float32 and torch.equal(t, torch.from_numpy(v)), k
    assert halved > 10 and sd["visual.positional_embedding"].dtype == torch.float32 and sd["ln_final.weight"].dtype == torch.float32

    pl_ref = synth.prompt_learner_state_dict(spec, 2, 11, True)
    raw = torch.load(os.path.join(CKPT, "prompt_learner", "model.pth.tar-30"), map_location="cpu", weights_only=False)
    assert sorted(raw) == ["epoch", "optimizer", "scheduler", "state_dict", "val_result"] and raw["epoch"] == 30
    assert "token_prefix" in raw["state_dict"] and "token_suffix" in raw["state_dict"]
    for epoch in (30, None):                                 # explicit epoch / `checkpoint` pointer file
        pl = checkpoint.load_prompt_learner_state(CKPT, epoch)
        assert sorted(pl) == sorted(pl_ref)
        assert all(torch.equal(pl[k], torch.from_numpy(pl_ref[k])) for k in pl_ref)
    with pytest.raises(FileNotFoundError, match="Model not found"):
        checkpoint.load_prompt_learner_state(CKPT, 7)


class _SchedulerLike:
    """Stands for the objects Dassl pickles next to the weights (the warm-up scheduler and its `successor`)."""
    def __init__(self):
        self.last_epoch = 30


_RAN = []


def _note(msg):
    _RAN.append(msg)


class _Payload:
    """What a hostile file would carry: unpickling it calls a function of the loader's process."""
    def __reduce__(self):
        return (_note, ("code from the file ran",))


def test_object_checkpoints_load_without_running_code(tmp_path, monkeypatch):
    """A checkpoint that pickles OBJECTS next to the weights (what the reference's save_checkpoint writes: scheduler instances --
    the very file `--model-dir ... --load-epoch 30` names) loads by default through the restricted unpickler: same tensors, the
    objects replaced by inert placeholders, and a callable smuggled into the file is NOT called.  OVMR_TRUSTED_CHECKPOINTS=1 switches
    to torch's full unpickling (then the file's code does run); a tensors-only file takes weights_only either way."""
    from ovmr_amd import checkpoint
    pl = {k: torch.from_numpy(v) for k, v in synth.prompt_learner_state_dict(synth.SPECS["micro"], 2, 11, True).items()}
    d = tmp_path / "prompt_learner"
    d.mkdir()
    torch.save({"state_dict": pl, "epoch": 30, "scheduler": _SchedulerLike(), "optimizer": {"state": {0: {"exp_avg": torch.ones(3)}}, "hook": _Payload()},
                "val_result": np.float64(0.5)}, d / "model.pth.tar-30")
    monkeypatch.delenv("OVMR_TRUSTED_CHECKPOINTS", raising=False)
    _RAN.clear()
    got, epoch, path = checkpoint.load_prompt_learner_checkpoint(str(tmp_path), 30)
    assert epoch == 30 and path.endswith("model.pth.tar-30") and not _RAN
    assert sorted(got) == sorted(pl) and all(torch.equal(got[k], pl[k]) and got[k].dtype == pl[k].dtype for k in pl)
    raw = checkpoint._torch_load(str(d / "model.pth.tar-30"))
    assert isinstance(raw["scheduler"], checkpoint._Inert) and isinstance(raw["optimizer"]["hook"], checkpoint._Inert)
    assert torch.equal(raw["optimizer"]["state"][0]["exp_avg"], torch.ones(3))
    monkeypatch.setenv("OVMR_TRUSTED_CHECKPOINTS", "1")
    with pytest.warns(UserWarning, match="full unpickling"):
        got = checkpoint.load_prompt_learner_state(str(tmp_path), 30)
    assert all(torch.equal(got[k], pl[k]) for k in pl) and _RAN == ["code from the file ran"]
    monkeypatch.delenv("OVMR_TRUSTED_CHECKPOINTS")
    torch.save({"state_dict": pl, "epoch": 31, "scheduler": None}, d / "model.pth.tar-31")
    assert sorted(checkpoint.load_prompt_learner_state(str(tmp_path), 31)) == sorted(pl)
    # the reference-written golden checkpoint (tests/golden/gen_checkpoints.py: Dassl's save_checkpoint) loads without the opt-in
    ref_written = checkpoint.load_prompt_learner_state(CKPT, 30)
    assert "cls_token" in ref_written and all(isinstance(v, torch.Tensor) for v in ref_written.values())


def test_default_tokenizer_from_environment(tmp_path, monkeypatch):
    """CustomCLIP(cfg, classnames, clip_model) -- the reference's three-argument form -- finds its tokenizer through
    OVMR_BPE_PATH; without a table the error says what to do."""
    from ovmr_amd import modules
    modules._DEFAULT_TOKENIZER.clear()
    monkeypatch.delenv("OVMR_BPE_PATH", raising=False)
    monkeypatch.chdir(tmp_path)
    with pytest.raises(FileNotFoundError, match="OVMR_BPE_PATH"):
        modules.default_tokenizer()
    p = str(tmp_path / "bpe.txt.gz")
    make_synthetic_bpe(p)
    monkeypatch.setenv("OVMR_BPE_PATH", p)
    tk = modules.default_tokenizer()
    assert tk is modules.default_tokenizer()
    t = modules.tokenize(["a sea horse."], tk)
    assert t.shape == (1, 77) and int(t[0, 0]) == synth.SOT_ID and int(t.max()) == synth.EOT_ID
    # ./clip/bpe_simple_vocab_16e6.txt.gz of a reference checkout
    modules._DEFAULT_TOKENIZER.clear()
    monkeypatch.delenv("OVMR_BPE_PATH")
    (tmp_path / "clip").mkdir()
    os.replace(p, tmp_path / "clip" / "bpe_simple_vocab_16e6.txt.gz")
    assert modules.default_tokenizer().encode("a") == tk.encode("a")
    modules._DEFAULT_TOKENIZER.clear()


def test_fewshot_draw_matches_the_reference():
    """cli.fewshot_items against what the reference's own generate_fewshot_dataset picked after random.seed(seed)
    (tests/golden/gen_fewshot.py -> fewshot.npz): the same items in the same order, for three seeds and two shot counts, incl.
    classes with fewer images than shots (kept whole).  exemplar_items lays them out as S consecutive rows per class."""
    from ovmr_amd import cli
    g = np.load(os.path.join(REPO, "tests", "golden", "fewshot.npz"))
    items = list(zip(g["paths"].tolist(), [int(x) for x in g["labels"]]))
    for seed in (1, 2, 3):
        for shots in (4, 8):
            got = [p for p, _ in cli.fewshot_items(items, shots, seed)]
            assert got == g[f"picked_seed{seed}_shots{shots}"].tolist(), (seed, shots)
            ex = cli.exemplar_items(items, shots, seed)
            labels = [l for _, l in ex]
            assert len(ex) == 7 * shots and all(len(set(labels[i:i + shots])) == 1 for i in range(0, len(ex), shots))
            assert set(p for p, _ in ex) == set(got)                       # the fill only repeats drawn images
    assert [p for p, _ in cli.fewshot_items(items, 4, 1)] != [p for p, _ in cli.fewshot_items(items, 4, 2)]


def test_resize_tables_reproduce_pil_bit_for_bit():
    """ovmr_amd/resize.py restates Pillow's 8-bit resampling (Resample.c: precompute_coeffs, normalize_coeffs_8bpc, the horizontal and
    the vertical integer pass) for Resize(size) + CenterCrop(size); the numpy form of the two passes over ITS tables must equal
    PIL.Image.resize + crop on every byte -- bicubic and bilinear, up- and down-scaling, odd aspect ratios, one-pixel images.
    (The GPU kernel runs the same two passes over the same tables: tests/test_hip_loader.py holds it to PIL directly.)"""
    from PIL import Image
    from ovmr_amd import resize
    from ovmr_amd._decode_worker import load_u8
    rng = np.random.default_rng(5)
    sizes = [(500, 375), (375, 500), (224, 224), (224, 300), (100, 37), (1200, 224), (1, 1), (2, 700), (1600, 1200), (223, 225)]
    sizes += [(int(rng.integers(3, 1000)), int(rng.integers(3, 1000))) for _ in range(30)]
    for w, h in sizes:
        img = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        for interp, R in (("bicubic", 224), ("bilinear", 224), ("bicubic", 96)):
            want = load_u8(Image.fromarray(img), R, False, interp)
            got = resize.resize_crop_reference(img, R, interp)
            assert np.array_equal(want, got), f"{w} x {h} -> {R} ({interp})"
    p = resize.plan(500, 375, 224)
    assert p["ksize_h"] == p["ksize_v"] == 9 and p["y0"] == 0 and p["ny"] == 375 and p["table"].dtype == np.int32
    assert resize.resized_size(500, 375, 224) == (298, 224) and resize.resized_size(375, 500, 224) == (224, 298)
    with pytest.raises(ValueError):
        resize.plan(10, 10, 224, "nearest")


def test_generate_classifier_script_maps_the_reference_arguments(tmp_path):
    """scripts/generate_classifier.sh takes the reference script's seven positional arguments (scripts/mm_cls/generate_classifier.sh:2-8)
    and builds the reference's command line (:30-44) for `python -m ovmr_amd.cli`, which `cli.parse` + `config.setup_cfg` accept."""
    import subprocess
    from ovmr_amd import cli
    script = os.path.join(REPO, "scripts", "generate_classifier.sh")
    env = dict(os.environ, DRY_RUN="1", CLIP_WEIGHTS="/w/ViT-B-16.pt", OVMR_REF="/ref", DIR=str(tmp_path / "out"))
    r = subprocess.run(["bash", script, "imagenet", "1", "all", "2", "fusion", "10", "0"], env=env, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    words = r.stdout.split("\n")[:-1]
    assert words[:3] == ["python", "-m", "ovmr_amd.cli"]
    argv = words[3:]
    flat = " ".join(argv)
    for piece in ("--root ./data", "--seed 1", "--trainer MM_CLS_OP", "--dataset-config-file /ref/configs/datasets/imagenet.yaml",
                  "--config-file /ref/configs/trainers/MM_CLS_OP/vit_b16_c4_ep50_imagenet21k_pretrain.yaml", "--model-dir ./checkpoints",
                  "--load-epoch 30", "--eval_mode fusion", "--eval_tau 10", "--n_ctx 2", "--eval-only", "--clip-weights /w/ViT-B-16.pt",
                  "DATASET.NUM_SHOTS 16", "DATASET.SUBSAMPLE_CLASSES all"):
        assert piece in flat, piece
    a = cli.parse(argv)
    assert a.eval_only and a.trainer == "MM_CLS_OP" and a.load_epoch == 30 and a.n_ctx == 2 and a.eval_mode == "fusion" and a.eval_tau == 10
    assert a.opts == ["DATASET.NUM_SHOTS", "16", "DATASET.SUBSAMPLE_CLASSES", "all"]
    r = subprocess.run(["bash", script, "imagenet", "1", "new", "2", "fusion", "10", "0,1,2,3"], env=env, capture_output=True, text=True)
    assert r.returncode == 0 and r.stdout.startswith("ranks 4") and "DATASET.SUBSAMPLE_CLASSES\nnew" in r.stdout
    r = subprocess.run(["bash", script, "imagenet", "1"], env=env, capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr
    env.pop("CLIP_WEIGHTS")
    r = subprocess.run(["bash", script, "imagenet", "1", "all", "2", "fusion", "10", "0"], env=env, capture_output=True, text=True)
    assert r.returncode != 0 and "CLIP_WEIGHTS" in r.stderr


def test_eval_ovmr_script_maps_the_reference_directories(tmp_path):
    """scripts/eval_ovmr.sh = the reference's scripts/mm_cls/eval_ovmr.sh:19-47: the same seven arguments and train.py flags as the generation
    script, the generator's checkpoint read from the base-to-new training output (:27) and the results written under
    base2new/test_<SUB>_<MODE>_tau<TAU>/<DATASET>/shots_16/MM_CLS_OP/<CFG>/seed<SEED> (:28); an existing result directory skips the job."""
    import subprocess
    from ovmr_amd import cli
    script = os.path.join(REPO, "scripts", "eval_ovmr.sh")
    env = dict(os.environ, DRY_RUN="1", CLIP_WEIGHTS="/w/ViT-B-16.pt", OVMR_REF="/ref")
    for k in ("DIR", "MODEL_DIR", "CFG", "SHOTS"):
        env.pop(k, None)
    r = subprocess.run(["bash", script, "caltech101", "3", "new", "2", "fusion", "10", "0"], env=env, capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0, r.stderr
    words = r.stdout.split("\n")[:-1]
    assert words[:3] == ["python", "-m", "ovmr_amd.cli"]
    a = cli.parse(words[3:])
    cfgname = "vit_b16_c4_ep50_imagenet21k_pretrain"
    assert a.model_dir == f"output_ovmr/base2new/train_base/imagenet_21k_P/shots_64/MM_CLS_OP/{cfgname}/seed1"
    assert a.output_dir == f"output_ovmr/base2new/test_new_fusion_tau10/caltech101/shots_16/MM_CLS_OP/{cfgname}/seed3"
    assert a.eval_only and a.load_epoch == 30 and a.seed == 3 and a.n_ctx == 2 and a.eval_mode == "fusion" and a.eval_tau == 10
    assert a.dataset_config_file == "/ref/configs/datasets/caltech101.yaml" and a.config_file == f"/ref/configs/trainers/MM_CLS_OP/{cfgname}.yaml"
    assert a.opts == ["DATASET.NUM_SHOTS", "16", "DATASET.SUBSAMPLE_CLASSES", "new"]
    # the environment still wins over the script's directories; several GPUs -> one rank per GPU
    r = subprocess.run(["bash", script, "caltech101", "3", "base", "2", "vision", "5", "0,1"], env=dict(env, MODEL_DIR="/ck", SHOTS="8"),
                       capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0 and r.stdout.startswith("ranks 2")
    a = cli.parse(r.stdout.split("\n")[4:-1])
    assert a.model_dir == "/ck" and "/shots_8/" in a.output_dir and "test_base_vision_tau5" in a.output_dir and a.opts[:2] == ["DATASET.NUM_SHOTS", "8"]
    # an existing result directory skips the job (reference :30-31) -- without DRY_RUN, nothing else is started
    done = tmp_path / "output_ovmr/base2new/test_new_fusion_tau10/caltech101/shots_16/MM_CLS_OP" / cfgname / "seed3"
    done.mkdir(parents=True)
    env2 = {k: v for k, v in env.items() if k != "DRY_RUN"}
    r = subprocess.run(["bash", script, "caltech101", "3", "new", "2", "fusion", "10", "0"], env=env2, capture_output=True, text=True, cwd=str(tmp_path))
    assert r.returncode == 0 and "results exist" in r.stdout
    r = subprocess.run(["bash", script, "caltech101"], env=env, capture_output=True, text=True)
    assert r.returncode == 2 and "usage" in r.stderr


def test_upload_ring_asks_dev_shm_before_it_allocates(monkeypatch):
    """The pipelined loader's upload ring lives in /dev/shm (576 MiB at batch 256 x 3 slots x 768 KiB).  SharedMemory only ftruncate()s,
    which tmpfs grants beyond its free space -- the workers would find out with SIGBUS.  The pool asks for the free space first: too
    little halves the share per image down to one crop, and below that the loader says what to change (no worker is started)."""
    from ovmr_amd import loader
    free = {"bytes": 40 << 20}
    monkeypatch.setattr(loader, "_shm_free_bytes", lambda: free["bytes"])
    seen = []
    real_pool = loader._Pool

    class Probe(real_pool):
        def __init__(self, key):
            seen.append(key[4])                                          # the share per image this attempt asks for
            real_pool.__init__(self, key)                                # raises OSError before anything is allocated or started

    monkeypatch.setattr(loader, "_Pool", Probe)
    ld = loader.PipelinedFolderLoader([("a.jpg", 0)] * 8, 256, 224, workers=2, prefetch=3, device="cpu")
    with pytest.raises(OSError, match="does not fit"):
        ld._pool()
    crop = 224 * 224 * 3
    assert seen[0] == 768 * 1024 and seen[-1] == crop and all(a > b for a, b in zip(seen, seen[1:]))
    assert not loader._POOLS
    # the message names the ring and /dev/shm
    with pytest.raises(OSError, match="/dev/shm"):
        real_pool((2, 3, 256, 224, crop, 8, "bicubic", False))
