"""The reference's configuration, for the keys the hot path reads -- without yacs / Dassl.

`train.py:134-150` builds its cfg in four layers, later ones winning:
  1. the defaults (Dassl.pytorch/dassl/config/defaults.py + `extend_cfg`, train.py:96-131),
  2. `--dataset-config-file` (configs/datasets/*.yaml),
  3. `--config-file` (configs/trainers/MM_CLS_OP/*.yaml),
  4. the command-line flags `reset_cfg` copies (train.py:41-94: only flags with a truthy value), then the trailing
     `KEY VALUE` opts (`cfg.merge_from_list`).
`setup_cfg` below does the same on a plain namespace tree.  Like yacs it refuses keys that do not exist; unlike yacs it knows two
classes of keys: HONOURED (they change the class set, the batches or the arithmetic of this path, table `DEFAULTS`) and
IGNORED (training, optimiser, augmentation, logging: accepted so that the reference's YAML files load unchanged, never read).
A key in neither class raises KeyError instead of being dropped silently.
"""
from __future__ import annotations

import ast
import copy
from types import SimpleNamespace
from typing import Any, Dict, Iterable, List, Sequence

# key -> default.  Dassl defaults.py values unless noted.
DEFAULTS: Dict[str, Any] = {
    "OUTPUT_DIR": "./output",
    "SEED": -1,
    "EVAL_MODE": "multimodal",                       # train.py:129
    "EVAL_TAU": 10,                                  # train.py:130
    "DATASET.ROOT": "",
    "DATASET.NAME": "",
    "DATASET.NUM_SHOTS": -1,
    "DATASET.SUBSAMPLE_CLASSES": "all",              # train.py:123  (all | base | new)
    "DATALOADER.NUM_WORKERS": 4,
    "DATALOADER.K_TRANSFORMS": 1,
    "DATALOADER.TEST.BATCH_SIZE": 32,
    "DATALOADER.TRAIN_X.BATCH_SIZE": 32,             # read by CustomCLIP.__init__ (trainers/mm_classifier_one_prompt.py:189-190), unused in eval
    "DATALOADER.TRAIN_X.N_INS": 16,
    "INPUT.SIZE": (224, 224),
    "INPUT.INTERPOLATION": "bilinear",
    "INPUT.PIXEL_MEAN": [0.485, 0.456, 0.406],
    "INPUT.PIXEL_STD": [0.229, 0.224, 0.225],
    "INPUT.TRANSFORMS": (),                          # the test transform normalises only if "normalize" is listed (transforms.py:514-518)
    "MODEL.BACKBONE.NAME": "",
    "MODEL.INIT_WEIGHTS": "",
    "TRAINER.NAME": "",
    "TRAINER.COCOOP.N_CTX": 16,                      # train.py:118
    "TRAINER.COCOOP.CTX_INIT": "",
    "TRAINER.COCOOP.PREC": "fp16",
    "TEST.SPLIT": "test",
}

# accepted and never read on this path (prefix match on the dotted key)
IGNORED_PREFIXES = ("OPTIM.", "TRAIN.", "TEST.EVALUATOR", "TEST.PER_CLASS_RESULT", "TEST.COMPUTE_CMAT", "TEST.NO_TEST", "TEST.FINAL_MODEL",
                    "DATALOADER.TRAIN_X.", "DATALOADER.TRAIN_U.", "DATALOADER.TEST.SAMPLER",
                    "DATALOADER.TEST.N_INS", "DATALOADER.RETURN_IMG0", "INPUT.", "MODEL.HEAD.", "MODEL.BACKBONE.PRETRAINED",
                    "TRAINER.COOP.", "DATASET.SOURCE_DOMAINS", "DATASET.TARGET_DOMAINS", "DATASET.VAL_PERCENT", "DATASET.STL10_FOLD",
                    "DATASET.CIFAR_C_TYPE", "DATASET.CIFAR_C_LEVEL", "DATASET.ALL_AS_UNLABELED", "DATASET.NUM_LABELED", "DATASET.REGION_AUG",
                    "VERSION", "RESUME", "USE_CUDA", "VERBOSE", "TEXT_ONLY", "GPU_NUMS", "TASK_ID", "FS_CLASSIFIER", "CLASSIFIER_PARAMETERS",
                    "STAGE_NUM", "USE_CLIP_TEXT")

CHOICES = {"DATASET.SUBSAMPLE_CLASSES": ("all", "base", "new"), "EVAL_MODE": ("text", "vision", "multimodal", "fusion"),
           "INPUT.INTERPOLATION": ("bilinear", "bicubic", "nearest")}


def _decode(value: Any) -> Any:
    """yacs `_decode_cfg_value`: strings that are Python literals become the literal ("(224, 224)" -> (224, 224), "16" -> 16)."""
    if not isinstance(value, str):
        return value
    try:
        return ast.literal_eval(value)
    except (ValueError, SyntaxError):
        return value


def _coerce(key: str, value: Any, default: Any) -> Any:
    """yacs `_check_and_coerce_cfg_value_type` for the types that occur here."""
    if isinstance(default, bool) or default is None:
        return value
    if isinstance(default, (tuple, list)):
        if isinstance(value, (tuple, list)):
            return type(default)(value)
    elif isinstance(default, float) and isinstance(value, int):
        return float(value)
    elif isinstance(default, int) and isinstance(value, float) and key == "EVAL_TAU":
        return value                                   # the reference's flag is an int, the arithmetic takes any real
    elif isinstance(value, type(default)):
        return value
    raise ValueError(f"config key {key}: value {value!r} of type {type(value).__name__} does not match the default's type "
                     f"{type(default).__name__}")


def _flatten(tree: Dict[str, Any], prefix: str = "") -> Iterable:
    for k, v in tree.items():
        if isinstance(v, dict):
            yield from _flatten(v, f"{prefix}{k}.")
        else:
            yield f"{prefix}{k}", v


def _set(flat: Dict[str, Any], key: str, value: Any, source: str) -> None:
    if key in DEFAULTS:
        v = _coerce(key, _decode(value), DEFAULTS[key])
        if key in CHOICES and v not in CHOICES[key]:
            raise ValueError(f"{source}: {key} must be one of {CHOICES[key]}, got {v!r}")
        flat[key] = v
    elif any(key == p or key.startswith(p) for p in IGNORED_PREFIXES):
        pass
    else:
        raise KeyError(f"{source}: non-existent config key {key!r} (the hot path honours {sorted(DEFAULTS)})")


def merge_from_file(flat: Dict[str, Any], path: str) -> None:
    import yaml
    with open(path) as f:
        tree = yaml.safe_load(f) or {}
    if not isinstance(tree, dict):
        raise ValueError(f"{path}: a config file holds a mapping")
    for key, value in _flatten(tree):
        _set(flat, key, value, path)


def merge_from_list(flat: Dict[str, Any], opts: Sequence[str]) -> None:
    opts = list(opts or [])
    if len(opts) % 2:
        raise ValueError(f"override list has odd length: {opts}; it must be a list of KEY VALUE pairs")      # yacs' assertion
    for key, value in zip(opts[0::2], opts[1::2]):
        _set(flat, key, value, "command line")


def to_namespace(flat: Dict[str, Any]) -> SimpleNamespace:
    root = SimpleNamespace()
    for key, value in flat.items():
        node, parts = root, key.split(".")
        for p in parts[:-1]:
            if not hasattr(node, p):
                setattr(node, p, SimpleNamespace())
            node = getattr(node, p)
        setattr(node, parts[-1], copy.copy(value))
    return root


def setup_cfg(args) -> SimpleNamespace:
    """train.py:134-155.  `args` carries the reference's flag names (argparse spelling: root, output_dir, seed, trainer, backbone,
    init_weight, n_ctx, eval_mode, eval_tau, dataset_config_file, config_file, opts); absent attributes count as unset."""
    flat = dict(DEFAULTS)
    g = lambda name: getattr(args, name, None)
    if g("dataset_config_file"):
        merge_from_file(flat, args.dataset_config_file)          # 1.
    if g("config_file"):
        merge_from_file(flat, args.config_file)                  # 2.
    for flag, key in (("root", "DATASET.ROOT"), ("output_dir", "OUTPUT_DIR"), ("seed", "SEED"), ("trainer", "TRAINER.NAME"),
                      ("backbone", "MODEL.BACKBONE.NAME"), ("init_weight", "MODEL.INIT_WEIGHTS"), ("n_ctx", "TRAINER.COCOOP.N_CTX"),
                      ("eval_mode", "EVAL_MODE"), ("eval_tau", "EVAL_TAU"), ("transforms", "INPUT.TRANSFORMS")):   # (--transforms decides "normalize", transforms.py:514-518)
        if g(flag):                                              # 3. reset_cfg: `if args.x:` -- 0 / "" / None leave the cfg alone
            _set(flat, key, g(flag), f"--{flag}")
    merge_from_list(flat, g("opts") or [])                       # 4.
    return to_namespace(flat)


def subsample_classes(*datasets: Sequence, subsample: str = "all") -> List[List]:
    """datasets/oxford_pets.py:141-202 on (path, label) items (+ anything after the label is kept): the sorted label set of the
    FIRST dataset is cut at m = ceil(n / 2) -- `base` keeps the first m labels, `new` the rest -- and the kept labels are renumbered
    from 0 in sorted order, in every dataset.  Returns one filtered, relabelled list per dataset; `all` returns them unchanged."""
    import math
    if subsample not in ("all", "base", "new"):
        raise AssertionError(f"DATASET.SUBSAMPLE_CLASSES must be all, base or new, got {subsample!r}")
    if subsample == "all":
        return [list(d) for d in datasets]
    labels = sorted({it[1] for it in datasets[0]})
    m = math.ceil(len(labels) / 2)
    print(f"SUBSAMPLE {subsample.upper()} CLASSES!")
    selected = labels[:m] if subsample == "base" else labels[m:]
    relabel = {y: y_new for y_new, y in enumerate(selected)}
    return [[(it[0], relabel[it[1]], *it[2:]) for it in d if it[1] in relabel] for d in datasets]
