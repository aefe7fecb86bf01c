"""Runner for the generation / evaluation modes of the reference's train.py.

    python -m ovmr_amd.cli --eval-only --trainer MM_CLS_OP \\
        --root DATA --clip-weights ViT-B-16.pt --bpe-path bpe_simple_vocab_16e6.txt.gz \\
        --model-dir ./checkpoints --load-epoch 30 --output-dir output_ovmr/generated_classifiers \\
        --eval_mode fusion --eval_tau 10 --n_ctx 2  DATASET.NUM_SHOTS 16

takes the command line of `scripts/mm_cls/generate_classifier.sh:30-44` / `train.py:183-255` as it is: `--dataset-config-file` and
`--config-file` (YAML, read with PyYAML), the flags `reset_cfg` copies (`--root --output-dir --seed --trainer --backbone --init_weight
--n_ctx --eval_mode --eval_tau`) and the trailing `KEY VALUE` opts, merged in train.py's order by `ovmr_amd.config.setup_cfg` with the
reference's defaults (EVAL_MODE multimodal, N_CTX 16, DATALOADER.TEST.BATCH_SIZE 32, bilinear resize and NO normalisation unless the
config file says otherwise -- exactly what train.py does without its YAML files).  Keys this path honours: `DATASET.NUM_SHOTS`,
`DATASET.SUBSAMPLE_CLASSES` (all | base | new, datasets/oxford_pets.py:141-202), `DATALOADER.TEST.BATCH_SIZE`, `DATALOADER.NUM_WORKERS`,
`INPUT.SIZE / INTERPOLATION / PIXEL_MEAN / PIXEL_STD / TRANSFORMS`, `TRAINER.COCOOP.N_CTX`, `MODEL.BACKBONE.NAME`, `MODEL.INIT_WEIGHTS`,
`EVAL_MODE`, `EVAL_TAU`, `SEED`, `OUTPUT_DIR`; training / optimiser keys are accepted and ignored; any other key raises.  yacs / Dassl
are not needed; `--transforms` sets INPUT.TRANSFORMS as in train.py:69-70, train.py's remaining flags (`--source-domains --target-domains
--fs_classifier --head --stage_num --visual_token_path`) are accepted and ignored.  Extra flags of this runner: `--clip-weights` (no download here), `--bpe-path`, `--eval-split / --test-split`, the
input-pipeline knobs, `--exemplar-list`.  Data layout (datasets/imagenet.py:146-159):
`<root>/<split>/<class folder>/<image>`; `<root>/classnames.txt` lines "<folder> <class name>" (optional: folder names
are used otherwise).  The exemplar (eval) set is NUM_SHOTS images per class folder of `--eval-split` (default "train") drawn
under `--seed` exactly as the reference's generate_fewshot_dataset draws them, the test set is every image of `--test-split`
(default "val").

Input pipeline (ovmr_amd/loader.py): `--workers` processes DECODE into a page-locked shared ring, a side stream uploads a batch and runs
ovmr_resize_crop_u8 (Resize + CenterCrop, bit-equal to PIL) and ovmr_preprocess_u8 (normalise + fp16) on the GPU while the encoder
works on the previous one; `--host-resize` keeps the resize in the workers.  JPEG decode on the host's cores remains the bound of the
whole job (profiles/r05e_pipeline_bench_device_vs_host_resize.log: 14.5 k images/s with 16 workers against ~30 k for the encoder): the
runner prints end-to-end images/s and the share of the time the host waited for the decoders.
"""
from __future__ import annotations

import argparse
import collections
import os
import os.path as osp
import sys
from typing import Dict, List, Sequence, Tuple

import numpy as np            # (torch is imported inside the functions: under `python -m ovmr_amd.cli` the spawned decode workers
                              #  re-import this module as __mp_main__ and must stay torch-free)

PIXEL_MEAN = (0.48145466, 0.4578275, 0.40821073)      # configs/trainers/MM_CLS_OP/*.yaml:14-15
PIXEL_STD = (0.26862954, 0.26130258, 0.27577711)


def test_transform(img, size: int = 224, interpolation: str = "bicubic", mean=PIXEL_MEAN, std=PIXEL_STD) -> torch.Tensor:
    """_build_transform_test (Dassl.pytorch/dassl/data/transforms/transforms.py:495-526): resize of the smaller edge to `size`
    (INPUT.INTERPOLATION; the MM_CLS_OP configs say bicubic), center crop, [0,1] tensor, normalise with INPUT.PIXEL_MEAN / PIXEL_STD
    (`mean=None`: no Normalize, what the reference does when "normalize" is not in INPUT.TRANSFORMS, :514-518)."""
    import torch
    from ._decode_worker import load_u8
    x = torch.from_numpy(load_u8(img, size, interpolation=interpolation).copy()).permute(2, 0, 1).float().div_(255.0)
    if mean is None:
        return x
    return (x - torch.tensor(tuple(mean)).view(3, 1, 1)) / torch.tensor(tuple(std)).view(3, 1, 1)


def list_split(root: str, split: str) -> Tuple[List[str], List[Tuple[str, int]]]:
    """datasets/imagenet.py:146-159: sorted class folders -> labels 0..C-1, every non-hidden file is an image."""
    d = osp.join(root, split)
    folders = sorted(f.name for f in os.scandir(d) if f.is_dir())
    items = []
    for label, folder in enumerate(folders):
        for name in sorted(n for n in os.listdir(osp.join(d, folder)) if not n.startswith(".")):
            items.append((osp.join(d, folder, name), label))
    return folders, items


def read_classnames(root: str, folders: Sequence[str]) -> List[str]:
    path = osp.join(root, "classnames.txt")
    names: Dict[str, str] = {}
    if osp.exists(path):
        with open(path) as f:
            for line in f:
                parts = line.strip().split(" ")
                if parts and parts[0]:
                    names[parts[0]] = " ".join(parts[1:])
    return [names.get(f, f) for f in folders]


class FolderLoader:
    """Iterable of {"img", "label"} dict batches (the loader protocol of SURVEY.md 8b).

    With world > 1 the loader is class-sharded: a rank keeps only the items whose class lies in its contiguous
    `shard_range` of the `num_classes` classes, so it opens and decodes only its own images; `presharded = True` then tells
    CustomCLIP.forward_prompt that every batch it receives is its own."""

    def __init__(self, items: Sequence[Tuple[str, int]], batch_size: int, size: int, rank: int = 0, world: int = 1,
                 num_classes: int = 0, interpolation: str = "bicubic", mean=PIXEL_MEAN, std=PIXEL_STD):
        self.bs, self.size = batch_size, size
        self.tfm = dict(interpolation=interpolation, mean=mean, std=std)
        self.presharded = world > 1
        if world > 1:
            from .shard import shard_range
            lo, hi = shard_range(num_classes or (1 + max(l for _, l in items)), rank, world)
            items = [it for it in items if lo <= it[1] < hi]
        self.items = list(items)

    def __len__(self):
        return (len(self.items) + self.bs - 1) // self.bs

    def __iter__(self):
        import torch
        for s in range(0, len(self.items), self.bs):
            chunk = self.items[s:s + self.bs]
            imgs = torch.stack([test_transform(p, self.size, **self.tfm) for p, _ in chunk])
            yield {"img": imgs, "label": torch.tensor([l for _, l in chunk], dtype=torch.long)}


def fewshot_items(items: Sequence[Tuple[str, int]], shots: int, seed: int = 1) -> List[Tuple[str, int]]:
    """The few-shot subset, drawn as the reference draws it: `random.seed(SEED)` (set_random_seed, train.py:183-186), then per
    class, in order of first appearance, `random.sample(items_of_the_class, NUM_SHOTS)`; a class with fewer images keeps them
    all and draws nothing (Dassl.pytorch/dassl/data/datasets/base_dataset.py:175-205 generate_fewshot_dataset, repeat=False).
    The sampling PROCEDURE is the reference's (tests/golden/fewshot.npz holds the reference's picks for the same item lists); the
    same images as a reference run additionally need the same item ORDER inside a class: the reference lists a class folder in
    raw os.listdir order (listdir_nohidden(sort=False), datasets/imagenet.py:149), this runner sorts the names.  To reproduce a
    particular reference run, hand its exemplar set over with `--exemplar-list`.  seed < 0: unseeded, as train.py:157-159."""
    import random
    rng = random.Random(seed) if seed >= 0 else random.Random()    # the stream of the global generator after random.seed(seed)
    per: Dict[int, List[Tuple[str, int]]] = {}
    for it in items:
        per.setdefault(it[1], []).append(it)
    out = []
    for its in per.values():
        out.extend(rng.sample(its, shots) if len(its) >= shots else its)
    return out


def exemplar_items(items: Sequence[Tuple[str, int]], shots: int, seed: int = 1) -> List[Tuple[str, int]]:
    """`fewshot_items` laid out for the eval-set loader: exactly NUM_SHOTS consecutive rows per class.  A class
    with fewer images is filled up with replacement, as RandomClassSampler does (samplers.py:148-149, there from numpy's global
    generator at iteration time; here from RandomState(seed)).  The reference additionally shuffles the order of the shots and
    of the classes (samplers.py:117-181), which the path does not depend on: the aggregator has no positional embedding."""
    return layout_exemplars(fewshot_items(items, shots, seed), shots, seed)


def layout_exemplars(few: Sequence[Tuple[str, int]], shots: int, seed: int = 1) -> List[Tuple[str, int]]:
    """The eval-set loader's row order for an already drawn few-shot subset (see `exemplar_items`)."""
    per: Dict[int, List[Tuple[str, int]]] = {}
    for it in few:
        per.setdefault(it[1], []).append(it)
    fill = np.random.RandomState(seed if seed >= 0 else None)
    out = []
    for label, its in per.items():
        if len(its) > shots:
            # forward_prompt groups rows purely by position (num_cls = B // S, label.reshape(num_cls, S)[:, 0],
            # trainers/mm_classifier_one_prompt.py:237-240): one surplus row would shift every later class group.  The reference's
            # RandomClassSampler emits exactly n_ins rows per class (samplers.py:117-181); a list with more is refused, not trimmed
            raise ValueError(f"class {label} has {len(its)} exemplar rows, more than DATASET.NUM_SHOTS = {shots}")
        if len(its) < shots:
            its = its + [its[int(k)] for k in fill.choice(len(its), size=shots - len(its), replace=True)]
        out.extend(its)
    return out


def read_exemplar_list(path: str, num_classes: int, shots: int) -> List[Tuple[str, int]]:
    """`--exemplar-list`: one `<image path> <label>` per line.  Checked before anything is decoded: every label inside
    [0, num_classes), every file present, at most NUM_SHOTS rows per class (fewer are filled up with replacement by
    `layout_exemplars`, as RandomClassSampler does) -- a malformed list fails here with the offending line, not as shifted class
    groups inside forward_prompt."""
    few: List[Tuple[str, int]] = []
    per: Dict[int, int] = {}
    with open(path) as f:
        for no, raw in enumerate(f, 1):
            ln = raw.strip()
            if not ln:
                continue
            parts = ln.rsplit(" ", 1)
            if len(parts) != 2 or not parts[1].lstrip("-").isdigit():
                raise SystemExit(f"{path}:{no}: expected `<image path> <label>`, got {ln!r}")
            img, label = parts[0], int(parts[1])
            if not 0 <= label < num_classes:
                raise SystemExit(f"{path}:{no}: label {label} outside [0, {num_classes}) -- the labels are the class-folder indices BEFORE class subsampling")
            if not os.path.isfile(img):
                raise SystemExit(f"{path}:{no}: image {img!r} does not exist")
            per[label] = per.get(label, 0) + 1
            if per[label] > shots:
                raise SystemExit(f"{path}:{no}: class {label} has more than DATASET.NUM_SHOTS = {shots} exemplars (the eval-set loader "
                                 "carries exactly NUM_SHOTS rows per class; trim the list or raise NUM_SHOTS)")
            few.append((img, label))
    if not few:
        raise SystemExit(f"{path}: no exemplars listed")
    return few


def parse(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    # the reference's flags, with the reference's defaults (train.py:183-255)
    ap.add_argument("--root", type=str, default="", help="path to dataset")
    ap.add_argument("--output-dir", type=str, default="", help="output directory")
    ap.add_argument("--resume", type=str, default="")
    ap.add_argument("--seed", type=int, default=-1, help="only positive value enables a fixed seed")
    ap.add_argument("--config-file", type=str, default="", help="path to config file")
    ap.add_argument("--dataset-config-file", type=str, default="", help="path to config file for dataset setup")
    ap.add_argument("--init_weight", type=str, default="")
    ap.add_argument("--trainer", type=str, default="", help="name of trainer")
    ap.add_argument("--backbone", type=str, default="", help="name of CNN backbone")
    ap.add_argument("--eval-only", action="store_true", help="evaluation only")
    ap.add_argument("--model-dir", type=str, default="", help="load model from this directory for eval-only mode.  A checkpoint written by the "
                    "reference's save_checkpoint also pickles scheduler / optimiser objects: it is read with a restricted unpickler (tensors load, the "
                    "objects become inert placeholders, no code from the file runs); OVMR_TRUSTED_CHECKPOINTS=1 asks for torch's full unpickling instead")
    ap.add_argument("--load-epoch", type=int, help="load model weights at this epoch for evaluation")
    ap.add_argument("--eval_tau", type=float)
    ap.add_argument("--eval_mode", type=str, default="multimodal")
    ap.add_argument("--n_ctx", type=int, help="number of ctx")
    ap.add_argument("--no-train", action="store_true")
    # flags of train.py the generation / evaluation modes never read (DA / DG domains, training augmentations, the stage-1 classifier,
    # head and stage number): accepted so that a reference command line runs unchanged, and ignored
    ap.add_argument("--source-domains", type=str, nargs="+", help="(accepted, ignored)")
    ap.add_argument("--target-domains", type=str, nargs="+", help="(accepted, ignored)")
    ap.add_argument("--transforms", type=str, nargs="+", help="INPUT.TRANSFORMS (train.py:69-70): of the test transform only `normalize` depends on it")
    ap.add_argument("--fs_classifier", type=str, default="", help="(accepted, ignored)")
    ap.add_argument("--head", type=str, default="", help="(accepted, ignored)")
    ap.add_argument("--stage_num", type=int, help="(accepted, ignored)")
    ap.add_argument("--visual_token_path", type=str, default="visual token path", help="(accepted, ignored)")
    # this runner's own
    ap.add_argument("--clip-weights", required=True, help="OpenAI CLIP .pt (TorchScript archive or state dict); the reference downloads it by MODEL.BACKBONE.NAME")
    ap.add_argument("--bpe-path", default=os.environ.get("OVMR_BPE_PATH"))
    ap.add_argument("--eval-split", default="train")
    ap.add_argument("--test-split", default="val")
    ap.add_argument("--exemplar-list", default="", help="text file, one `<image path> <label>` per line: use exactly these exemplars (e.g. a "
                    "reference run's few-shot set) instead of drawing NUM_SHOTS per class under --seed; labels are those BEFORE class subsampling")
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--workers", type=int, default=None,
                    help="decode worker processes of the pipelined loader (default: DATALOADER.NUM_WORKERS); 0 = decode in this thread")
    ap.add_argument("--prefetch", type=int, default=3, help="batches the workers decode ahead")
    ap.add_argument("--host-resize", action="store_true", help="Resize + CenterCrop in the decode workers (PIL) instead of on the GPU (ovmr_resize_crop_u8; same bytes either way)")
    ap.add_argument("--fast-decode", action="store_true", help="JPEG draft mode (DCT-domain downscale): faster, pixels differ slightly from the reference's")
    ap.add_argument("opts", default=None, nargs=argparse.REMAINDER, help="modify config options using the command-line (KEY VALUE pairs)")
    return ap.parse_args(argv)


def build_splits(cfg, eval_split: str = "train", test_split: str = "val", exemplar_list: str = ""):
    """What the reference's dataset class hands the DataManager (datasets/imagenet.py:16-64), for a folder dataset: the few-shot
    draw of `eval_split` under cfg.SEED (or the images listed in `exemplar_list`), THEN the class subsampling of
    DATASET.SUBSAMPLE_CLASSES applied to the exemplar and the test items alike (:60-62), the class names of the surviving labels in
    label order.  Returns (classnames, exemplar items laid out NUM_SHOTS rows per class, test items)."""
    from . import config
    shots, seed, sub = cfg.DATASET.NUM_SHOTS, cfg.SEED, cfg.DATASET.SUBSAMPLE_CLASSES
    root = cfg.DATASET.ROOT
    folders, eval_all = list_split(root, eval_split)
    all_names = read_classnames(root, folders)
    _, test_items = list_split(root, test_split)
    if exemplar_list:
        few = read_exemplar_list(exemplar_list, len(folders), shots)
    else:
        few = fewshot_items(eval_all, shots, seed)
    all_labels = sorted({l for _, l in few})                  # the label set `subsample_classes` splits (train_x, oxford_pets.py:160-168)
    few, test_items = config.subsample_classes(few, test_items, subsample=sub)
    half = -(-len(all_labels) // 2)
    kept = {"all": all_labels, "base": all_labels[:half], "new": all_labels[half:]}[sub]
    if sub == "all" and kept != list(range(len(folders))):
        raise SystemExit(f"{len(folders) - len(kept)} class folder(s) of {eval_split!r} hold no exemplar image")
    classnames = [all_names[y] for y in kept]                 # lab2cname of the relabelled train_x (base_dataset.py get_lab2cname)
    return classnames, layout_exemplars(few, shots, seed), test_items


BACKBONES = {"ViT-B/16": (768, 16, 12), "ViT-B/32": (768, 32, 12), "ViT-L/14": (1024, 14, 24), "ViT-L/14@336px": (1024, 14, 24)}   # clip/clip.py:32-40 (ViT entries)


def main(argv=None) -> Dict[str, float]:
    from . import config
    args = parse(argv)
    cfg = config.setup_cfg(args)                              # train.py:134-155
    if cfg.TRAINER.NAME != "MM_CLS_OP" or not args.eval_only:
        raise SystemExit("only `--eval-only --trainer MM_CLS_OP` (classifier generation / evaluation) is on the hot path")
    out_dir = cfg.OUTPUT_DIR
    if osp.isdir(out_dir) and osp.exists(osp.join(out_dir, "mm_classifiers.pt")):
        print(f"Oops! The results exist at {out_dir} (so skip this job)")       # generate_classifier.sh:27-28
        return {}
    shots, batch, seed = cfg.DATASET.NUM_SHOTS, cfg.DATALOADER.TEST.BATCH_SIZE, cfg.SEED
    if shots < 1:
        raise SystemExit("DATASET.NUM_SHOTS must be >= 1: the eval-set loader draws NUM_SHOTS rows per class (data_manager.py:157-170)")
    if batch < shots:
        raise SystemExit(f"DATALOADER.TEST.BATCH_SIZE {batch} is smaller than DATASET.NUM_SHOTS {shots}: RandomClassSampler needs one class per batch")
    if cfg.DATALOADER.K_TRANSFORMS != 1:
        raise SystemExit("DATALOADER.K_TRANSFORMS > 1 only applies to training transforms; the test transform has one view")

    import torch
    from . import checkpoint, modules
    from .evaluator import Classification
    from .tokenizer import BPETokenizer

    # launched by torch.distributed.run (scripts/generate_classifier.sh with several GPUs): one rank per GPU, process group nccl = RCCL;
    # forward_prompt then shards the classes over the ranks (two collectives) and rank 0 evaluates and writes
    import torch.distributed as dist
    own_group = False
    if int(os.environ.get("WORLD_SIZE", "1")) > 1 and not dist.is_initialized():
        local = int(os.environ.get("LOCAL_RANK", "0"))
        backend = os.environ.get("OVMR_DIST_BACKEND", "nccl")
        if backend == "nccl":
            args.device = f"cuda:{local}"
        torch.cuda.set_device(torch.device(args.device))
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # ranks > 0 wait in the closing barrier while rank 0 evaluates the whole test set alone: the collective timeout has to cover a test
        # pass, not a collective (the backend's default of 10 minutes would abort the waiting ranks, and the launcher then rank 0)
        import datetime
        to = datetime.timedelta(seconds=float(os.environ.get("OVMR_DIST_TIMEOUT", 6 * 3600)))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device(args.device), timeout=to)
        else:
            dist.init_process_group(backend, timeout=to)
        own_group = True
    if seed >= 0:
        print(f"Setting fixed seed: {seed}")                  # train.py:157-159
        torch.manual_seed(seed)
    classnames, exemplars, test_items = build_splits(cfg, args.eval_split, args.test_split, args.exemplar_list)

    clip_model = modules.build_model(checkpoint.load_clip_state_dict(args.clip_weights), device=args.device)
    spec = clip_model.spec
    size = spec.image_resolution
    name = cfg.MODEL.BACKBONE.NAME
    if name:
        if name not in BACKBONES:
            raise SystemExit(f"MODEL.BACKBONE.NAME {name!r}: only the ViT CLIP models are on the hot path ({sorted(BACKBONES)})")
        if (spec.vision_width, spec.vision_patch_size, spec.vision_layers) != BACKBONES[name]:
            raise SystemExit(f"--clip-weights holds a ViT of width {spec.vision_width}, patch {spec.vision_patch_size}, {spec.vision_layers} layers: "
                             f"not MODEL.BACKBONE.NAME {name!r}")
    if tuple(cfg.INPUT.SIZE) != (size, size):
        raise SystemExit(f"INPUT.SIZE {tuple(cfg.INPUT.SIZE)} does not match the model's input resolution {size} "
                         "(the positional embedding has one row per patch of that resolution, clip/model.py:416)")
    normalize = "normalize" in tuple(cfg.INPUT.TRANSFORMS)    # transforms.py:514-518
    tfm = dict(interpolation=cfg.INPUT.INTERPOLATION, mean=tuple(cfg.INPUT.PIXEL_MEAN) if normalize else None,
               std=tuple(cfg.INPUT.PIXEL_STD) if normalize else None)
    pl_state = None
    if cfg.MODEL.INIT_WEIGHTS:                                # load_pretrained_weights (trainers/mm_classifier_one_prompt.py:403-404)
        ck = checkpoint._torch_load(cfg.MODEL.INIT_WEIGHTS)
        pl_state = ck["state_dict"] if "state_dict" in ck else ck
    if args.model_dir:
        pl_state = checkpoint.load_prompt_learner_state(args.model_dir, args.load_epoch)
    else:
        print("Note that load_model() is skipped as no pretrained model is given")       # :464-466
    import torch.distributed as dist
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)
    workers = cfg.DATALOADER.NUM_WORKERS if args.workers is None else args.workers
    if workers > 0:
        from .loader import PipelinedFolderLoader
        kw = dict(workers=workers, prefetch=args.prefetch, device=args.device, fast_decode=args.fast_decode, device_resize=not args.host_resize, **tfm)
        eval_loader = PipelinedFolderLoader(exemplars, batch // shots * shots, size, rank, world, len(classnames), **kw)
        test_loader = PipelinedFolderLoader(test_items, batch, size, **kw)
        # the decode workers start while the engine takes the weights; on rank 0 one ring (sized for the larger batch) serves both loaders,
        # ranks > 0 never touch the test set: theirs is sized for the exemplar batches alone
        (eval_loader if rank > 0 or eval_loader.bs > test_loader.bs else test_loader).warm()
    else:
        eval_loader = FolderLoader(exemplars, batch // shots * shots, size, rank, world, len(classnames), **tfm)
        test_loader = FolderLoader(test_items, batch, size, **tfm)
    model = modules.CustomCLIP(cfg, classnames, clip_model, tokenizer=BPETokenizer(args.bpe_path),
                               prompt_learner_state=pl_state, reserve=(batch, 256, max(1024, len(classnames))))
    evaluator = Classification(len(classnames), classnames, device=args.device)
    # the reference does this inside the first forward (:341-342); up front it keeps the two loaders' statistics apart.  Rank 0's two
    # files are written by a worker thread while the test set runs (CustomCLIP._write_files), joined below
    model.forward_prompt(eval_loader, wait_files=False)
    if rank > 0:                             # the classifiers are complete on every rank; rank 0 evaluates the test set and reports
        if own_group:
            dist.barrier()
            dist.destroy_process_group()
        return {}
    labels = collections.deque()

    def test_images():
        for b in test_loader:
            labels.append(b["label"])
            yield b["img"]

    for out in model.forward_batches(test_images(), eval_set_loader=eval_loader):      # two test batches in flight (modules.py)
        evaluator.process(out, labels.popleft())
    model.wait_files()
    results = dict(evaluator.evaluate(out_dir))
    for name, ld in (("exemplar set", eval_loader), ("test set", test_loader)):
        st = getattr(ld, "stats", None)
        if st:
            print(f"input pipeline, {name}: {st['images']} images in {st['wall_s']:.2f} s = {st['images_per_s']:.0f} img/s end to end "
                  f"({st['workers']} decode workers, {st.get('device_resized', 0)} images resized on the GPU / {st.get('host_resized', 0)} by the workers), "
                  f"host blocked on decode {st['decode_wait_s']:.2f} s = {100 * st['decode_bound_fraction']:.0f} % of the time")
            results[f"pipeline_{name.split()[0]}"] = st
    results["classnames"] = classnames
    if own_group:
        dist.barrier()
        dist.destroy_process_group()
    return results


if __name__ == "__main__":
    main()
