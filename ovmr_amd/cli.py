"""Runner for the generation / evaluation modes of the reference's train.py.

    python -m ovmr_amd.cli --eval-only --trainer MM_CLS_OP \\
        --root DATA --clip-weights ViT-B-16.pt --bpe-path bpe_simple_vocab_16e6.txt.gz \\
        --model-dir ./checkpoints --load-epoch 30 --output-dir output_ovmr/generated_classifiers \\
        --eval_mode fusion --eval_tau 10 --n_ctx 2  DATASET.NUM_SHOTS 16

keeps the flags of `scripts/mm_cls/generate_classifier.sh:30-44` / `train.py:183-255` that the hot path reads
(`--output-dir --model-dir --load-epoch --eval_mode --eval_tau --n_ctx --seed`, trailing `KEY VALUE` opts
`DATASET.NUM_SHOTS`, `TEST.BATCH_SIZE`); yacs/Dassl are not needed.  Data layout (datasets/imagenet.py:146-159):
`<root>/<split>/<class folder>/<image>`; `<root>/classnames.txt` lines "<folder> <class name>" (optional: folder names
are used otherwise).  The exemplar (eval) set is NUM_SHOTS images per class folder of `--eval-split` (default "train") drawn
under `--seed` exactly as the reference's generate_fewshot_dataset draws them, the test set is every image of `--test-split`
(default "val").

Input pipeline (ovmr_amd/loader.py): `--workers` processes decode + resize + crop to uint8 into a page-locked shared ring, a side
stream uploads a batch and runs ovmr_preprocess_u8 (normalise + fp16 on the GPU) while the encoder works on the previous one.
JPEG decode on the host's cores remains the bound of the whole job (a few thousand images/s against ~29 k for the encoder): the
runner prints end-to-end images/s and the fraction of the time the encoder sat idle.
"""
from __future__ import annotations

import argparse
import collections
import os
import os.path as osp
import sys
from typing import Dict, List, Sequence, Tuple

import numpy as np
import torch

PIXEL_MEAN = (0.48145466, 0.4578275, 0.40821073)      # configs/trainers/MM_CLS_OP/*.yaml:14-15
PIXEL_STD = (0.26862954, 0.26130258, 0.27577711)


def test_transform(img, size: int = 224) -> torch.Tensor:
    """_build_transform_test (Dassl.pytorch/dassl/data/transforms/transforms.py:495-526): bicubic resize of the
    smaller edge to `size`, center crop, [0,1] tensor, normalise."""
    from PIL import Image
    img = img.convert("RGB")
    w, h = img.size
    if w <= h:
        nw, nh = size, max(size, int(size * h / w))
    else:
        nw, nh = max(size, int(size * w / h)), size
    img = img.resize((nw, nh), Image.BICUBIC)
    left, top = int(round((nw - size) / 2.0)), int(round((nh - size) / 2.0))
    img = img.crop((left, top, left + size, top + size))
    x = torch.from_numpy(np.asarray(img, dtype=np.uint8).copy()).permute(2, 0, 1).float().div_(255.0)
    mean = torch.tensor(PIXEL_MEAN).view(3, 1, 1)
    std = torch.tensor(PIXEL_STD).view(3, 1, 1)
    return (x - mean) / std


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
                 num_classes: int = 0):
        self.bs, self.size = batch_size, size
        self.presharded = world > 1
        if world > 1:
            from .shard import shard_range
            lo, hi = shard_range(num_classes or (1 + max(l for _, l in items)), rank, world)
            items = [it for it in items if lo <= it[1] < hi]
        self.items = list(items)

    def __len__(self):
        return (len(self.items) + self.bs - 1) // self.bs

    def __iter__(self):
        from PIL import Image
        for s in range(0, len(self.items), self.bs):
            chunk = self.items[s:s + self.bs]
            imgs = torch.stack([test_transform(Image.open(p), self.size) for p, _ in chunk])
            yield {"img": imgs, "label": torch.tensor([l for _, l in chunk], dtype=torch.long)}


def fewshot_items(items: Sequence[Tuple[str, int]], shots: int, seed: int = 1) -> List[Tuple[str, int]]:
    """The few-shot subset, drawn as the reference draws it: `random.seed(SEED)` (set_random_seed, train.py:183-186), then per
    class, in order of first appearance, `random.sample(items_of_the_class, NUM_SHOTS)`; a class with fewer images keeps them
    all and draws nothing (Dassl.pytorch/dassl/data/datasets/base_dataset.py:175-205 generate_fewshot_dataset, repeat=False).
    Same dataset + same seed = the same images as the reference's run (tests/golden/fewshot.npz holds the reference's picks)."""
    import random
    rng = random.Random(seed)                  # the stream of the global generator after random.seed(seed)
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
    per: Dict[int, List[Tuple[str, int]]] = {}
    for it in fewshot_items(items, shots, seed):
        per.setdefault(it[1], []).append(it)
    fill = np.random.RandomState(seed)
    out = []
    for its in per.values():
        if len(its) < shots:
            its = its + [its[int(k)] for k in fill.choice(len(its), size=shots - len(its), replace=True)]
        out.extend(its)
    return out


def parse(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--root", required=True)
    ap.add_argument("--output-dir", default="output_ovmr/generated_classifiers")
    ap.add_argument("--seed", type=int, default=1)
    ap.add_argument("--trainer", default="MM_CLS_OP")
    ap.add_argument("--config-file", default="")
    ap.add_argument("--dataset-config-file", default="")
    ap.add_argument("--eval-only", action="store_true")
    ap.add_argument("--model-dir", default="")
    ap.add_argument("--load-epoch", type=int, default=None)
    ap.add_argument("--eval_mode", default="fusion", choices=["text", "vision", "multimodal", "fusion"])
    ap.add_argument("--eval_tau", type=float, default=10.0)
    ap.add_argument("--n_ctx", type=int, default=2)
    ap.add_argument("--clip-weights", required=True, help="OpenAI CLIP .pt (TorchScript archive or state dict)")
    ap.add_argument("--bpe-path", default=os.environ.get("OVMR_BPE_PATH"))
    ap.add_argument("--eval-split", default="train")
    ap.add_argument("--test-split", default="val")
    ap.add_argument("--device", default="cuda:0")
    ap.add_argument("--workers", type=int, default=8,
                    help="decode worker processes of the pipelined loader (DATALOADER.NUM_WORKERS of the reference: 8); 0 = decode in this thread")
    ap.add_argument("--prefetch", type=int, default=3, help="batches the workers decode ahead")
    ap.add_argument("--fast-decode", action="store_true", help="JPEG draft mode (DCT-domain downscale): faster, pixels differ slightly from the reference's")
    ap.add_argument("opts", nargs=argparse.REMAINDER, help="KEY VALUE pairs: DATASET.NUM_SHOTS, TEST.BATCH_SIZE")
    return ap.parse_args(argv)


def main(argv=None) -> Dict[str, float]:
    args = parse(argv)
    if args.trainer != "MM_CLS_OP" or not args.eval_only:
        raise SystemExit("only `--eval-only --trainer MM_CLS_OP` (classifier generation / evaluation) is on the hot path")
    opts = dict(zip(args.opts[0::2], args.opts[1::2]))
    shots = int(opts.get("DATASET.NUM_SHOTS", 16))
    batch = int(opts.get("TEST.BATCH_SIZE", 256))
    if osp.isdir(args.output_dir) and osp.exists(osp.join(args.output_dir, "mm_classifiers.pt")):
        print(f"Oops! The results exist at {args.output_dir} (so skip this job)")       # generate_classifier.sh:27-28
        return {}

    from . import checkpoint, modules
    from .evaluator import Classification
    from .tokenizer import BPETokenizer

    torch.manual_seed(args.seed)
    folders, eval_all = list_split(args.root, args.eval_split)
    classnames = read_classnames(args.root, folders)
    clip_model = modules.build_model(checkpoint.load_clip_state_dict(args.clip_weights), device=args.device)
    size = clip_model.spec.image_resolution
    cfg = modules.make_cfg(n_ctx=args.n_ctx, num_shots=shots, eval_mode=args.eval_mode, eval_tau=args.eval_tau,
                           output_dir=args.output_dir, test_batch_size=batch, size=224)
    cfg.SEED = args.seed
    pl_state = checkpoint.load_prompt_learner_state(args.model_dir, args.load_epoch) if args.model_dir else None
    if pl_state is None:
        print("Note that load_model() is skipped as no pretrained model is given")       # :464-466
    model = modules.CustomCLIP(cfg, classnames, clip_model, tokenizer=BPETokenizer(args.bpe_path),
                               prompt_learner_state=pl_state, reserve=(batch, 256, max(1024, len(classnames))))
    import torch.distributed as dist
    rank, world = (dist.get_rank(), dist.get_world_size()) if dist.is_available() and dist.is_initialized() else (0, 1)
    exemplars = exemplar_items(eval_all, shots, args.seed)
    _, test_items = list_split(args.root, args.test_split)
    if args.workers > 0:
        from .loader import PipelinedFolderLoader
        kw = dict(workers=args.workers, prefetch=args.prefetch, device=args.device, fast_decode=args.fast_decode)
        eval_loader = PipelinedFolderLoader(exemplars, batch // shots * shots, size, rank, world, len(classnames), **kw)
        test_loader = PipelinedFolderLoader(test_items, batch, size, **kw)
    else:
        eval_loader = FolderLoader(exemplars, batch // shots * shots, size, rank, world, len(classnames))
        test_loader = FolderLoader(test_items, batch, size)
    evaluator = Classification(len(classnames), classnames, device=args.device)
    model.forward_prompt(eval_loader)        # the reference does this inside the first forward (:341-342); up front it keeps the two loaders' statistics apart
    labels = collections.deque()

    def test_images():
        for b in test_loader:
            labels.append(b["label"])
            yield b["img"]

    for out in model.forward_batches(test_images(), eval_set_loader=eval_loader):      # two test batches in flight (modules.py)
        evaluator.process(out, labels.popleft())
    results = dict(evaluator.evaluate(args.output_dir))
    for name, ld in (("exemplar set", eval_loader), ("test set", test_loader)):
        st = getattr(ld, "stats", None)
        if st:
            print(f"input pipeline, {name}: {st['images']} images in {st['wall_s']:.2f} s = {st['images_per_s']:.0f} img/s end to end "
                  f"({st['workers']} decode workers), encoder idle {100 * st['encoder_idle_fraction']:.0f} % of the time, "
                  f"host blocked on decode {st['decode_wait_s']:.2f} s")
            results[f"pipeline_{name.split()[0]}"] = st
    return results


if __name__ == "__main__":
    main()
