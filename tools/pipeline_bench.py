#!/usr/bin/env python3
"""End-to-end run of the runner's input pipeline on a generated folder of JPEGs: ovmr_amd.cli with ViT-B/16
(synthetic weights), classifier generation + evaluation, reporting images/s end to end and the fraction of the time the encoder
sat idle waiting for the host's JPEG decode -- for the pipelined loader (N worker processes) and, with --compare-plain, for the
single-threaded loader it replaces.

    python tools/pipeline_bench.py [--classes 64] [--shots 16] [--val-per-class 48] [--workers 16]
"""
import argparse
import json
import os
import shutil
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def _write_class(job):
    """One class folder of one split (a worker of make_dataset; numpy + PIL only)."""
    import numpy as np
    from PIL import Image
    d, c, per, seed = job
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:375, 0:500].astype(np.float32)
    os.makedirs(d, exist_ok=True)
    for i in range(per):
        f = rng.uniform(0.004, 0.03, (3, 2))
        ph = rng.uniform(0, 6.28, (3, 2))
        img = np.stack([127 + 70 * np.sin(f[k, 0] * xx + ph[k, 0]) * np.cos(f[k, 1] * yy + ph[k, 1]) for k in range(3)], -1)
        cx, cy = rng.uniform(100, 400), rng.uniform(80, 300)
        blob = np.exp(-((xx - cx) ** 2 + (yy - cy) ** 2) / (2 * 60.0 ** 2))[..., None]
        col = np.array([(37 * c) % 255, (91 * c) % 255, (53 * c) % 255], dtype=np.float32)
        img = img * (1 - blob) + col * blob + rng.normal(0, 6, img.shape)
        Image.fromarray(img.clip(0, 255).astype(np.uint8)).save(os.path.join(d, f"{i:04d}.JPEG"), quality=88)
    return per


def make_dataset(root, classes, train_per_class, val_per_class, seed=0):
    """Smooth synthetic photographs (low-frequency fields + a class-coloured blob), 500 x 375 JPEGs like ImageNet's typical size;
    written by a pool of processes (started before this process touches torch or the GPU)."""
    from concurrent.futures import ProcessPoolExecutor
    jobs = [(os.path.join(root, split, f"n{c:04d}"), c, per, seed * 100003 + si * 10007 + c)
            for si, (split, per) in enumerate((("train", train_per_class), ("val", val_per_class))) for c in range(classes)]
    with ProcessPoolExecutor(max_workers=min(16, len(os.sched_getaffinity(0)))) as ex:
        n = sum(ex.map(_write_class, jobs))
    with open(os.path.join(root, "classnames.txt"), "w") as f:
        f.write("".join(f"n{c:04d} thing number {c}\n" for c in range(classes)))
    return n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--classes", type=int, default=64)
    ap.add_argument("--shots", type=int, default=16)
    ap.add_argument("--val-per-class", type=int, default=48)
    ap.add_argument("--workers", type=int, nargs="+", default=[16], help="decode worker counts to run, one CLI run each")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--dir", default="/tmp/ovmr_pipeline_bench")
    ap.add_argument("--compare-plain", action="store_true", help="also run the single-threaded loader (--workers 0)")
    ap.add_argument("--fast-decode", action="store_true")
    ap.add_argument("--host-resize", action="store_true", help="also run every worker count with Resize + CenterCrop in the workers (the round-3/4 pipeline)")
    args = ap.parse_args()
    import torch
    from ovmr_amd import checkpoint, cli, synth
    from test_next_rows_cpu import make_synthetic_bpe
    shutil.rmtree(args.dir, ignore_errors=True)
    data = os.path.join(args.dir, "data")
    t0 = time.perf_counter()
    n = make_dataset(data, args.classes, args.shots + 4, args.val_per_class)
    print(f"wrote {n} JPEGs in {time.perf_counter() - t0:.1f} s", flush=True)
    spec = synth.SPECS["ViT-B/16"]
    sd = {k: torch.from_numpy(v) for k, v in synth.clip_state_dict(spec, 11, jitter=True).items()}
    torch.save(sd, os.path.join(args.dir, "clip.pt"))
    pl = {k: torch.from_numpy(v) for k, v in synth.prompt_learner_state_dict(spec, 2, 11, True).items()}
    checkpoint.save_prompt_learner_state(pl, os.path.join(args.dir, "ckpt"), 30)
    bpe = os.path.join(args.dir, "bpe.txt.gz")
    make_synthetic_bpe(bpe)
    out = {}
    runs = [(w, False) for w in args.workers] + ([(w, True) for w in args.workers] if args.host_resize else []) + ([(0, False)] if args.compare_plain else [])
    for workers, host in runs:
        odir = os.path.join(args.dir, f"out_w{workers}{'_host' if host else ''}")
        t0 = time.perf_counter()
        res = cli.main(["--root", data, "--trainer", "MM_CLS_OP", "--n_ctx", "2", "--eval_mode", "fusion", "--eval_tau", "10", "--clip-weights", os.path.join(args.dir, "clip.pt"), "--bpe-path", bpe, "--eval-only",
                        "--model-dir", os.path.join(args.dir, "ckpt"), "--load-epoch", "30", "--output-dir", odir,
                        "--workers", str(workers)] + (["--fast-decode"] if args.fast_decode else []) + (["--host-resize"] if host else []) +
                       ["DATASET.NUM_SHOTS", str(args.shots), "DATALOADER.TEST.BATCH_SIZE", str(args.batch)])
        wall = time.perf_counter() - t0
        images = args.classes * (args.shots + args.val_per_class)
        tag = f"workers_{workers}" + ("_host_resize" if host else "")
        out[tag] = {"images": images, "resize": "workers (PIL)" if host else "GPU (ovmr_resize_crop_u8)", "wall_s_incl_model_setup": round(wall, 2),
                                    "pipeline_exemplar": res.get("pipeline_exemplar"), "pipeline_test": res.get("pipeline_test"),
                                    "accuracy": res.get("accuracy")}
        print(json.dumps(out[tag]), flush=True)
    if args.compare_plain:
        a = torch.load(os.path.join(args.dir, f"out_w{args.workers[0]}", "mm_classifiers.pt"), map_location="cpu")
        b = torch.load(os.path.join(args.dir, "out_w0", "mm_classifiers.pt"), map_location="cpu")
        cos = {k: float(torch.nn.functional.cosine_similarity(a[k], b[k], dim=-1).min()) for k in a if k != "fusion_weight"}
        out["min_row_cosine_pipelined_vs_plain"] = cos
        out["rows_bit_equal"] = {k: bool(torch.equal(a[k], b[k])) for k in a}
        print(json.dumps({"min_row_cosine_pipelined_vs_plain": cos, "rows_bit_equal": out["rows_bit_equal"]}))
    print(json.dumps({"cpus": os.cpu_count(), "usable_cpus": len(os.sched_getaffinity(0))}))


if __name__ == "__main__":
    main()
