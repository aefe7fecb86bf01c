#!/usr/bin/env python3
"""ovmr_resize_crop_u8 against PIL on many random input sizes (default 2000; bicubic and bilinear, R = 224 and 336): every byte equal.
The GPU test (tests/test_hip_loader.py) runs 350 sizes; this is the wider sweep behind the claim in DESIGN.md section 8."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from PIL import Image
from ovmr_amd import _decode_worker, loader

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=2000)
ap.add_argument("--max-edge", type=int, default=2600)
a = ap.parse_args()
rng = np.random.default_rng(2025)
bad, done, t0 = [], 0, time.time()
for R, interp, share in ((224, "bicubic", 0.6), (336, "bicubic", 0.15), (224, "bilinear", 0.25)):
    n = int(a.n * share)
    for lo in range(0, n, 32):
        frames = []
        for _ in range(min(32, n - lo)):
            if rng.random() < 0.5:                                   # photograph-like aspect ratios around common sizes
                w = int(rng.integers(120, 1400)); h = max(3, int(w * rng.uniform(0.45, 2.2)))
            else:
                w, h = int(rng.integers(3, a.max_edge)), int(rng.integers(3, a.max_edge))
            if w * h > 3_000_000:
                h = max(3, 3_000_000 // w)
            frames.append(rng.integers(0, 256, (h, w, 3), dtype=np.uint8))
        want = [_decode_worker.load_u8(Image.fromarray(f), R, False, interp) for f in frames]
        got = loader.resize_crop_u8(frames, R, interp).cpu().numpy()
        for f, g_, w_ in zip(frames, got, want):
            done += 1
            if not np.array_equal(g_, w_):
                bad.append((f.shape[1], f.shape[0], R, interp, int((g_ != w_).sum())))
print(json.dumps({"sizes": done, "mismatching": len(bad), "first": bad[:5], "seconds": round(time.time() - t0, 1)}))
sys.exit(1 if bad else 0)
