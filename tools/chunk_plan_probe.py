#!/usr/bin/env python3
"""Encoder launch-sequence plans for a batch that is not a whole number of chunks: ovmr_encode_image on B images with the engine's plan
(chunks of 775, remainder last) against pinned chunk sizes (option enc_chunk), same box, alternating, median of --reps calls."""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from ovmr_amd import modules, synth

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=5)
args = ap.parse_args()
dev = torch.device("cuda:0")
spec = synth.SPECS["ViT-B/16"]
gen = torch.Generator(device=dev).manual_seed(1)
cm = modules.CLIPModel(bench.device_clip_state(spec, gen, dev), spec, str(dev))
e = cm.engine(2)
e.load_state_dict({}, bench.device_pl_state(spec, 2, gen, dev))
e._pl_loaded = True
e.finalize(775, 64, 1024)
for B, chunks in ((2000, (0, 667, 700, 640, 500)), (1275, (0, 638, 664)), (800, (0, 400))):
    img = torch.randn((B, 3, 224, 224), generator=gen, device=dev).half()
    out = torch.empty((B, spec.embed_dim), dtype=torch.float16, device=dev)
    res = {c: [] for c in chunks}
    for rep in range(args.reps + 1):
        for c in chunks:
            e.set_option("enc_chunk", c)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            e.encode_image(img, out=out)
            torch.cuda.synchronize()
            if rep:
                res[c].append((time.perf_counter() - t0) * 1e3)
    e.set_option("enc_chunk", 0)
    line = {}
    for c in chunks:
        e.set_option("enc_chunk", c)
        line[f"chunk {c or 'auto'} {e.encode_plan(B)}"] = round(sorted(res[c])[len(res[c]) // 2], 3)
    e.set_option("enc_chunk", 0)
    print(json.dumps({"images": B, "ms": line}), flush=True)
