#!/usr/bin/env python3
"""Latency of one ovmr_encode_image call at serving batch sizes (1 ... 32 images), ViT-B/16 and ViT-L/14@336px: the launch shapes of
such calls are latency-bound (a fraction of one round of tiles), which is what the 64 x 64 split-K GEMM of round 4 is for.
A/B in one process: option "gemm" 8 (default) against 7 (8 without the split-K kernel).  HIP events, median of 20 calls.

    python tools/latency_probe.py [--models ViT-B/16 ViT-L/14@336px]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ovmr_amd import modules, synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--models", nargs="+", default=["ViT-B/16", "ViT-L/14@336px"])
    ap.add_argument("--batches", type=int, nargs="+", default=[1, 2, 4, 8, 16, 32])
    a = ap.parse_args()
    sys.path.insert(0, ROOT)
    import bench
    dev = torch.device("cuda:0")
    out = {}
    for name in a.models:
        spec = synth.SPECS[name]
        gen = torch.Generator(device=dev).manual_seed(1234)
        cm = modules.CLIPModel(bench.device_clip_state(spec, gen, dev), spec, str(dev))
        e = cm.engine(2)
        e.load_state_dict({}, bench.device_pl_state(spec, 2, gen, dev))
        e._pl_loaded = True
        e.finalize(max(a.batches), 8, 8)
        for B in a.batches:
            img = torch.randn((B, 3, spec.image_resolution, spec.image_resolution), generator=gen, device=dev).half()
            res = {}
            for v in (7, 8):
                e.set_option("gemm", v)
                for _ in range(3):
                    e.encode_image(img)
                torch.cuda.synchronize()
                ts = []
                for _ in range(20):
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(); e.encode_image(img); e1.record()
                    torch.cuda.synchronize()
                    ts.append(e0.elapsed_time(e1) * 1000)
                res[v] = sorted(ts)[len(ts) // 2]
            e.set_option("gemm", 8)
            out[f"{name} B={B}"] = {"tile_kernels_us": round(res[7], 1), "with_split_k_us": round(res[8], 1), "images_per_s": round(B / res[8] * 1e6, 1)}
            print(f"{name:16s} B {B:3d}  tile kernels only {res[7]:8.1f} us   with the split-K kernel {res[8]:8.1f} us  ({B / res[8] * 1e6:8.1f} img/s)", flush=True)
        del e, cm
        torch.cuda.empty_cache()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
