#!/usr/bin/env python3
"""Where the 64 x 64 split-K kernel (gemm_f16_small.hip, variant 9) beats the tile kernels (variant 7 = 8 without it) on the launch
shapes of a small shard's head: text tower (W = 512), CLS-row chain of the last vision block (W = 768), logits.  HIP events on the
launch stream, variants interleaved in one process, cache-cold operands rotate through 8 copies.

    python tools/small_gemm_bench.py [--rounds 5] [--reps 20]
"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from ovmr_amd import runtime


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=20)
    ap.add_argument("--variants", type=int, nargs="+", default=[7, 9])
    args = ap.parse_args()
    lib = runtime.load_library()
    dev = "cuda"
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    s = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    shapes = [("text_in_proj", 1536, 512, 1), ("text_out_proj", 512, 512, 3), ("text_c_fc", 2048, 512, 2), ("text_c_proj", 512, 2048, 3),
              ("vis_out_proj", 768, 768, 3), ("vis_c_fc", 3072, 768, 2), ("vis_c_proj", 768, 3072, 3), ("proj", 512, 768, 0), ("logits", 1000, 512, 5)]
    out = {}
    for name, n, k, epi in shapes:
        for m in (40, 256, 775, 1500, 2600, 3250, 5000, 8192, 16384):
            g = torch.Generator(device=dev).manual_seed(1)
            A = (torch.randn((m, k), generator=g, device=dev) * 0.5).half()
            W = (torch.randn((n, k), generator=g, device=dev) * k ** -0.5).half()
            b = (torch.randn((n,), generator=g, device=dev) * 0.1).half()
            C = torch.zeros((m, n), dtype=torch.float16, device=dev)
            call = lambda v: lib.ovmr_debug_gemm(0, v, p(A), p(W), p(b), p(C) if epi == 3 else None, None, p(C), m, n, k, n, epi, 100.0, 0, 0, s())
            res = {}
            for v in args.variants:
                for _ in range(3):
                    assert call(v) == 0
            torch.cuda.synchronize()
            for r in range(args.rounds):
                for v in args.variants:
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record()
                    for _ in range(args.reps):
                        call(v)
                    e1.record()
                    torch.cuda.synchronize()
                    res.setdefault(v, []).append(e0.elapsed_time(e1) * 1000 / args.reps)
            med = {v: sorted(t)[len(t) // 2] for v, t in res.items()}
            tiles = ((m + 127) // 128) * ((n + 255) // 256)
            out[f"{name}_{m}"] = {"M": m, "N": n, "K": k, "tiles_128x256": tiles, **{f"v{v}_us": round(x, 1) for v, x in med.items()}}
            print(f"{name:14s} M {m:6d} N {n:5d} K {k:5d} tiles {tiles:4d}  " + "  ".join(f"v{v} {x:7.1f} us" for v, x in med.items()), flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
