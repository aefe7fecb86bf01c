#!/usr/bin/env python3
"""Reference point only: torch.nn.functional.scaled_dot_product_attention (the flash / efficient kernels PyTorch-ROCm ships) on the
attention shapes of the image towers, next to this repository's kernels (tools/attn_bench.py).  fp16, hd = 64, non-causal; q / k / v
as [B, H, L, 64] contiguous tensors (the layout those kernels want; ours read the in-projection's [B*L, 3*H*64] output in place)."""
import json, sys, torch
import torch.nn.functional as F
from torch.nn.attention import SDPBackend, sdpa_kernel

def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / reps

for name, B, L, H in (("ViT-B/16 image", 775, 197, 12), ("ViT-L/14@336 image", 128, 577, 16), ("ViT-L/14 image", 256, 257, 16)):
    q, k, v = (torch.randn((B, H, L, 64), device="cuda").half() for _ in range(3))
    fl = 4.0 * B * H * L * L * 64
    res = {}
    for label, backend in (("flash", SDPBackend.FLASH_ATTENTION), ("efficient", SDPBackend.EFFICIENT_ATTENTION), ("math", SDPBackend.MATH)):
        try:
            with sdpa_kernel(backend):
                us = t(lambda: F.scaled_dot_product_attention(q, k, v))
            res[label] = {"us": round(us, 1), "tflops": round(fl / us / 1e6, 1)}
        except Exception as e:
            res[label] = "unavailable: " + str(e).split("\n")[0][:80]
    print(name, (B, H, L), json.dumps(res), flush=True)
