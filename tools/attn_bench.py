#!/usr/bin/env python3
"""Times the fp16 attention kernel variants at the image-tower shape (B x 12 heads, L = 197) and the text shape."""
import ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# variant 4's timing-only ablations (400 + mode, attention_v4.hip) live in the experiment build only
# The product library is timed unless --exp is given (experiment build: ablation variants and A/B switches); the library loaded is printed.
if "--exp" in sys.argv:
    os.environ.setdefault("OVMR_HIP_LIB", os.path.join(ROOT, "ovmr_amd", "lib", "libovmr_hip_exp.so"))
import torch
from ovmr_amd import runtime

import argparse
ap = argparse.ArgumentParser()
ap.add_argument("--only", default="")
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--batch", type=int, default=512, help="images of the image-tower shape")
ap.add_argument("--variants", type=int, nargs="+", default=[0, 1, 3])
ap.add_argument("--exp", action="store_true", help="time libovmr_hip_exp.so (experiment build: variant 4 and its ablation modes) instead of the product library")
args = ap.parse_args()
lib = runtime.load_library()
print("library:", os.environ.get("OVMR_HIP_LIB", runtime.LIB_PATH), flush=True)
p = lambda t: ctypes.c_void_p(t.data_ptr())
s = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for name, B, L, H, causal in (("image", args.batch, 197, 12, 0), ("text", 1000, 10, 8, 1), ("text77", 256, 77, 8, 1), ("vit-l336", 64, 577, 16, 0),
                               ("vit-l336-b128", 128, 577, 16, 0), ("vit-l224", 256, 257, 16, 0)):
    if args.only and name != args.only:
        continue
    qkv = torch.randn((B * L, 3 * H * 64), device="cuda").half()
    out = torch.empty((B * L, H * 64), device="cuda", dtype=torch.float16)
    res = {}
    for r in range(3):
        for v in args.variants:
            for _ in range(2):
                assert lib.ovmr_debug_attention(0, v, p(qkv), p(out), B, L, H, causal, s()) == 0
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(args.reps):
                lib.ovmr_debug_attention(0, v, p(qkv), p(out), B, L, H, causal, s())
            e1.record(); torch.cuda.synchronize()
            res.setdefault(v, []).append(e0.elapsed_time(e1) * 1000 / args.reps)
    fl = 4.0 * B * H * L * L * 64
    print(name, {f"v{v}": {"us": round(min(t), 1), "tflops": round(fl / min(t) / 1e6, 1)} for v, t in res.items()}, flush=True)
