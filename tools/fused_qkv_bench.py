#!/usr/bin/env python3
"""Round 6, item 4: in_proj + attention of one vision block fused per (image, head) (csrc/experiments/qkv_attn_fused.hip, experiment build)
against today's two launches (gemm_f16_v5 with ln_1 folded -> attention variant 3) on the same inputs: outputs compared bit for bit,
both timed with HIP events inside ovmr_debug_qkv_attn (mean of --reps launches)."""
import argparse, ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OVMR_HIP_LIB", os.path.join(ROOT, "ovmr_amd", "lib", "libovmr_hip_exp.so"))
import torch
from ovmr_amd import runtime

ap = argparse.ArgumentParser()
ap.add_argument("--batches", type=int, nargs="+", default=[8, 64, 256, 775])
ap.add_argument("--reps", type=int, default=10)
args = ap.parse_args()
lib = runtime.load_library()
lib.ovmr_debug_qkv_attn.restype = ctypes.c_int
lib.ovmr_debug_qkv_attn.argtypes = [ctypes.c_void_p] * 7 + [ctypes.c_int] * 4 + [ctypes.POINTER(ctypes.c_float), ctypes.c_void_p]
p = lambda t: ctypes.c_void_p(t.data_ptr())
L, W = 197, 768
g = torch.Generator(device="cuda").manual_seed(1)
in_w = (torch.randn((3 * W, W), generator=g, device="cuda") * W ** -0.5).half()
in_b = (torch.randn((3 * W,), generator=g, device="cuda") * 0.1).half()
gamma = torch.exp(torch.randn((W,), generator=g, device="cuda") * 0.3)
beta = torch.randn((W,), generator=g, device="cuda") * 0.1
for B in args.batches:
    x = (torch.randn((B * L, W), generator=g, device="cuda") * 1.5 + 0.2).half()
    x[::37, 5] = 30.0                                      # a few large activations, as a trained residual stream has
    ref = torch.zeros((B * L, W), dtype=torch.float16, device="cuda")
    fus = torch.full((B * L, W), float("nan"), dtype=torch.float16, device="cuda")
    us = (ctypes.c_float * 3)()
    rc = lib.ovmr_debug_qkv_attn(p(x), p(in_w), p(in_b), p(gamma), p(beta), p(ref), p(fus), B, L, W, args.reps, us, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    same = bool(torch.equal(ref, fus))
    d = (ref.float() - fus.float()).abs()
    print(json.dumps({"images": B, "rc": rc, "in_proj_us": round(us[0], 1), "attention_us": round(us[1], 1), "pair_us": round(us[0] + us[1], 1),
                      "fused_us": round(us[2], 1), "fused_over_pair": round(us[2] / (us[0] + us[1]), 3), "bit_equal": same,
                      "max_abs_diff": float(d.max()), "finite": bool(torch.isfinite(fus).all()), "rows_differing": int((d.max(1).values > 0).sum())}), flush=True)
