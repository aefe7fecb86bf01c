#!/usr/bin/env python3
"""Reference point only: what the vendor library (torch.matmul -> hipBLASLt / rocBLAS) reaches on the hot GEMM shapes, with
and without the fused work our epilogues do (bias add, QuickGELU, residual) as separate torch ops."""
import json, torch
dev = "cuda"
import sys
M = (int(sys.argv[1]) if len(sys.argv) > 1 else 775) * 197     # images x tokens (bench.py: 775 images per launch sequence)
def t(fn, reps=10):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1000 / reps
for name, n, k in (("qkv", 2304, 768), ("out_proj", 768, 768), ("c_fc", 3072, 768), ("c_proj", 768, 3072)):
    A = (torch.randn(M, k, device=dev) * 0.5).half(); W = (torch.randn(n, k, device=dev) * k ** -0.5).half()
    b = torch.randn(n, device=dev).half(); R = torch.randn(M, n, device=dev).half()
    us_mm = t(lambda: torch.matmul(A, W.t()))
    us_lin = t(lambda: torch.nn.functional.linear(A, W, b))
    if name == "c_fc":
        def full():
            u = torch.nn.functional.linear(A, W, b); return u * torch.sigmoid(1.702 * u)
    elif name in ("out_proj", "c_proj"):
        def full(): return R + torch.nn.functional.linear(A, W, b)
    else:
        def full(): return torch.nn.functional.linear(A, W, b)
    us_full = t(full)
    fl = 2.0 * M * n * k
    print(name, json.dumps({"matmul_us": round(us_mm, 1), "matmul_tflops": round(fl / us_mm / 1e6, 1), "linear_us": round(us_lin, 1),
                            "with_epilogue_ops_us": round(us_full, 1), "with_epilogue_tflops": round(fl / us_full / 1e6, 1)}), flush=True)
