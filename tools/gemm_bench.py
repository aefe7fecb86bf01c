#!/usr/bin/env python3
"""Times the hot-path GEMM launch shapes for each kernel variant with HIP events on the
launch stream, interleaving variants in one process (cdna guide rule 24).  Random operands (rule 25).

    python tools/gemm_bench.py [--batch 256] [--variants 0 1] [--rounds 5]
"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# The product library is timed unless --exp is given: the experiment build (python -m ovmr_amd.build --experiments) carries the A/B
# environment switches and the timing-only ablation variants; the library actually loaded is printed.
if "--exp" in sys.argv:
    os.environ.setdefault("OVMR_HIP_LIB", os.path.join(ROOT, "ovmr_amd", "lib", "libovmr_hip_exp.so"))
import torch
from ovmr_amd import runtime


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--variants", type=int, nargs="+", default=[0, 1])
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--reps", type=int, default=10)
    ap.add_argument("--zeros", action="store_true", help="zero-filled operands: shows how much of the time is clock/power (cdna guide rule 25)")
    ap.add_argument("--ldpad", type=int, default=0, help="row padding (halves) of A and W: stride experiment")
    ap.add_argument("--shapes", default="", help="comma-separated shape names to run (default: all) -- one name per profiler pass gives per-shape "
                    "counters for launches that share a kernel instantiation AND a grid (out_proj / c_proj: tools/pmc_gemm.sh)")
    ap.add_argument("--exp", action="store_true", help="time libovmr_hip_exp.so (experiment build) instead of the product library")
    args = ap.parse_args()
    if args.ldpad:
        os.environ["OVMR_DEBUG_LDPAD"] = str(args.ldpad)
    lib = runtime.load_library()
    print("library:", os.environ.get("OVMR_HIP_LIB", runtime.LIB_PATH), flush=True)
    dev = "cuda"
    M = args.batch * 197
    shapes = [("qkv", M, 2304, 768, 1), ("out_proj", M, 768, 768, 3), ("c_fc", M, 3072, 768, 2), ("c_fc_bias_only", M, 3072, 768, 1), ("n1536_bias", M, 1536, 768, 1), ("c_proj", M, 768, 3072, 3),
              ("qkv_ln", M, 2304, 768, 6), ("c_fc_ln", M, 3072, 768, 7), ("out_proj_st", M, 768, 768, 13), ("c_proj_st", M, 768, 3072, 13),
              ("cls_out_proj", args.batch, 768, 768, 3), ("cls_c_fc", args.batch, 3072, 768, 2), ("cls_c_proj", args.batch, 768, 3072, 3),   # last block: CLS rows only
              ("patch", args.batch * 196, 768, 768, 0), ("text_qkv", 1000 * 10, 1536, 512, 1), ("xval_logits", 16000, 1000, 512, 5)]
    p = lambda t: ctypes.c_void_p(t.data_ptr())
    s = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
    out = {}
    only = {x for x in args.shapes.split(",") if x}
    for name, m, n, k, epi in shapes:
        if only and name not in only:
            continue
        g = torch.Generator(device=dev).manual_seed(1)
        A = (torch.randn((m + 256, k + args.ldpad), generator=g, device=dev) * 0.5).half()   # slack rows: blocked-layout experiment
        W = (torch.randn((n + 256, k + args.ldpad), generator=g, device=dev) * k ** -0.5).half()
        if args.zeros:
            A.zero_(); W.zero_()
        b = (torch.randn((n,), generator=g, device=dev) * 0.1).half()
        C = torch.zeros((m, n), dtype=torch.float16, device=dev)
        res = {}
        # LayerNorm-folding epilogues (csrc/common.h): 6/7 consume row statistics, 13 = BIAS_RES that also emits them
        st = torch.zeros((m, max(k, n) // 256 + 1, 2), device=dev)
        st[:, 0, 1] = float(k)
        lg, lb = torch.zeros(n, device=dev), torch.zeros(n, device=dev)
        if epi in (6, 7):
            call = lambda v: lib.ovmr_debug_gemm(0, v, p(A), p(W), p(lb), p(st), p(lg), p(C), m, n, k, n, epi, 100.0, 0, 0, s())
        elif epi == 13:
            call = lambda v: lib.ovmr_debug_gemm(0, v, p(A), p(W), p(b), p(C), p(st), p(C), m, n, k, n, 3, 100.0, 0, 0, s())
        else:
            call = lambda v: lib.ovmr_debug_gemm(0, v, p(A), p(W), p(b), p(C) if epi == 3 else None, None, p(C), m, n, k, n, epi, 100.0, 0, 0, s())
        if epi in (6, 7, 13) and any(12 <= v <= 19 or v in (28, 29, 59) for v in args.variants):
            continue                                  # timing-only ablation variants carry no LN-folding epilogues
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
        fl = 2.0 * m * n * k
        out[name] = {f"v{v}": {"us_median": round(sorted(t)[len(t) // 2], 1), "us_min": round(min(t), 1),
                               "tflops_median": round(fl / sorted(t)[len(t) // 2] / 1e6, 1)} for v, t in res.items()}
        print(name, (m, n, k), json.dumps(out[name]), flush=True)


if __name__ == "__main__":
    main()
