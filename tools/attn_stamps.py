#!/usr/bin/env python3
"""Where a key block of attention variant 5 spends its cycles: shader-clock stamps of one mid-grid workgroup (experiment build, mode 8).
Per wave and block: [wait at the barrier, DMA issue, score MFMA issue, scores back + row maxima, reference check, first two PV steps,
last two PV steps].

    python -m ovmr_amd.build --experiments && python tools/attn_stamps.py [--b 128] [--l 577] [--h 16]
"""
import argparse, ctypes, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OVMR_HIP_LIB", os.path.join(ROOT, "ovmr_amd", "lib", "libovmr_hip_exp.so"))
import numpy as np
import torch
from ovmr_amd import runtime

ap = argparse.ArgumentParser()
ap.add_argument("--b", type=int, default=128)
ap.add_argument("--l", type=int, default=577)
ap.add_argument("--h", type=int, default=16)
args = ap.parse_args()
lib = runtime.load_library()
raw = ctypes.CDLL(os.environ["OVMR_HIP_LIB"])
p = lambda t: ctypes.c_void_p(t.data_ptr())
s = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
B, L, H = args.b, args.l, args.h
qkv = torch.randn((B * L, 3 * H * 64), device="cuda").half()
out = torch.empty((B * L, H * 64), device="cuda", dtype=torch.float16)
for _ in range(3):
    assert lib.ovmr_debug_attention(0, 58, p(qkv), p(out), B, L, H, 0, s()) == 0
torch.cuda.synchronize()
buf = (ctypes.c_longlong * 1024)()
assert raw.ovmr_debug_attn5_stamps(buf, 1024) == 0
t = np.frombuffer(buf, dtype=np.int64).reshape(4, 256)
nb = (L + 63) // 64
nfull = nb - 1 if (L - (nb - 1) * 64) <= 16 and nb > 1 else nb
names = ["barrier_wait", "dma_issue", "score_issue", "scores_back_rowmax", "reference", "pv_01", "pv_23"]
res = {}
for w in range(4):
    x = t[w, :8 * nfull].reshape(nfull, 8)
    d = np.diff(x, axis=1)
    gap = x[1:, 0] - x[:-1, 7]
    res[f"wave{w}"] = {n: [int(d[1:, i].mean()), int(d[1:, i].min()), int(d[1:, i].max())] for i, n in enumerate(names)}
    res[f"wave{w}"]["between_blocks"] = [int(gap.mean()), int(gap.min()), int(gap.max())]
    res[f"wave{w}"]["block_cycles"] = int((x[-1, 7] - x[1, 0]) / (nfull - 1))
    if w == 0:
        print("wave 0, per block:", [[int(v) for v in row] for row in d])
print(json.dumps(res, indent=1))
