#!/usr/bin/env python3
"""Where a K-tile of the ping-pong GEMM loop spends its cycles: shader-clock stamps (s_memtime) of one workgroup of the c_fc-shaped
launch, taken by the experiment build's variant 58 at the three points of each phase where no LDS read is outstanding -- phase
start, fragments in registers (behind lgkmcnt(0), in front of the mid barrier), MFMA start, MFMA end.  Per wave: load work (fragment
reads + LDS-DMA issue + counted wait), wait at the mid barrier, MFMA segment (32 MFMAs), wait at the phase-end barrier.

    python -m ovmr_amd.build --experiments && python tools/gemm_stamps.py [--n 3072] [--k 768]
"""
import argparse
import ctypes
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OVMR_HIP_LIB", os.path.join(ROOT, "ovmr_amd", "lib", "libovmr_hip_exp.so"))
import numpy as np
import torch
from ovmr_amd import runtime

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=768)
ap.add_argument("--n", type=int, default=3072)
ap.add_argument("--k", type=int, default=768)
args = ap.parse_args()
lib = runtime.load_library()
p = lambda t: ctypes.c_void_p(t.data_ptr())
s = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
m, n, k = args.batch * 197, args.n, args.k
g = torch.Generator(device="cuda").manual_seed(1)
A = (torch.randn((m, k), generator=g, device="cuda") * 0.5).half()
W = (torch.randn((n, k), generator=g, device="cuda") * k ** -0.5).half()
b = torch.zeros(n, dtype=torch.float16, device="cuda")
C = torch.empty((m, n), dtype=torch.float16, device="cuda")
st = torch.zeros((8, 4096), dtype=torch.int64, device="cuda")
for _ in range(5):
    assert lib.ovmr_debug_gemm(0, 58, p(A), p(W), p(b), None, p(st), p(C), m, n, k, n, 1, 1.0, 0, 0, s()) == 0
torch.cuda.synchronize()
t = st.cpu().numpy()
nk = k // 64
per = 16 * (nk // 2)                                   # stamps per wave: 4 per phase, 4 phases per pair of K-tiles
out = {}
for wave in range(8):
    x = t[wave, :per].astype(np.int64).reshape(-1, 4)     # [phase][start, fragments in registers (before the mid barrier), mfma_start, mfma_end]
    load = x[:, 1] - x[:, 0]
    midw = x[:, 2] - x[:, 1]
    mfma = x[:, 3] - x[:, 2]
    endw = x[1:, 0] - x[:-1, 3]
    out[f"wave{wave} (row {wave >> 2})"] = {"load_work": [int(load[2:].mean()), int(load[2:].min()), int(load[2:].max())],
                                            "mid_barrier": [int(midw[2:].mean()), int(midw[2:].min()), int(midw[2:].max())],
                                            "mfma_seg": [int(mfma.mean()), int(mfma.min()), int(mfma.max())],
                                            "end_barrier": [int(endw.mean()), int(endw.min()), int(endw.max())],
                                            "k_tile_cycles": int((x[-1, 3] - x[0, 0]) / nk)}
    if wave in (0, 4):
        print(f"wave {wave}: first 16 phases [load work, mid barrier, mfma, end barrier]:", [(int(a), int(b_), int(c), int(e)) for a, b_, c, e in zip(load[:16], midw[:16], mfma[:16], list(endw[:16]))])
print(json.dumps(out, indent=1))
