#!/usr/bin/env python3
"""Experiment build only: gemm variant 10 (experiments/gemm_f16_w2.hip, two workgroups per CU) against variant 8 -- results, then time."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("OVMR_HIP_LIB", os.path.join(ROOT, "ovmr_amd", "lib", "libovmr_hip_exp.so"))
import torch
from ovmr_amd import runtime
lib = runtime.load_library()
p = lambda t: ctypes.c_void_p(t.data_ptr())
s = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
for (M, N, K, epi) in ((5000, 768, 768, 3), (1000, 3072, 768, 1), (257, 128, 3072, 3), (4097, 2304, 768, 0)):
    g = torch.Generator(device="cuda").manual_seed(M)
    A = (torch.randn((M, K), generator=g, device="cuda") * 0.5).half()
    W = (torch.randn((N, K), generator=g, device="cuda") * K ** -0.5).half()
    b = (torch.randn((N,), generator=g, device="cuda") * 0.1).half()
    res = torch.randn((M, N), generator=g, device="cuda").half()
    outs = []
    for v in (8, 10):
        C = res.clone()
        assert lib.ovmr_debug_gemm(0, v, p(A), p(W), p(b), p(C) if epi == 3 else None, None, p(C), M, N, K, N, epi, 1.0, 0, 0, s()) == 0
        outs.append(C)
    torch.cuda.synchronize()
    d = (outs[0].float() - outs[1].float()).abs().max().item()
    print((M, N, K, epi), "max |v8 - v10| =", d, "equal" if torch.equal(outs[0], outs[1]) else "", flush=True)
    assert d <= 4e-3 * max(1.0, outs[0].float().abs().max().item())
