import os, sys, ctypes, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["OVMR_HIP_LIB"] = os.path.join(ROOT, "ovmr_amd", "lib", "libovmr_hip_exp.so")
from ovmr_amd import runtime
lib = runtime.load_library()
p = lambda t: ctypes.c_void_p(t.data_ptr())
s = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
ok = True
for (M, N, K, epi) in ((4096, 1024, 768, 2), (5500, 768, 3072, 3), (2500, 2304, 768, 1), (100864, 3072, 768, 2), (100864, 768, 768, 3), (16400, 256, 128, 0)):
    g = torch.Generator(device="cuda").manual_seed(M + N)
    A = (torch.randn((M, K), generator=g, device="cuda") * 0.5).half()
    W = (torch.randn((N, K), generator=g, device="cuda") * K ** -0.5).half()
    b = (torch.randn((N,), generator=g, device="cuda") * 0.1).half()
    res = torch.randn((M, N), generator=g, device="cuda").half()
    outs = []
    for v in (8, 9):
        for rep in range(3):
            C = res.clone()
            rc = lib.ovmr_debug_gemm(0, v, p(A), p(W), p(b), p(C) if epi == 3 else None, None, p(C), M, N, K, N, epi, 1.0, 0, 0, s())
            assert rc == 0
            torch.cuda.synchronize()
            outs.append(C)
    same = all(torch.equal(outs[0], o) for o in outs[1:])
    ok &= same
    print((M, N, K, epi), "variant 9 == variant 8 over 3 runs each:", same, flush=True)
print("ALL EQUAL" if ok else "MISMATCH")
