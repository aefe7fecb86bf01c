#!/usr/bin/env python3
"""Race screen for the hand-synchronised kernels (CDNA4 guide: "a sync-structure edit makes a NEW template: screen it for races
over many runs at several sizes").  Every launch of a kernel on fixed inputs must reproduce the first launch bit for bit; the
counted-vmcnt / staggered-barrier GEMM K loop (variant 8, also gathering patch rows from an image tensor) and the attention kernels with
asm LDS-DMA (variants 3 and 5) are run
--reps times per shape, interleaved with a cache-thrashing copy so that DMA arrival times vary.

    python tools/race_screen.py [--reps 200]
"""
import argparse
import ctypes
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ovmr_amd import runtime

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=200)
a = ap.parse_args()
lib = runtime.load_library()
p = lambda t: ctypes.c_void_p(t.data_ptr())
s = lambda: ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)
junk = torch.empty(64 << 20, dtype=torch.float32, device="cuda")
bad = 0
for (M, N, K, epi) in ((4096, 1024, 768, 2), (5500, 768, 3072, 3), (2500, 2304, 768, 1), (100864, 3072, 768, 2), (100864, 768, 768, 3),
                       (16400, 256, 128, 0), (50432, 768, 3072, 3), (8000, 1536, 512, 1)):
    g = torch.Generator(device="cuda").manual_seed(M + N)
    A = (torch.randn((M, K), generator=g, device="cuda") * 0.5).half()
    W = (torch.randn((N, K), generator=g, device="cuda") * K ** -0.5).half()
    b = (torch.randn((N,), generator=g, device="cuda") * 0.1).half()
    res = torch.randn((M, N), generator=g, device="cuda").half()
    ref, C = None, torch.empty_like(res)
    reps = a.reps if M < 50000 else max(20, a.reps // 5)
    mism = 0
    for rep in range(reps):
        C.copy_(res)
        if rep % 3 == 0:
            junk.add_(1.0)                                    # evict L2 / Infinity Cache contents between launches
        assert lib.ovmr_debug_gemm(0, 8, p(A), p(W), p(b), p(C) if epi == 3 else None, None, p(C), M, N, K, N, epi, 1.0, 0, 0, s()) == 0
        if ref is None:
            ref = C.clone()
        elif not torch.equal(C, ref):
            mism += 1
    bad += mism
    print(f"gemm variant 8 {(M, N, K, epi)}: {reps} launches, {mism} differ from the first", flush=True)
# (round 4: the 64 x 64 split-K kernel -- four waves' partial tiles summed through LDS behind ONE barrier, prefetch depths 1 / 3 / 4 -- on
#  latency-bound shapes incl. the in-place residual epilogue; variant 9 forces it)
for (M, N, K, epi) in ((40, 512, 2048, 3), (775, 768, 3072, 3), (1500, 512, 512, 1), (256, 3072, 768, 2), (65, 130, 384, 1), (63, 72, 640, 3), (2000, 100, 512, 5)):
    g = torch.Generator(device="cuda").manual_seed(M + N)
    A = (torch.randn((M, K), generator=g, device="cuda") * 0.5).half()
    W = (torch.randn((N, K), generator=g, device="cuda") * K ** -0.5).half()
    b = (torch.randn((N,), generator=g, device="cuda") * 0.1).half()
    res = torch.randn((M, N), generator=g, device="cuda").half()
    ref, C, mism = None, torch.empty_like(res), 0
    for rep in range(a.reps):
        C.copy_(res)
        if rep % 3 == 0:
            junk.add_(1.0)
        assert lib.ovmr_debug_gemm(0, 9, p(A), p(W), p(b), p(C) if epi == 3 else None, None, p(C), M, N, K, N, epi, 10.0, 0, 0, s()) == 0
        if ref is None:
            ref = C.clone()
        elif not torch.equal(C, ref):
            mism += 1
    bad += mism
    print(f"gemm split-K (variant 9) {(M, N, K, epi)}: {a.reps} launches, {mism} differ from the first", flush=True)
# (variant 5: the 32x32x16 flash kernel of the ViT-L lengths -- asm LDS-DMA into a two-stage ring, one barrier per key block, 3- and 4-wave workgroups)
for variant, shapes in ((3, ((512, 197, 12), (37, 197, 12), (3, 208, 4), (300, 193, 12))),
                        (5, ((64, 577, 16), (5, 577, 3), (96, 257, 16), (7, 300, 5), (9, 288, 1)))):
  for (B, L, H) in shapes:
    g = torch.Generator(device="cuda").manual_seed(B + L)
    qkv = torch.randn((B * L, 3 * H * 64), generator=g, device="cuda").half()
    out, ref, mism = torch.empty((B * L, H * 64), dtype=torch.float16, device="cuda"), None, 0
    for rep in range(a.reps):
        out.zero_()
        if rep % 3 == 0:
            junk.add_(1.0)
        assert lib.ovmr_debug_attention(0, variant, p(qkv), p(out), B, L, H, 0, s()) == 0
        if ref is None:
            ref = out.clone()
        elif not torch.equal(out, ref):
            mism += 1
    bad += mism
    print(f"attention variant {variant} {(B, L, H)}: {a.reps} launches, {mism} differ from the first", flush=True)
# the whole image tower (patch rows gathered by the GEMM, LayerNorm folding, CLS-only last block) on a small model
from ovmr_amd import modules, synth
spec = synth.SPECS["small"]
e = modules.CLIPModel({k: torch.from_numpy(v) for k, v in synth.clip_state_dict(spec, 11, jitter=True).items()}, spec).engine(2)
e.load_state_dict({}, {k: torch.from_numpy(v) for k, v in synth.prompt_learner_state_dict(spec, 2, 11, True).items()})
e._pl_loaded = True
e.finalize(300, 64, 256)
img = torch.from_numpy(synth.images(300, spec.image_resolution, seed=3)).half().cuda()
ref, mism = None, 0
for rep in range(a.reps):
    if rep % 3 == 0:
        junk.add_(1.0)
    f = e.encode_image(img, normalize=False)
    if ref is None:
        ref = f.clone()
    elif not torch.equal(f, ref):
        mism += 1
bad += mism
print(f"image tower, small model, 300 images: {a.reps} encodes, {mism} differ from the first", flush=True)
# (round 5: the one-launch classifier head -- tiles by ticket, device-coherent per-tile statistics, a count of finished tiles, above 16 class
#  tiles a second ticketed merge phase, self re-arming counters -- alone, with its grid capped (workgroups take several tiles: the recompute
#  queue), and from TWO handles on two streams at once (the test loop's two batches in flight))
from ovmr_amd.runtime import Engine
tiny = synth.ModelSpec("head", 512, 32, 1, 128, 16, 77, 1000, 512, 8, 1)
engines = []
for _ in range(2):
    en = Engine(tiny, 2)
    en.load_state_dict({k: torch.from_numpy(v) for k, v in synth.clip_state_dict(tiny, 1).items()},
                       {k: torch.from_numpy(v) for k, v in synth.prompt_learner_state_dict(tiny, 2, 1).items()})
    en.finalize(64, 64, 22000)
    en.set_option("fused_head", 2)
    engines.append(en)
streams = [torch.cuda.Stream(), torch.cuda.Stream()]
for (B, C, cap) in ((256, 1000, 0), (256, 10000, 0), (37, 1003, 0), (256, 1000, 5), (256, 10000, 100), (512, 21841, 0)):
    g = torch.Generator(device="cuda").manual_seed(B + C)
    f = torch.nn.functional.normalize(torch.randn((B, 512), generator=g, device="cuda"), dim=-1).half()
    clf = [torch.nn.functional.normalize(torch.randn((C, 512), generator=g, device="cuda"), dim=-1).half() for _ in range(3)]
    w = torch.softmax(torch.randn((C, 3), generator=g, device="cuda"), -1)
    for en in engines:
        en.set_option("head_max_grid", cap)
    ref, mism, reps = None, 0, max(20, a.reps // 2)
    torch.cuda.synchronize()
    for rep in range(reps):
        if rep % 3 == 0:
            junk.add_(1.0)
        torch.cuda.synchronize()
        outs = []
        for en, st in zip(engines, streams):                      # the two handles' launches overlap on the device
            with torch.cuda.stream(st):
                outs.append(en.fused_logits(f, *clf, w, "fusion"))
        torch.cuda.synchronize()
        if ref is None:
            ref = outs[0].clone()
        mism += sum(0 if torch.equal(o, ref) else 1 for o in outs)
    bad += mism
    print(f"one-launch head {(B, C)} grid cap {cap}: {reps} x 2 concurrent launches, {mism} differ from the first", flush=True)
torch.cuda.synchronize()
print("RACE SCREEN", "CLEAN" if bad == 0 else f"FAILED ({bad} differing launches)")
sys.exit(1 if bad else 0)
