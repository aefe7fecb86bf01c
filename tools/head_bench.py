#!/usr/bin/env python3
"""Classifier head (ovmr_fused_logits, fusion mode): one launch (head_fused.hip) against the five-launch path; 20 calls captured
into a hipGraph, 10 replays timed with HIP events (device time per call, no per-call host overhead)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if "--acqrel" in sys.argv:      # experiment build: release / acquire on the phase counters of the one-launch head (head_fused.hip hf_signal / hf_wait)
    os.environ["OVMR_HEAD_ACQREL"] = "1"
if "--stamps" in sys.argv or "--acqrel" in sys.argv or "--exp" in sys.argv:
    os.environ.setdefault("OVMR_HIP_LIB", os.path.join(ROOT, "ovmr_amd", "lib", "libovmr_hip_exp.so"))
import torch
from ovmr_amd import synth
from ovmr_amd.runtime import Engine

for name, B, C in (("ViT-B/16", 256, 1000), ("ViT-B/16", 256, 10000), ("ViT-B/16", 256, 21841), ("ViT-B/16", 512, 4096), ("ViT-B/16", 64, 1000), ("ViT-B/16", 2048, 1000), ("ViT-L/14@336px", 128, 1000)):
    spec = synth.SPECS[name]
    D = spec.embed_dim
    # only the head runs: an engine with no tower weights cannot be finalized, so build a tiny stand-in spec of the same embed_dim
    tiny = synth.ModelSpec("head", D, 32, 1, 128, 16, 77, 1000, D, D // 64, 1)
    e = Engine(tiny, 2)
    e.load_state_dict({k: torch.from_numpy(v) for k, v in synth.clip_state_dict(tiny, 1).items()},
                      {k: torch.from_numpy(v) for k, v in synth.prompt_learner_state_dict(tiny, 2, 1).items()})
    e.finalize(64, 64, max(C, 1024))
    g = torch.Generator(device="cuda").manual_seed(1)
    f = torch.nn.functional.normalize(torch.randn((B, D), generator=g, device="cuda"), dim=-1).half()
    clf = [torch.nn.functional.normalize(torch.randn((C, D), generator=g, device="cuda"), dim=-1).half() for _ in range(3)]
    w = torch.softmax(torch.randn((C, 3), generator=g, device="cuda"), -1)
    res = {}
    for tag, fused in (("one_launch", 2), ("five_launches", 0)):
        e.set_option("fused_head", fused)
        for _ in range(5):
            e.fused_logits(f, *clf, w, "fusion")
        torch.cuda.synchronize()
        # 20 calls captured into a hipGraph and replayed: device time without Python / ctypes per-call overhead
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            e.fused_logits(f, *clf, w, "fusion")
            gr = torch.cuda.CUDAGraph()
            with torch.cuda.graph(gr, stream=st):
                for _ in range(20):
                    e.fused_logits(f, *clf, w, "fusion")
            gr.replay()
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record(st)
            for _ in range(10):
                gr.replay()
            e1.record(st)
            torch.cuda.synchronize()
        res[tag + "_us"] = round(e0.elapsed_time(e1) * 1000.0 / 200, 1)
    if "--stamps" in sys.argv:
        import ctypes
        e.set_option("fused_head", 2)
        e.fused_logits(f, *clf, w, "fusion")
        torch.cuda.synchronize()
        buf = (ctypes.c_longlong * 16)()
        e.lib.ovmr_debug_head_stamps.argtypes = [ctypes.c_void_p]
        e.lib.ovmr_debug_head_stamps(buf)
        names = ["start", "ticket", "compute", "stats", "release+done", "all done", "acquire", "emit", "exit"]
        res["stamps_shader_cycles"] = {names[i + 1]: int(buf[i + 1] - buf[i]) for i in range(8)}
    print(json.dumps({"embed_dim": D, "queries": B, "classes": C, **res}), flush=True)
