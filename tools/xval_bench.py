#!/usr/bin/env python3
"""Cross-validation step (K18-K20: logits of every exemplar row against every class, row argmax, counters, fusion weights) at
large vocabularies: the fused path (row argmax inside the logits GEMM's epilogue, logits never written) against the path that
materialises fp16 logits in workspace chunks.  Synthetic unit-norm features and classifier rows; counters must be identical.

    python tools/xval_bench.py [--classes 10000] [--shots 8] [--dim 512]
"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from ovmr_amd import modules, synth

ap = argparse.ArgumentParser()
ap.add_argument("--classes", type=int, default=10000)
ap.add_argument("--shots", type=int, default=8)
ap.add_argument("--model", default="ViT-B/16")
ap.add_argument("--reps", type=int, default=3)
a = ap.parse_args()
spec = synth.SPECS["tiny"]._replace if False else synth.SPECS[a.model]
dev = "cuda"
# an engine is only needed for its handle (logit scale, workspace): tiny towers, the real embed_dim
import dataclasses
spec = dataclasses.replace(synth.SPECS["tiny"], name="xval", embed_dim=synth.SPECS[a.model].embed_dim,
                           transformer_width=synth.SPECS[a.model].embed_dim, transformer_heads=synth.SPECS[a.model].embed_dim // 64)
sd = {k: torch.from_numpy(v) for k, v in synth.clip_state_dict(spec, 1).items()}
cm = modules.CLIPModel(sd, spec)
e = cm.engine(2)
e.load_state_dict({}, {k: torch.from_numpy(v) for k, v in synth.prompt_learner_state_dict(spec, 2, 1).items()})
e._pl_loaded = True
e.finalize(8, 8, 8)
C, S, D = a.classes, a.shots, spec.embed_dim
g = torch.Generator(device=dev).manual_seed(1)
clf = [torch.nn.functional.normalize(torch.randn(C, D, generator=g, device=dev), dim=-1).half() for _ in range(3)]
lab = torch.arange(C, dtype=torch.int32, device=dev).repeat_interleave(S)
feats = torch.nn.functional.normalize(clf[0][lab.long()].float() + 0.5 * torch.randn(C * S, D, generator=g, device=dev), dim=-1).half()
out = {}
for fused in (1, 0):
    e.set_option("xval_fused", fused)
    ts = []
    for rep in range(a.reps + 1):
        counts = torch.zeros((3, 2, C), dtype=torch.int32, device=dev)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for m in range(3):
            e.xval_counts(feats, lab, clf[m], counts[m, 0], counts[m, 1])
        w = e.fusion_weights(counts, torch.full((C,), S, dtype=torch.int32), 10.0)
        torch.cuda.synchronize()
        if rep:
            ts.append(time.perf_counter() - t0)
    out[fused] = (min(ts), counts.cpu())
flops = 3 * 2.0 * C * S * C * D
assert torch.equal(out[0][1], out[1][1]), "fused and materialised counters differ"
print(json.dumps({"classes": C, "shots": S, "dim": D, "rows": C * S, "logit_elements_per_classifier": C * S * C,
                  "fused_ms": round(out[1][0] * 1e3, 2), "materialised_ms": round(out[0][0] * 1e3, 2),
                  "fused_tflops": round(flops / out[1][0] / 1e12, 1), "materialised_tflops": round(flops / out[0][0] / 1e12, 1),
                  "logits_bytes_not_written": 3 * 2 * C * S * C, "counters_identical": True}))
