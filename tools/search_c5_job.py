#!/usr/bin/env python3
"""Searches, on the CPU oracle, the parameters of the whole-c5 parity job (tests/test_hip_parity.py:C5_JOB: ViT-L/14@336px,
a few classes x shots) for clear cross-validation margins: prints, per candidate, the smallest top-2 margin of the oracle's
argmaxes for the three classifiers and the classes a near-tie (< 0.26) could touch.  CPU only; minutes per candidate.

    python tools/search_c5_job.py [--gains 1.5 3] [--classes 3] [--shots 2]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import torch


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--model", default="ViT-L/14@336px")
    ap.add_argument("--gains", type=float, nargs="+", default=[1.5, 3.0])
    ap.add_argument("--classes", type=int, default=3)
    ap.add_argument("--shots", type=int, default=2)
    ap.add_argument("--strength", type=float, default=0.9)
    ap.add_argument("--tile", type=int, default=14)
    ap.add_argument("--pool", type=int, default=64)
    ap.add_argument("--tau", type=float, default=3.0)
    a = ap.parse_args()
    from oracle import ovmr_oracle as O
    from ovmr_amd import synth
    from conftest import near_tie_classes
    import test_hip_parity as T
    spec = synth.SPECS[a.model]
    for gain in a.gains:
        sd_np, pl_np, sd, labels, pattern, img, tok, f = T.aligned_job(O, spec, a.classes, a.shots, gain, a.strength, a.tile, pool=a.pool)
        with torch.no_grad():
            r = O.forward_prompt(torch.from_numpy(img), torch.from_numpy(labels), tok, sd, O.to_torch(pl_np), 2, a.tau, a.classes, "fp16")
        ls = sd["logit_scale"].float().exp()
        rep, affected = {}, set()
        for k in ("mm_classifier", "vision_classifier", "text_classifier"):
            lg = O.cross_validation_logits(r["eval_feat4cls"], r[k].half(), ls).float().numpy()
            srt = np.sort(lg, axis=1)
            rep[k] = round(float((srt[:, -1] - srt[:, -2]).min()), 3)
            affected |= near_tie_classes(lg, 0.26)
        print(f"gain {gain}: min top-2 margins {rep}, classes touched by near-ties {sorted(affected)}\n  fusion_weight\n{np.round(r['fusion_weight'].numpy(), 3)}", flush=True)


if __name__ == "__main__":
    main()
