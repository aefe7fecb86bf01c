#!/usr/bin/env python3
"""Does replaying the classifier head of a small shard as a HIP graph shorten it?  The head of 125 classes (token generator +
one text-tower pass over three prompt groups) is ~180 launches of 5-10 us with ~5 us of command-processor hand-over between
dependent launches.  Eager launches against torch.cuda.CUDAGraph replay of the same calls, ViT-B/16 weights, HIP events.

    python tools/graph_head_probe.py [--classes 125] [--shots 16]
"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from ovmr_amd import modules, synth


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--classes", type=int, nargs="+", default=[16, 125, 1000])
    ap.add_argument("--shots", type=int, default=16)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    spec = synth.SPECS["ViT-B/16"]
    gen = torch.Generator(device=dev).manual_seed(1234)
    cm = modules.CLIPModel(bench.device_clip_state(spec, gen, dev), spec, str(dev))
    pl_state = bench.device_pl_state(spec, 2, gen, dev)
    for C in a.classes:
        cfg = modules.make_cfg(n_ctx=2, num_shots=a.shots, output_dir="")
        tok = torch.from_numpy(synth.class_token_ids(C, seed=4321))
        model = modules.CustomCLIP(cfg, tok, cm, prompt_learner_state=pl_state, reserve=(64, max(256, C), max(1024, C)), stream_text=True)
        pl, e = model.prompt_learner, model.engine
        feats = torch.nn.functional.normalize(torch.randn((C, a.shots, spec.embed_dim), generator=gen, device=dev), dim=-1).half()
        label = torch.arange(C, device=dev)

        def head():
            mm_p, mm_l, v_p, v_l, tokens = pl(feats, label, pl.eos_index[label])
            return model.get_mm_v_feats(mm_p, mm_l, v_p, v_l, model.tokenized_prompts[label])

        for _ in range(3):
            want = head()
        torch.cuda.synchronize()

        def timed(fn, reps=20):
            ts = []
            for _ in range(reps):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record(); fn(); e1.record()
                torch.cuda.synchronize()
                ts.append(e0.elapsed_time(e1) * 1000)
            return sorted(ts)[len(ts) // 2]

        t_eager = timed(head)
        g = torch.cuda.CUDAGraph()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            head()
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=side):
                got = head()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        g.replay()
        torch.cuda.synchronize()
        same = all(torch.equal(x, y) for x, y in zip(got, want))
        t_graph = timed(g.replay)
        print(f"{C:5d} classes x {a.shots} shots: head eager {t_eager:8.1f} us   graph replay {t_graph:8.1f} us   bit-equal {same}", flush=True)
        del model, g
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
