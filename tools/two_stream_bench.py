#!/usr/bin/env python3
"""Experiment: two handles on two HIP streams, each encoding half-size image batches concurrently, against one handle with
full-size batches.  Question: does the second stream's work fill the partial last round of tiles of every GEMM launch?"""
import argparse, json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ovmr_amd import modules, synth

ap = argparse.ArgumentParser()
ap.add_argument("--images", type=int, default=8192)
ap.add_argument("--batch", type=int, default=512)
ap.add_argument("--reps", type=int, default=3)
args = ap.parse_args()
dev = torch.device("cuda:0")
spec = synth.SPECS["ViT-B/16"]
gen = torch.Generator(device=dev).manual_seed(1234)
sd = bench.device_clip_state(spec, gen, dev)
pl = bench.device_pl_state(spec, 2, gen, dev)

def make_engine(max_images):
    cm = modules.CLIPModel(sd, spec, str(dev))
    e = cm.engine(2)
    e.load_state_dict({}, pl)
    e._pl_loaded = True
    e.finalize(max_images, 64, 64)
    return e

img = torch.randn((args.images, 3, 224, 224), generator=gen, device=dev).half()

def run_single(e, B):
    out = torch.empty((args.images, spec.embed_dim), dtype=torch.float16, device=dev)
    for s in range(0, args.images, B):
        e.encode_image(img[s:s + B], normalize=True, out=out[s:s + B])
    return out

def run_dual(es, streams, B):
    out = torch.empty((args.images, spec.embed_dim), dtype=torch.float16, device=dev)
    cur = torch.cuda.current_stream()
    for st in streams:
        st.wait_stream(cur)
    for i, s in enumerate(range(0, args.images, B)):
        k = i % len(es)
        with torch.cuda.stream(streams[k]):
            es[k].encode_image(img[s:s + B], normalize=True, out=out[s:s + B])
    for st in streams:
        cur.wait_stream(st)
    return out

def timeit(fn):
    fn(); torch.cuda.synchronize()
    t = []
    for _ in range(args.reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); t.append(time.perf_counter() - t0)
    return args.images / min(t)

e1 = make_engine(args.batch)
res = {"single_b%d" % args.batch: round(timeit(lambda: run_single(e1, args.batch)), 1)}
ref = run_single(e1, args.batch)
for nstream, B in ((2, args.batch // 2), (2, args.batch), (3, args.batch // 2)):
    es = [make_engine(B) for _ in range(nstream)]
    streams = [torch.cuda.Stream() for _ in range(nstream)]
    res["%dstreams_b%d" % (nstream, B)] = round(timeit(lambda: run_dual(es, streams, B)), 1)
    got = run_dual(es, streams, B); torch.cuda.synchronize()
    res["%dstreams_b%d_maxdiff" % (nstream, B)] = float((got.float() - ref.float()).abs().max())
    del es
print(json.dumps(res))
