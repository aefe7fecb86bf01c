#!/usr/bin/env python3
"""Does a second engine on a second stream fill the partial last round of tiles of batch-256 inference (N = 768 launches: 2.31 rounds)?
Encodes 16 x 256 resident images with ONE engine on one stream, then with TWO engines (own workspaces, same weights) alternating on two streams.

    python tools/two_stream_probe.py [--batch 256] [--n 16]
"""
import argparse, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from ovmr_amd import synth
from ovmr_amd.runtime import Engine

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=256)
ap.add_argument("--n", type=int, default=16)
ap.add_argument("--reps", type=int, default=5)
ap.add_argument("--model", default="ViT-B/16")
ap.add_argument("--streams", type=int, default=2, help="engines / streams for the multi-stream leg")
args = ap.parse_args()
dev = torch.device("cuda:0")
spec = synth.SPECS[args.model]
gen = torch.Generator(device=dev).manual_seed(1234)
sd = bench.device_clip_state(spec, gen, dev)
pl = bench.device_pl_state(spec, 2, gen, dev)
engines = []
for _ in range(args.streams):
    e = Engine(spec, 2, "cuda:0")
    e.load_state_dict(sd, pl)
    e.finalize(args.batch, 256, 1024)
    engines.append(e)
R = spec.image_resolution
img = torch.randn((args.n * args.batch, 3, R, R), device=dev).half()
outs = [torch.empty((args.batch, spec.embed_dim), dtype=torch.float16, device=dev) for _ in range(args.streams)]
streams = [torch.cuda.Stream() for _ in range(args.streams)]

def one():
    for b in range(args.n):
        engines[0].encode_image(img[b * args.batch:(b + 1) * args.batch], out=outs[0])

def two():
    cur = torch.cuda.current_stream()
    for s in streams:
        s.wait_stream(cur)
    for b in range(args.n):
        k = b % args.streams
        with torch.cuda.stream(streams[k]):
            engines[k].encode_image(img[b * args.batch:(b + 1) * args.batch], out=outs[k])
    for s in streams:
        cur.wait_stream(s)

res = {}
for name, fn in (("one_stream", one), ("two_streams", two), ("one_stream_again", one)):
    fn(); torch.cuda.synchronize()
    ts = []
    for _ in range(args.reps):
        t = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t)
    res[name] = {"ms": round(min(ts) * 1e3, 2), "images_per_s": round(args.n * args.batch / min(ts), 1)}
# same results from both engines
engines[0].encode_image(img[:args.batch], out=outs[0]); engines[1].encode_image(img[:args.batch], out=outs[1]); torch.cuda.synchronize()
res["bit_equal_engines"] = bool(torch.equal(outs[0], outs[1]))
res["model"], res["batch"], res["streams"] = args.model, args.batch, args.streams
print(json.dumps(res))
