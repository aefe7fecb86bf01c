#!/usr/bin/env python3
"""How long the HOST takes to enqueue one encoder launch sequence (ovmr_encode_image on a 256-image batch: ~75 kernel launches from one
ctypes call) against how long the GPU takes to run it -- and the same sequence replayed from a HIP graph.  A test loop's batch cannot start
on the GPU before the host has enqueued it: with two batches in flight (forward_batches) the second batch starts one enqueue time late."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from ovmr_amd import modules, synth

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
spec = synth.SPECS["ViT-B/16"]
gen = torch.Generator(device=dev).manual_seed(1)
sd = bench.device_clip_state(spec, gen, dev)
pl = bench.device_pl_state(spec, 2, gen, dev)
cm = modules.CLIPModel(sd, spec, str(dev))
e = cm.engine(2)
e.load_state_dict({}, pl)
e._pl_loaded = True
e.finalize(256, 64, 1024)
for B in (64, 256):
    img = torch.randn((B, 3, 224, 224), generator=gen, device=dev).half()
    out = torch.empty((B, spec.embed_dim), dtype=torch.float16, device=dev)
    for _ in range(3):
        e.encode_image(img, out=out)
    torch.cuda.synchronize()
    host, total = [], []
    for _ in range(10):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e.encode_image(img, out=out)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        host.append((t1 - t0) * 1e6); total.append((t2 - t0) * 1e6)
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        e.encode_image(img, out=out)
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=st):
            e.encode_image(img, out=out)
    torch.cuda.synchronize()
    ghost, gtotal = [], []
    for _ in range(10):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        g.replay()
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        ghost.append((t1 - t0) * 1e6); gtotal.append((t2 - t0) * 1e6)
    med = lambda v: round(sorted(v)[len(v) // 2], 1)
    print(json.dumps({"images": B, "host_enqueue_us": med(host), "enqueue_to_done_us": med(total), "graph_replay_host_us": med(ghost),
                      "graph_replay_to_done_us": med(gtotal)}), flush=True)
