#!/usr/bin/env python3
"""How the CPU oracle scales over worker processes on this host (no GPU needed): bench.cpu_baseline at several process counts.
    python tools/cpu_baseline_probe.py [--procs 1 4 16] [--model ViT-B/16]"""
import argparse
import json
import os
import sys
import time
from types import SimpleNamespace

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from ovmr_amd import synth

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--procs", type=int, nargs="+", default=[1, 4, 16])
    ap.add_argument("--threads", type=int, default=16)
    ap.add_argument("--model", default="ViT-B/16")
    ap.add_argument("--reps", type=int, default=1)
    a = ap.parse_args()
    print("cpu_count", os.cpu_count(), "effective", bench.effective_cpus(), "loadavg", os.getloadavg(), flush=True)
    for f in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us", "/sys/fs/cgroup/cpuset.cpus.effective"):
        if os.path.exists(f):
            print(f, open(f).read().strip(), flush=True)
    spec = synth.SPECS[a.model]
    gen = torch.Generator().manual_seed(1)
    sd = bench.device_clip_state(spec, gen, "cpu")
    pl = bench.device_pl_state(spec, 2, gen, "cpu")
    tok = torch.from_numpy(synth.class_token_ids(256, seed=4321))
    for n in a.procs:
        args = SimpleNamespace(shots=16, cpu_sample_classes=2, cpu_threads=a.threads, cpu_reps=a.reps, cpu_timeout=200.0, cpu_procs=n)
        t = time.time()
        r = bench.cpu_baseline(spec, sd, pl, tok, args, 2)
        print(n, "procs:", json.dumps({k: r.get(k) for k in ("value", "fp32_images_per_s", "fp16_images_per_s", "per_process_images_per_s", "sample")}), f"{time.time() - t:.0f}s", flush=True)
