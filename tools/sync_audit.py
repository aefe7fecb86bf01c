#!/usr/bin/env python3
"""Host synchronisation points of one sharded generation + query pass (an emulated rank of eight, bench.py's own code path), found with
torch.cuda.set_sync_debug_mode("warn"): every ATen call that makes the host wait for the stream (a scalar copied to the device from
pageable memory, .item(), bool(tensor), a synchronous copy) is reported with the Python line that issued it.  A sync in front of
enqueued work is a bubble: the host stops running ahead of the GPU, and every launch behind it pays the host's latency.

    python tools/sync_audit.py [--rank 1] [--world 8]
"""
import argparse
import os
import sys
import traceback
import warnings

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rank", type=int, default=1)
    ap.add_argument("--world", type=int, default=8)
    ap.add_argument("--classes", type=int, default=1000)
    a = ap.parse_args()
    import torch
    import bench
    from ovmr_amd.data import ResidentEvalSet
    from ovmr_amd.shard import shard_range, local_class_bound
    args = bench.parse(["--no-cpu-baseline", "--presets", "0", "--classes", str(a.classes)])
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    spec, sd, pl, tok, model = bench.make_model(args, dev, output_dir="")
    C, S, Q, R, D = args.classes, args.shots, args.queries, spec.image_resolution, spec.embed_dim
    c0, c1 = shard_range(C, a.rank, a.world)
    q0, q1 = shard_range(Q, a.rank, a.world)
    g = torch.Generator(device=dev).manual_seed(1)
    ex = torch.randn(((c1 - c0) * S, 3, R, R), generator=g, device=dev).half()
    q = torch.randn((q1 - q0, 3, R, R), generator=g, device=dev).half()
    loader = ResidentEvalSet(ex, torch.arange(c0, c1, device=dev), S, args.classes_per_batch, presharded=True)
    emu = bench.EmulatedPeers(a.rank, a.world)
    bound = local_class_bound(C, a.world, True, 1)
    emu.peer_blocks = torch.zeros((a.world * bound, 3 * D + 2 * D + 2), dtype=torch.float16, device=dev)
    lab = torch.full((a.world * bound,), -1, dtype=torch.int32, device=dev)
    for r in range(a.world):                                   # the peers' rows: zeros under their own labels (every class seen once)
        r0, r1 = shard_range(C, r, a.world)
        lab[r * bound:r * bound + (r1 - r0)] = torch.arange(r0, r1, dtype=torch.int32, device=dev)
    emu.peer_blocks[:, -2:] = lab.view(torch.float16).reshape(-1, 2)
    model._dist, model._text_streamed = emu, True
    model._twin()

    def step():
        model.forward_prompt(loader, wait_files=False)
        for _ in model.forward_batches((q[b:b + args.query_batch] for b in range(0, q.shape[0], args.query_batch)), stable_inputs=True):
            pass

    step()
    emu.peer_counts = torch.zeros_like(emu.local_counts)
    step()
    torch.cuda.synchronize()
    seen = {}
    torch.cuda.set_sync_debug_mode("warn")
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        orig = warnings.showwarning

        def show(message, category, filename, lineno, file=None, line=None):
            stack = [f for f in traceback.extract_stack() if "/ovmr_amd/" in f.filename or f.filename.endswith("bench.py")]
            where = " <- ".join(f"{os.path.basename(f.filename)}:{f.lineno} {f.line}" for f in reversed(stack[-2:]))
            seen[where] = seen.get(where, 0) + 1
        warnings.showwarning = show
        try:
            step()
        finally:
            warnings.showwarning = orig
    torch.cuda.set_sync_debug_mode("default")
    print(f"host synchronisation points in one step of rank {a.rank} of {a.world} ({c1 - c0} classes, {q1 - q0} queries): {sum(seen.values())}")
    for k, v in seen.items():
        print(f"  {v} x  {k}")


if __name__ == "__main__":
    main()
