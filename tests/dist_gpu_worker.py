"""Worker of tests/test_hip_distributed.py: one rank of a multi-process classifier-generation job on the REAL engine.
Several ranks share the one GPU of the test box, so the process group uses gloo (ovmr_amd.shard stages the two collectives
through host memory for that backend); on an 8-GPU node the same code runs with backend nccl = RCCL, one GPU per rank.

With OVMR_TEST_BACKEND=nccl (one rank per GPU: a single rank on the one-GPU test box) the process group is RCCL, the sharded path
is forced with CustomCLIP(distributed=True) and the two collectives run on device tensors through librccl.

    RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT in the environment;  argv: <result path> <presharded 0|1> <classes>"""
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ovmr_amd import modules, synth  # noqa: E402
from ovmr_amd.data import ResidentEvalSet  # noqa: E402
from ovmr_amd.shard import shard_range  # noqa: E402

SEED, S = 11, 4


def main():
    result, presharded, C = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    backend = os.environ.get("OVMR_TEST_BACKEND", "gloo")
    force = backend == "nccl"
    if backend == "nccl":
        torch.cuda.set_device(rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(f"cuda:{rank}"))
    elif world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    spec = synth.SPECS["small"]
    sd = {k: torch.from_numpy(v) for k, v in synth.clip_state_dict(spec, SEED, jitter=True).items()}
    pl = {k: torch.from_numpy(v) for k, v in synth.prompt_learner_state_dict(spec, 2, SEED, True).items()}
    cm = modules.CLIPModel(sd, spec, f"cuda:{rank}" if force else "cuda:0")
    out_dir = os.path.join(os.path.dirname(result), f"out_w{world}")
    cfg = modules.make_cfg(n_ctx=2, num_shots=S, output_dir=out_dir, test_batch_size=3 * S)
    tok = torch.from_numpy(synth.class_token_ids(C, seed=4321))
    model = modules.CustomCLIP(cfg, tok, cm, prompt_learner_state=pl, reserve=(64, 64, 64), distributed=True if force else None)
    labels = np.repeat(np.random.default_rng(2).permutation(C), S)
    img = torch.from_numpy(synth.images(C * S, spec.image_resolution, 1234, labels, 0.6))
    if presharded and (world > 1 or force):
        a, b = shard_range(C, rank, world)
        mine = np.concatenate([np.nonzero(labels == c)[0] for c in range(a, b)]) if b > a else np.zeros(0, dtype=np.int64)
        loader = ResidentEvalSet(img[mine].cuda().half(), torch.arange(a, b), S, 3, presharded=True)
    else:
        loader = [{"img": img[s:s + 3 * S], "label": torch.from_numpy(labels[s:s + 3 * S])} for s in range(0, C * S, 3 * S)]
    q = torch.from_numpy(synth.images(5, spec.image_resolution, 777))
    out = model(q, eval_set_loader=loader)
    model.wait_files()                        # (the first forward generated the classifiers: rank 0's files are written behind its back)
    torch.cuda.synchronize()
    devices_seen, rccl_version = None, None
    if dist.is_initialized():
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        info = bench.process_group_identity(dist, model.device, backend, world)      # raises on every rank when RCCL ranks share a device
        devices_seen, rccl_version = info["devices_seen"], info["rccl_version"]
    if rank == 0:
        torch.save({"devices_seen": devices_seen, "rccl_version": rccl_version, "world": world,
                    "out": out.cpu(), "mm": model.mm_classifier.cpu(), "v": model.visual_classifer.cpu(),
                    "t": model.zero_shot_classifier.cpu(), "w": model.fusion_weight.cpu(), "counts": model.xval_counts.cpu(),
                    "tokens": model.visual_tokens.cpu(), "files": sorted(os.listdir(out_dir)),
                    "backend": dist.get_backend() if dist.is_initialized() else "none",
                    "sharded_path": model._dist is not None,
                    "rccl_loaded": any("librccl" in line for line in open("/proc/self/maps"))}, result)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
