"""bench.py's host logic that runs without a GPU: the presets, the `--gpus N` preflight (environment + device count, before anything
touches the GPU), the CPU-baseline statistics."""
import os
import subprocess
import sys

import pytest

from conftest import REPO

BENCH = os.path.join(REPO, "bench.py")


def test_presets_cover_every_baseline_configuration():
    import json
    sys.path.insert(0, REPO)
    import bench
    configs = json.load(open(os.path.join(REPO, "BASELINE.json")))["configs"]
    assert len(configs) == 5 and {"metric", "c1", "c2", "c3", "c4", "c5"} == set(bench.PRESETS)
    a = bench.parse(["--preset", "c1"])
    assert (a.model, a.classes, a.query_batch) == ("ViT-B/16", 10, 256) and bench.PRESETS["c1"]["value"] == "zeroshot"
    assert bench.parse(["--preset", "c1", "--queries", "512"]).queries == 512               # explicit flags win over a preset
    a = bench.parse([])
    assert (a.classes, a.shots, a.queries, a.cpu_sample_classes, a.cpu_reps) == (1000, 16, 4096, 4, 5)
    ids = bench.zeroshot_prompt_ids(10)
    assert ids.shape == (10, 77) and (ids[:, 0] == 49406).all() and (ids[:, 1:5] == [320, 1125, 539, 320]).all()
    eot = ids.argmax(1)
    assert ((eot >= 7) & (eot <= 9)).all() and all(ids[i, eot[i] - 1] == 269 and ids[i, eot[i] + 1:].sum() == 0 for i in range(10))


def _run(args, **env):
    e = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "OVMR_DIST_BACKEND")}
    e.update(env)
    return subprocess.run([sys.executable, BENCH] + args, env=e, capture_output=True, text=True, timeout=300)


def test_multi_gpu_preflight_fails_in_one_line_before_touching_the_gpu():
    """`bench.py --gpus N` (what the driver runs for the scaling record) on a node with fewer than N devices: ONE line on stderr, a non-zero
    exit code, nothing launched -- as a launcher (no WORLD_SIZE) and as a rank under torch.distributed.run (WORLD_SIZE set)."""
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip("this node has the devices the preflight asks for")
    r = _run(["--gpus", "2", "--steps", "1", "--warmup", "0"])
    assert r.returncode == 2 and r.stdout == ""
    lines = [l for l in r.stderr.splitlines() if l.strip()]
    assert len(lines) == 1 and "needs 2 visible devices" in lines[0] and "OVMR_DIST_BACKEND=gloo" in lines[0]
    r = _run(["--gpus", "2"], WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    assert r.returncode == 2 and "needs 2 visible devices" in r.stderr and len([l for l in r.stderr.splitlines() if l.strip()]) == 1
    r = _run(["--gpus", "2"], WORLD_SIZE="2", RANK="1", LOCAL_RANK="1")                   # only rank 0 speaks
    assert r.returncode == 2 and r.stderr.strip() == ""


def test_preflight_sets_the_environment_of_the_multi_process_gpu_tests(monkeypatch):
    """Every multi-process GPU test of this repository runs with HSA_ENABLE_IPC_MODE_LEGACY=0 (dmabuf IPC: the only mode the pool's host
    driver supports -- without it RCCL fails with `hipIpcGetMemHandle: invalid argument`); bench.py sets the same before any GPU call and
    keeps an operator's explicit value."""
    sys.path.insert(0, REPO)
    import bench
    args = bench.parse(["--gpus", "1"])
    monkeypatch.delenv("HSA_ENABLE_IPC_MODE_LEGACY", raising=False)
    monkeypatch.delenv("WORLD_SIZE", raising=False)
    bench.preflight(args)
    assert os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "0"
    monkeypatch.setenv("HSA_ENABLE_IPC_MODE_LEGACY", "1")
    bench.preflight(args)
    assert os.environ["HSA_ENABLE_IPC_MODE_LEGACY"] == "1"
    for f in ("tests/test_hip_distributed.py", "tests/test_hip_parity.py", "scripts/generate_classifier.sh"):
        assert "HSA_ENABLE_IPC_MODE_LEGACY" in open(os.path.join(REPO, f)).read(), f
