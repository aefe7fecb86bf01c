"""N > 1 on the real engine (`pytest -m gpu`): classifier generation as 2 and 3 processes sharing the test box's one GPU
(process group gloo; on a multi-GPU node the same orchestration runs over RCCL) must reproduce the single-process HIP result
BIT FOR BIT -- classifier rows are computed independently per class, the argmax counters are integer sums."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
WORKER = os.path.join(os.path.dirname(os.path.abspath(__file__)), "dist_gpu_worker.py")


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _launch(world, result, presharded, C, backend="gloo"):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ, RANK=str(rank), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", OVMR_TEST_BACKEND=backend)
        procs.append(subprocess.Popen([sys.executable, WORKER, result, str(presharded), str(C)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    outs = [p.communicate(timeout=600)[0] for p in procs]
    for p, o in zip(procs, outs):
        assert p.returncode == 0, o[-3000:]
    return torch.load(result)


@pytest.mark.timeout(1500)
def test_multi_process_generation_bit_equal_to_single_process(tmp_path):
    C = 7                                                             # ragged over 2 and 3 ranks
    single = _launch(1, str(tmp_path / "w1.pt"), 0, C)
    assert single["files"] == ["mm_classifiers.pt", "visual_tokens.pt"]
    for world, presharded in ((2, 0), (2, 1), (3, 1)):
        got = _launch(world, str(tmp_path / f"w{world}_{presharded}.pt"), presharded, C)
        for k in ("mm", "v", "t", "tokens", "counts", "w", "out"):
            assert torch.equal(single[k], got[k]), f"world {world} presharded {presharded}: {k} differs"
        assert got["files"] == ["mm_classifiers.pt", "visual_tokens.pt"]          # rank 0 wrote the files


@pytest.mark.timeout(900)
def test_rccl_collectives_with_one_rank_bit_equal(tmp_path):
    """The `nccl` (= RCCL) branch on the hardware a test box has: ONE rank, process group "nccl" bound to cuda:0, the sharded
    path forced with CustomCLIP(distributed=True).  The packed all-gather and the counter all-reduce then run on DEVICE tensors
    through librccl (no host staging, shard._staged is the identity for this backend) and must leave rows, counters, fusion
    weights and outputs bit-equal to the run without a process group -- for the round-robin and for the class-sharded loader."""
    C = 7
    single = _launch(1, str(tmp_path / "w1.pt"), 0, C)
    assert single["backend"] == "none" and not single["sharded_path"]
    for presharded in (0, 1):
        got = _launch(1, str(tmp_path / f"rccl_{presharded}.pt"), presharded, C, backend="nccl")
        assert got["backend"] == "nccl" and got["sharded_path"] and got["rccl_loaded"]
        for k in ("mm", "v", "t", "tokens", "counts", "w", "out"):
            assert torch.equal(single[k], got[k]), f"nccl world 1, presharded {presharded}: {k} differs"
        assert got["files"] == ["mm_classifiers.pt", "visual_tokens.pt"]


@pytest.mark.skipif(torch.cuda.device_count() < 2, reason="needs at least 2 GPUs: one RCCL rank per device (armed for the 8-GPU node)")
@pytest.mark.timeout(1500)
def test_rccl_one_rank_per_device_bit_equal_to_single_process(tmp_path):
    """N > 1 over RCCL, one process per GPU (world = min(8, devices)): the class-sharded job -- packed all-gather and counter
    all-reduce on device tensors through librccl over xGMI -- must leave classifier rows, visual tokens, counters, fusion weights
    and fused outputs BIT-EQUAL to the one-process run, for the round-robin and for the class-sharded loader; every rank sits on a
    device of its own (bench.process_group_identity: devices_seen == world, checked by physical identity) and the RCCL version
    is on record.  Skipped on a one-GPU box (collected there, so the day more devices exist it runs by itself)."""
    world = min(8, torch.cuda.device_count())
    C = 2 * world + 3                                                 # ragged over the ranks
    single = _launch(1, str(tmp_path / "w1.pt"), 0, C)
    assert single["backend"] == "none" and not single["sharded_path"]
    for presharded in (0, 1):
        got = _launch(world, str(tmp_path / f"rccl_w{world}_{presharded}.pt"), presharded, C, backend="nccl")
        assert got["backend"] == "nccl" and got["sharded_path"] and got["rccl_loaded"]
        assert got["world"] == world and got["devices_seen"] == world
        assert got["rccl_version"] and got["rccl_version"][0].isdigit()
        for k in ("mm", "v", "t", "tokens", "counts", "w", "out"):
            assert torch.equal(single[k], got[k]), f"nccl world {world}, presharded {presharded}: {k} differs"
        assert got["files"] == ["mm_classifiers.pt", "visual_tokens.pt"]


@pytest.mark.timeout(900)
def test_bench_line_proves_its_process_group(tmp_path):
    """A SCALE record must say what it ran on: `bench.py --force-dist` (one rank, process group nccl = RCCL on cuda:0, the sharded
    path) carries `dist` = {backend, rccl_version, ranks, devices_seen, device_ids}, devices_seen equal to the rank count (bench.py
    raises on every rank otherwise); without a process group the field is null.  Small job: 12 classes x 4 shots + 32 queries."""
    import json
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    base = [sys.executable, os.path.join(repo, "bench.py"), "--classes", "12", "--shots", "4", "--queries", "32", "--query-batch", "16",
            "--steps", "1", "--warmup", "1", "--no-cpu-baseline", "--presets", "0"]
    lines = {}
    for tag, extra in (("dist", ["--force-dist", "--dist-timeout", "120"]), ("plain", [])):
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0")
        r = subprocess.run(base + extra, env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        lines[tag] = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"metric')][-1])
    d = lines["dist"]["dist"]
    assert lines["plain"]["dist"] is None
    assert d["backend"] == "nccl" and d["collective_library"] == "RCCL" and d["ranks"] == 1 and d["devices_seen"] == 1
    assert len(d["device_ids"]) == 1 and d["rccl_version"] and d["rccl_version"][0].isdigit()
    assert lines["dist"]["n_gpus"] == 1 and lines["dist"]["value"] > 0
    ct = d["collective_times_us"]                                                    # the two collectives at their real payloads, on this group
    assert ct["all_gather_rows"] > 0 and ct["all_reduce_counts"] > 0 and ct["payload_bytes"]["all_reduce_counts"] == 3 * 2 * 12 * 4


@pytest.mark.timeout(900)
def test_emulated_world_projection_runs_the_sharded_path(tmp_path):
    """`bench.py --emulate-world 2`: the whole job as one rank, then each rank's shard alone through the SHARDED code path (the two
    collectives served from the other rank's recorded contribution) -- a record with one entry per rank, shards that add up to the job,
    a projected speed-up between 1 and the rank count, labelled a projection.  Small job: 12 classes x 4 shots + 32 queries."""
    import json
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = [sys.executable, os.path.join(repo, "bench.py"), "--classes", "12", "--shots", "4", "--queries", "32", "--query-batch", "16",
           "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--emulate-world", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = json.loads([l for l in r.stdout.splitlines() if l.startswith('{"metric')][-1])
    assert d["projection"] is True and d["emulated_world"] == 2 and "PROJECTION" in d["metric"]
    per = d["per_rank"]
    assert [p["rank"] for p in per] == [0, 1]
    assert sum(p["classes"] for p in per) == 12 and sum(p["exemplar_images"] for p in per) == 48 and sum(p["query_images"] for p in per) == 32
    assert all(p["ms_per_step"] > 0 for p in per) and d["slowest_rank_ms"] == max(p["ms_per_step"] for p in per)
    assert 0.5 < d["projected_speedup"] <= 2.0 + 1e-6 or d["whole_job_ms_one_rank"] < 30.0      # (a toy job is launch-bound: no speed-up to expect)
    assert d["collective_payload_bytes"]["all_reduce_counts"] == 3 * 2 * 12 * 4
