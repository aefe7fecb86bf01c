import os
import sys

import numpy as np
import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if REPO not in sys.path:
    sys.path.insert(0, REPO)

GOLDEN = os.path.join(REPO, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def cosine_rows(a, b):
    """Row-wise cosine similarity of two [..., D] arrays, computed in float64."""
    a = np.asarray(a, dtype=np.float64).reshape(-1, np.shape(a)[-1])
    b = np.asarray(b, dtype=np.float64).reshape(-1, np.shape(b)[-1])
    return (a * b).sum(-1) / (np.linalg.norm(a, axis=-1) * np.linalg.norm(b, axis=-1) + 1e-300)


# The parity bar of BASELINE.json:north_star: "within 1e-3 fp16 cosine tolerance".
COS_TOL = 1e-3


def assert_cosine(a, b, tol=COS_TOL, what=""):
    c = cosine_rows(a, b)
    assert np.all(np.isfinite(c)), f"{what}: non-finite cosine"
    assert (1.0 - c).max() <= tol, f"{what}: 1-cos max {(1.0 - c).max():.3e} > {tol:.1e}"


@pytest.fixture(scope="session")
def golden():
    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"))
    return load


def near_tie_classes(logits, margin):
    """Classes whose cross-validation counters (tp, n_pred: trainers/mm_classifier_one_prompt.py:266-270) can differ
    between two correct fp16 implementations: for every row whose runner-up lies within `margin` of the maximum, all
    classes within `margin` of that maximum.  fusion_weight rows / fused-probability columns of every OTHER class must
    agree with the reference exactly."""
    logits = np.asarray(logits, dtype=np.float32)
    cand = logits >= logits.max(1, keepdims=True) - margin
    rows = cand.sum(1) > 1
    return set(np.nonzero(cand[rows].any(0))[0].tolist())


def aligned_case(g, name, tag="l2a", seed=11, n_ctx=None):
    """Inputs of an `l2a*` golden case (tests/golden/gen_golden.py:gen_l2_aligned), regenerated from the fixture's
    metadata: aligned fp32 state dicts, exemplar images / labels, query images.  n_ctx comes from the fixture (2 where an older
    file does not record it)."""
    from ovmr_amd import synth
    spec = synth.SPECS[name]
    if n_ctx is None:
        n_ctx = int(g[f"{tag}_meta_n_ctx"]) if f"{tag}_meta_n_ctx" in g.files else 2
    sd = synth.clip_state_dict(spec, seed, jitter=True)
    pl = synth.prompt_learner_state_dict(spec, n_ctx, seed, True)
    synth.align_state_dicts(sd, pl, spec, float(g[f"{tag}_meta_gain"]))
    labels, pattern = g[f"{tag}_eval_labels"], g[f"{tag}_eval_pattern_ids"]
    s, tile = float(g[f"{tag}_meta_strength"]), int(g[f"{tag}_meta_tile"])
    img = synth.images(len(labels), spec.image_resolution, seed=1234, class_ids=pattern, class_strength=s, tile=tile)
    qlab = g[f"{tag}_query_labels"]
    q = synth.images(len(qlab), spec.image_resolution, seed=777, class_ids=qlab, class_strength=s, tile=tile)
    return spec, sd, pl, labels, img, qlab, q


def usable_threads(cap=32):
    """Threads the CPU oracle may use without oversubscribing: the affinity mask capped by the cgroup CPU quota (the GPU boxes report 256
    hardware threads but grant 16 CPUs: a 32-thread pool there runs slower than a 16-thread one), at most `cap`."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q, per = f.read().split()
            if q != "max":
                n = min(n, max(1, int(float(q) / float(per))))
    except Exception:                                      # noqa: BLE001 -- cgroup v1 / no cgroup: the affinity mask stands
        pass
    return max(1, min(cap, n))


@pytest.fixture(scope="session", autouse=True)
def _oracle_thread_pool():
    """The CPU oracle's intra-op pool: torch's default is one thread per HARDWARE thread of the host (256 on the GPU boxes, whose cgroup
    grants 16 CPUs) -- an oversubscribed pool is several times slower.  Tests that want fewer set their own."""
    import torch
    torch.set_num_threads(usable_threads())
    yield
