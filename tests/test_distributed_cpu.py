"""N > 1 path on CPU: two gloo ranks run the REAL orchestration code (ovmr_amd.modules.CustomCLIP.forward_prompt:
batch sharding, all-gather of packed classifier rows, all-reduce of the F1 counters) on top of a test-only engine
that answers the Engine calls with the CPU oracle.  The result must equal the single-process result.
(The product Engine has no CPU path; this stand-in lives in tests/ only.)"""
import os
import socket
import sys
import tempfile

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO
from ovmr_amd import synth

SEED, N_CTX, S = 11, 2, 4


class OracleEngine:
    """Implements the subset of ovmr_amd.runtime.Engine that modules.py calls, with oracle/ovmr_oracle.py."""

    def __init__(self, spec, n_ctx):
        from oracle import ovmr_oracle as O
        self.O, self.spec, self.n_ctx = O, spec, n_ctx
        self.device = torch.device("cpu")
        self.finalized = False
        self.sd = O.convert_weights(O.to_torch(synth.clip_state_dict(spec, SEED, jitter=True)), "fp16")
        self.pl = {}
        self.logit_scale = float(self.sd["logit_scale"].exp())

    def set_weight(self, name, t):
        if name.startswith("prompt_learner."):
            self.pl[name[len("prompt_learner."):]] = t.float()

    def finalize(self, *a):
        self.finalized = True

    def embed_tokens(self, ids):
        return self.O.prompt_embeddings(ids, self.sd)

    def encode_text_ids(self, ids, seq_len=None, normalize=0):
        with torch.no_grad():
            t = self.O.encode_text(ids, self.sd)
        return self.O.l2_normalize(t) if normalize else t

    def encode_image(self, img, normalize=True, out=None):
        with torch.no_grad():
            f = self.O.encode_image(img.half(), self.sd)
        return self.O.l2_normalize(f) if normalize else f

    def generate_tokens(self, feats):
        O = self.O
        with torch.no_grad():
            cls = self.pl["cls_token"].unsqueeze(0).repeat(feats.shape[0], 1, 1)
            x = torch.cat([cls, feats.float()], dim=1)
            return O.transformer(x, self.pl, "aggregator.resblocks.", feats.shape[-1] // 64, None)[:, :self.n_ctx]

    def assemble_prompts(self, base, labels, tokens):
        src = base[labels.long()] if labels is not None else base[:1].repeat(tokens.shape[0], 1, 1)
        return self.O.update_prompts(src, tokens, self.n_ctx)

    def encode_text_embedded(self, prompts, index, seq_len=None, normalize=0):
        with torch.no_grad():
            t = self.O.text_encoder_forward(prompts, index, self.sd)
        for _ in range(normalize):
            t = self.O.l2_normalize(t)
        return t

    def encode_text_groups(self, groups):
        """One pass over several prompt families = the per-family calls (a sequence does not depend on its neighbours)."""
        return [self.encode_text_ids(g["ids"], g.get("seq_len"), g.get("normalize", 0)) if g.get("ids") is not None else
                self.encode_text_embedded(g["prompts"], g["index"], g.get("seq_len"), g.get("normalize", 0)) for g in groups]

    def pack_rows(self, mm, v, t, tokens, labels, bound):
        from ovmr_amd.shard import pack_block
        labels = labels.long()
        return pack_block(torch.cat([mm[labels], v[labels], t[labels], tokens[labels].flatten(1)], dim=1), labels, bound)

    def unpack_rows(self, gathered, C, D, n_ctx):
        K = (3 + n_ctx) * D
        labels = gathered[:, K:].contiguous().view(torch.int32).reshape(-1).long()
        ok = (labels >= 0) & (labels < C)
        full = torch.zeros((C, K), dtype=torch.float16)
        full[labels[ok]] = gathered[ok, :K]
        seen = torch.zeros(C + 1, dtype=torch.int32)
        seen.index_add_(0, torch.where(ok, labels, torch.full_like(labels, C)), ((labels != -1) | ok).int())
        return (full[:, :D].contiguous(), full[:, D:2 * D].contiguous(), full[:, 2 * D:3 * D].contiguous(),
                full[:, 3 * D:].reshape(C, n_ctx, D).contiguous(), seen)

    def xval_counts(self, feats, labels, clf, tp, n_pred):
        lg = (self.logit_scale * (feats @ clf.t())).float()
        pred = lg.argmax(1)
        n_pred += torch.bincount(pred, minlength=clf.shape[0]).int()
        tp += torch.bincount(labels.long()[pred == labels.long()], minlength=clf.shape[0]).int()

    def fusion_weights(self, counts, n_label, tau):
        f1 = torch.stack([self.O.f1_from_counts(counts[m, 0], counts[m, 1], n_label) for m in range(3)], -1)
        return (tau * f1).softmax(-1)

    def fused_logits(self, f, mm, v, t, w, mode):
        return self.O.inference_logits(f, mm, v, t, w, torch.tensor(self.logit_scale), mode)


class FakeCLIPModel:
    def __init__(self, spec):
        self.spec, self.dtype = spec, torch.float16
        self._e = {}
        self.logit_scale = torch.tensor(float(np.log(100.0)))

    def engine(self, n_ctx):
        if n_ctx not in self._e:
            self._e[n_ctx] = OracleEngine(self.spec, n_ctx)
        return self._e[n_ctx]


def _run(rank, world, port, outdir, result, C=6, presharded=False):
    sys.path.insert(0, REPO)
    torch.set_num_threads(2)
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)
    from ovmr_amd import modules
    spec = synth.SPECS["tiny"]
    cfg = modules.make_cfg(n_ctx=N_CTX, num_shots=S, output_dir=outdir if rank == 0 else "")
    pl = {k: torch.from_numpy(v) for k, v in synth.prompt_learner_state_dict(spec, N_CTX, SEED, True).items()}
    tok = torch.from_numpy(synth.class_token_ids(C, seed=4321))
    model = modules.CustomCLIP(cfg, tok, FakeCLIPModel(spec), prompt_learner_state=pl, reserve=(8, 8, 8))
    labels = np.repeat(np.random.default_rng(2).permutation(C), S)
    img = torch.from_numpy(synth.images(C * S, spec.image_resolution, 1234, labels, 0.6))
    if presharded:                                                  # class-sharded loader: only this rank's classes exist here
        from ovmr_amd.data import ResidentEvalSet
        from ovmr_amd.shard import shard_range
        a, b = shard_range(C, rank, world)
        mine = np.concatenate([np.nonzero(labels == c)[0] for c in range(a, b)]) if b > a else np.zeros(0, dtype=np.int64)
        loader = ResidentEvalSet(img[mine], torch.arange(a, b), S, 2, presharded=world > 1)
    else:                                                           # one class per batch, batch i -> rank i % world
        loader = [{"img": img[s:s + S], "label": torch.from_numpy(labels[s:s + S])} for s in range(0, C * S, S)]
    q = torch.from_numpy(synth.images(5, spec.image_resolution, 777))
    out = model(q, eval_set_loader=loader)
    w_gen = model.fusion_weight.clone()
    # coop_mm_classifier.get_fusion_weight variant: same exemplars, externally supplied classifiers, tau fixed at 10
    w_coop = model.get_fusion_weight(loader, model.mm_classifier.float(), model.visual_classifer, model.zero_shot_classifier)
    assert torch.equal(w_coop, w_gen)
    if rank == 0:
        from oracle import ovmr_oracle as O
        ref = O.get_fusion_weight_coop(model.eval_feat4cls if world == 1 else None, model.mm_classifier,
                                       model.visual_classifer, model.zero_shot_classifier,
                                       torch.tensor(model.engine.logit_scale)) if world == 1 else w_coop
        assert torch.allclose(ref, w_coop, atol=1e-6)
        torch.save({"out": out, "mm": model.mm_classifier, "v": model.visual_classifer, "t": model.zero_shot_classifier,
                    "w": model.fusion_weight, "counts": model.xval_counts, "tokens": model.visual_tokens}, result)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,C,presharded", [(2, 6, False), (4, 7, False), (4, 7, True), (4, 3, True), (4, 3, False)])
def test_multi_rank_generation_matches_single_process(world, C, presharded):
    """2 ranks x 6 classes; 4 ranks x 7 classes (ragged: shards of 2, 2, 2, 1), round-robin batches and class-sharded loader;
    4 ranks x 3 classes: a rank that owns NO class still takes part in both collectives (an empty block, zero votes)."""
    with tempfile.TemporaryDirectory() as d:
        r1, r2 = os.path.join(d, "single.pt"), os.path.join(d, "dist.pt")
        _run(0, 1, 0, os.path.join(d, "o1"), r1, C, False)
        mp.spawn(_run, args=(world, _free_port(), os.path.join(d, "o2"), r2, C, presharded), nprocs=world, join=True)
        a, b = torch.load(r1), torch.load(r2)
        for k in ("mm", "v", "t", "tokens"):
            assert torch.equal(a[k], b[k]), k                      # rows are computed independently per class
        assert torch.equal(a["counts"], b["counts"])               # int32 counters summed over ranks
        assert torch.allclose(a["w"], b["w"], atol=0)
        assert torch.allclose(a["out"], b["out"], atol=1e-6)
        saved = torch.load(os.path.join(d, "o2", "mm_classifiers.pt"))
        assert sorted(saved) == ["fusion_weight", "mm_classifier", "text_classifier", "vision_classifier"]


def _gather(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, REPO)
    from ovmr_amd.shard import all_gather_rows, all_reduce_counts, local_class_bound, shard_range
    C = 70001                                                       # labels beyond fp16's exact-integer range
    full = torch.arange(7 * 3, dtype=torch.float16).reshape(7, 3)
    ids = torch.tensor([0, 5, 2047, 2049, 40000, 65537, 70000])
    a, b = shard_range(7, rank, world)                              # ragged: world 4 -> 2 + 2 + 2 + 1 rows
    bound = local_class_bound(7, world, True, 1)
    rows, labels = all_gather_rows(full[a:b], ids[a:b], bound, dist)
    ok = rows.shape == (world * bound, 3) and labels.dtype == torch.int32
    keep = labels >= 0
    ok = ok and int(keep.sum()) == 7 and torch.equal(labels[keep].long(), ids) and torch.equal(rows[keep], full)
    ok = ok and bool((rows[~keep] == 0).all())
    counts = torch.full((3, 2, 5), rank + 1, dtype=torch.int32)
    ok = ok and torch.equal(all_reduce_counts(counts, dist), torch.full((3, 2, 5), world * (world + 1) // 2, dtype=torch.int32))
    try:
        all_gather_rows(full, torch.arange(7), 3, dist)
        ok = False
    except RuntimeError as e:
        ok = ok and "more than the bound" in str(e)
    if rank == 0:
        q.put(ok)
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(300)
@pytest.mark.parametrize("world", [2, 4])
def test_ragged_all_gather_rows(world):
    ctx = mp.get_context("spawn")
    q = ctx.SimpleQueue()
    mp.spawn(_gather, args=(world, _free_port(), q), nprocs=world, join=True)
    assert q.get() is True


def test_local_class_bound_covers_round_robin_batches():
    """The bound every rank computes without communication is never below what a rank can receive."""
    from ovmr_amd.shard import local_class_bound, shard_batches, shard_range
    for C in (1, 6, 7, 100, 1000, 1003):
        for world in (1, 2, 3, 4, 8):
            assert local_class_bound(C, world, True, 16) == max(shard_range(C, r, world)[1] - shard_range(C, r, world)[0] for r in range(world))
            for cfg_cpb in (1, 4, 16):
                for cpb in range(1, cfg_cpb + 1):                   # loaders may yield smaller batches than TEST.BATCH_SIZE // S
                    nb = -(-C // cpb)
                    sizes = [min(cpb, C - i * cpb) for i in range(nb)]
                    worst = max(sum(sizes[i] for i in shard_batches(nb, r, world)) for r in range(world))
                    assert worst <= local_class_bound(C, world, False, cfg_cpb), (C, world, cfg_cpb, cpb)


def test_emulated_peers_reproduce_the_sharded_job():
    """bench.py --emulate-world rests on `bench.EmulatedPeers`: rank r of N runs the SHARDED code path of forward_prompt alone, the two
    collectives answered from the other ranks' recorded contributions.  On the CPU stand-in engine: for every rank of a 3-rank job
    (ragged: 7 classes = 3 + 2 + 2) the emulated run must end with the whole job's classifier rows, tokens, counters and fusion weights --
    the same result the real multi-process run above ends with."""
    sys.path.insert(0, REPO)
    import bench
    from ovmr_amd import modules
    from ovmr_amd.data import ResidentEvalSet
    from ovmr_amd.shard import local_class_bound, pack_block, shard_range
    torch.set_num_threads(2)
    spec, C, world = synth.SPECS["tiny"], 7, 3
    cfg = modules.make_cfg(n_ctx=N_CTX, num_shots=S, output_dir="")
    pl = {k: torch.from_numpy(v) for k, v in synth.prompt_learner_state_dict(spec, N_CTX, SEED, True).items()}
    tok = torch.from_numpy(synth.class_token_ids(C, seed=4321))
    model = modules.CustomCLIP(cfg, tok, FakeCLIPModel(spec), prompt_learner_state=pl, reserve=(8, 8, 8), stream_text=True)
    labels = np.repeat(np.arange(C), S)
    img = torch.from_numpy(synth.images(C * S, spec.image_resolution, 1234, labels, 0.6))
    # (one class per loader batch everywhere: torch's CPU GEMMs are not batch-invariant, the HIP kernels' rows are)
    model.forward_prompt(ResidentEvalSet(img, torch.arange(C), S, 1, presharded=True))          # the whole job, one process
    ref = {k: getattr(model, k).clone() for k in ("mm_classifier", "visual_classifer", "zero_shot_classifier", "fusion_weight", "visual_tokens")}
    counts_full = model.xval_counts.clone()
    bound = local_class_bound(C, world, True, 1)
    blocks = []
    for r in range(world):
        a, b = shard_range(C, r, world)
        loc = torch.arange(a, b)
        blocks.append(pack_block(torch.cat([ref["mm_classifier"][loc], ref["visual_classifer"][loc], ref["zero_shot_classifier"][loc],
                                            ref["visual_tokens"][loc].flatten(1)], dim=1), loc, bound))
    peer_blocks = torch.cat(blocks)
    for r in range(world):
        a, b = shard_range(C, r, world)
        emu = bench.EmulatedPeers(r, world)
        emu.peer_blocks = peer_blocks
        model._dist, model._text_streamed = emu, True
        loader = ResidentEvalSet(img[a * S:b * S], torch.arange(a, b), S, 1, presharded=True)
        model.forward_prompt(loader)                                  # records this rank's own votes
        assert int(emu.local_counts[:, 1].sum()) == 3 * (b - a) * S   # every one of its rows voted once per classifier
        emu.peer_counts = counts_full - emu.local_counts
        model.forward_prompt(loader)
        for k in ref:
            assert torch.equal(getattr(model, k), ref[k]), f"rank {r}: {k}"
        assert torch.equal(model.xval_counts, counts_full), f"rank {r}: counters"
    model._dist = None
