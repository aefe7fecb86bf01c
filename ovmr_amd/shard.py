"""Class sharding for multi-GPU classifier generation (one process per GPU, torch.distributed;
backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests).

The reference has no distributed path; the path shards naturally because classes are
independent until the cross-validation step:
  * every rank encodes the exemplars of its own classes and produces their mm / vision / text
    classifier rows and visual tokens;
  * ONE all-gather of the packed rows [bound, 3*D + n_ctx*D + 2] fp16 (mm | vision | text | visual tokens | label bits)
    assembles the classifier matrix (5 MB in total at 1000 classes: latency bound, far below the ~153 GB/s of one xGMI
    link); the per-rank block size `bound` follows from C, the world size and the loader contract, so no size exchange
    precedes it;
  * every rank runs the argmax counting for its own exemplar rows against all classes and ONE
    all-reduce sums the int32 [3,2,C] counters (n_pred[c] receives votes from other ranks' rows).
"""
from __future__ import annotations

from typing import List, Tuple

import torch


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced split of n units: ranks < n % world get one extra unit."""
    base, rem = divmod(n, world)
    start = rank * base + min(rank, rem)
    return start, start + base + (1 if rank < rem else 0)


def shard_batches(num_batches: int, rank: int, world: int) -> List[int]:
    """Round-robin assignment used by CustomCLIP.forward_prompt (batch i belongs to rank i % world)."""
    return [i for i in range(num_batches) if i % world == rank]


def local_class_bound(num_classes: int, world: int, presharded: bool, classes_per_batch: int) -> int:
    """Upper bound of the classes ONE rank produces, computed from values every rank knows (no communication).
    presharded loaders own a contiguous `shard_range` of the classes; otherwise batch i belongs to rank i % world and a
    batch holds at most `classes_per_batch` = TEST.BATCH_SIZE // NUM_SHOTS classes, which bounds a rank's
    share by C / world + 2 * classes_per_batch for any batch size up to that."""
    if presharded:
        return -(-num_classes // world)
    return -(-num_classes // world) + 2 * max(1, classes_per_batch)


def _staged(t: torch.Tensor, dist) -> torch.Tensor:
    """gloo moves host memory: stage device tensors through the CPU (CPU tests, or several ranks sharing one GPU)."""
    return t.cpu() if dist.get_backend() == "gloo" and t.is_cuda else t


def pack_block(rows: torch.Tensor, labels: torch.Tensor, bound: int) -> torch.Tensor:
    """One rank's contribution to the all-gather: [bound, K + 2] fp16 = its rows, the int32 label of each row carried in two
    extra fp16 columns (bit pattern, not value), padding rows labelled -1."""
    n, K = rows.shape
    if n > bound:
        raise RuntimeError(f"this rank produced {n} classes, more than the bound {bound} every rank agreed on: the eval-set "
                           "loader yields more than TEST.BATCH_SIZE // NUM_SHOTS classes per batch")
    assert rows.dtype == torch.float16
    lab = torch.full((bound,), -1, dtype=torch.int32, device=rows.device)
    lab[:n] = labels.to(torch.int32)
    block = torch.zeros((bound, K + 2), dtype=torch.float16, device=rows.device)
    block[:n, :K] = rows
    block[:, K:] = lab.view(torch.float16).reshape(bound, 2)
    return block


def all_gather_rows(rows: torch.Tensor, labels: torch.Tensor, bound: int, dist) -> Tuple[torch.Tensor, torch.Tensor]:
    """ONE all-gather of a ragged set of fp16 rows.  rows [n_local, K] fp16, labels [n_local] (class ids), n_local <= bound
    (`local_class_bound`, identical on every rank).  Every rank contributes one `pack_block`.
    Returns (all_rows [world * bound, K], all_labels [world * bound] int32 with -1 on padding rows), rank-major."""
    K = rows.shape[1]
    world = dist.get_world_size()
    block = _staged(pack_block(rows, labels, bound), dist)
    out = torch.empty((world * bound, K + 2), dtype=torch.float16, device=block.device)
    dist.all_gather_into_tensor(out, block)
    out = out.to(rows.device)
    return out[:, :K], out[:, K:].contiguous().view(torch.int32).reshape(-1)


def all_gather_block(block: torch.Tensor, dist) -> torch.Tensor:
    """ONE all-gather of every rank's `pack_block`-format block [bound, K + 2] fp16 (Engine.pack_rows builds it in one launch) ->
    [world * bound, K + 2], rank-major, on the block's device."""
    b = _staged(block, dist)
    out = torch.empty((dist.get_world_size() * b.shape[0], b.shape[1]), dtype=torch.float16, device=b.device)
    dist.all_gather_into_tensor(out, b)
    return out.to(block.device)


def all_reduce_counts(counts: torch.Tensor, dist) -> torch.Tensor:
    """ONE all-reduce (sum) of the int32 [3, 2, C] argmax counters."""
    c = _staged(counts, dist)
    dist.all_reduce(c)
    return c.to(counts.device)
