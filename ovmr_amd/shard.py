"""Class sharding for multi-GPU classifier generation (one process per GPU, torch.distributed;
backend "nccl" is RCCL over xGMI on ROCm, "gloo" in the CPU tests).

The reference has no distributed path; the path shards naturally because classes are
independent until the cross-validation step:
  * every rank encodes the exemplars of its own classes and produces their mm / vision / text
    classifier rows and visual tokens;
  * ONE all-gather of the packed rows [n_local, 3*D + n_ctx*D] fp16 assembles the classifier matrix
    (3 MB in total at 1000 classes: latency bound, far below the ~153 GB/s of one xGMI link);
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


def all_gather_rows(rows: torch.Tensor, labels: torch.Tensor, dist) -> Tuple[torch.Tensor, torch.Tensor]:
    """All-gather a ragged set of rows.  rows [n_local, K], labels [n_local] int64 (class ids).
    Returns (all_rows [n_total, K], all_labels [n_total]) in rank order."""
    world = dist.get_world_size()
    n = torch.tensor([rows.shape[0]], dtype=torch.int64, device=rows.device)
    counts = [torch.zeros_like(n) for _ in range(world)]
    dist.all_gather(counts, n)
    nmax = max(int(c.item()) for c in counts)
    pad_rows = torch.zeros((nmax, rows.shape[1]), dtype=rows.dtype, device=rows.device)
    pad_labels = torch.full((nmax,), -1, dtype=torch.int64, device=rows.device)
    pad_rows[:rows.shape[0]] = rows
    pad_labels[:rows.shape[0]] = labels.to(torch.int64)
    out_rows = [torch.empty_like(pad_rows) for _ in range(world)]
    out_labels = [torch.empty_like(pad_labels) for _ in range(world)]
    dist.all_gather(out_rows, pad_rows)
    dist.all_gather(out_labels, pad_labels)
    all_rows, all_labels = torch.cat(out_rows), torch.cat(out_labels)
    keep = all_labels >= 0
    return all_rows[keep], all_labels[keep]
