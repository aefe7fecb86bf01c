"""Eval-set loader contract of the hot path on tensors already resident in HBM.

The reference's DataManager.eval_set_loader (Dassl RandomClassSampler, n_ins = NUM_SHOTS) yields
dict batches {"img": [Cb*S,3,R,R], "label": [Cb*S] int64} with S consecutive rows per class.  Image
decoding is out of scope; this loader only reproduces that layout over a pre-built tensor.
"""
from __future__ import annotations

from typing import Iterator, Optional

import torch


class ResidentEvalSet:
    """images [n_cls*S,3,R,R] (row c*S+s belongs to class_ids[c]); yields `classes_per_batch` classes at a
    time.  `presharded=True` tells CustomCLIP.forward_prompt that this rank owns every batch."""

    def __init__(self, images: torch.Tensor, class_ids: torch.Tensor, shots: int, classes_per_batch: int,
                 presharded: bool = False):
        assert images.shape[0] == class_ids.shape[0] * shots
        self.images, self.class_ids, self.shots = images, class_ids.to(torch.int64), shots
        self.cpb = max(1, classes_per_batch)
        self.presharded = presharded
        self._labels = self.class_ids.repeat_interleave(shots)

    def __len__(self) -> int:
        return (self.class_ids.shape[0] + self.cpb - 1) // self.cpb

    def __iter__(self) -> Iterator[dict]:
        step = self.cpb * self.shots
        for s in range(0, self.images.shape[0], step):
            yield {"img": self.images[s:s + step], "label": self._labels[s:s + step]}
