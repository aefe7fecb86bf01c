"""On-device classification evaluator.

Arithmetic of Dassl's `Classification` evaluator (Dassl.pytorch/dassl/evaluation/evaluator.py:50-138): top-1 accuracy,
error rate, macro-F1 and per-class accuracy / F1, but accumulated as three int64 histograms on the GPU -- the reference
does `.item()` and `.cpu().numpy()` per batch (:59-67), i.e. one host sync per batch of 256 images.
"""
from __future__ import annotations

import os
from collections import OrderedDict
from typing import Optional, Sequence

import torch


class Classification:
    def __init__(self, num_classes: int, classnames: Optional[Sequence[str]] = None, device="cuda"):
        self.num_classes = num_classes
        self.classnames = list(classnames) if classnames is not None else [str(i) for i in range(num_classes)]
        self.device = torch.device(device)
        self.reset()

    def reset(self):
        z = lambda: torch.zeros(self.num_classes, dtype=torch.int64, device=self.device)
        self._tp, self._n_pred, self._n_label = z(), z(), z()

    @torch.no_grad()
    def process(self, mo: torch.Tensor, gt: torch.Tensor):
        """mo: [B, C] model output, gt: [B] labels (evaluator.py:50-67).  No host synchronisation."""
        pred = mo.argmax(dim=1)                                          # mo.max(1)[1]
        gt = gt.to(pred.device).long()
        C = self.num_classes
        self._n_label += torch.bincount(gt, minlength=C)
        self._n_pred += torch.bincount(pred, minlength=C)
        self._tp += torch.bincount(gt[pred == gt], minlength=C)

    def evaluate(self, output_dir: Optional[str] = None) -> "OrderedDict[str, float]":
        tp, n_pred, n_label = (t.double().cpu() for t in (self._tp, self._n_pred, self._n_label))
        total = float(n_label.sum())
        acc = 100.0 * float(tp.sum()) / max(total, 1.0)
        precision = torch.where(n_pred > 0, tp / n_pred.clamp(min=1), torch.zeros_like(tp))
        recall = torch.where(n_label > 0, tp / n_label.clamp(min=1), torch.zeros_like(tp))
        f1 = torch.where(precision + recall > 0, 2 * precision * recall / (precision + recall).clamp(min=1e-300),
                         torch.zeros_like(tp))
        present = n_label > 0                                             # f1_score(labels=np.unique(y_true)), evaluator.py:104-123
        macro_f1 = 100.0 * float(f1[present].mean()) if bool(present.any()) else 0.0
        res = OrderedDict(accuracy=acc, error_rate=100.0 - acc, macro_f1=macro_f1)
        self.per_class_accuracy = (100.0 * recall).tolist()
        self.per_class_f1 = (100.0 * f1).tolist()
        print("=> result\n"
              f"* total: {int(total):,}\n* correct: {int(tp.sum()):,}\n* accuracy: {acc:.1f}%\n"
              f"* error: {100.0 - acc:.1f}%\n* macro_f1: {macro_f1:.1f}%")       # the format parse_test_res.py greps for
        if output_dir:                                                    # evaluator.py:84-113 (csv module formats)
            import csv
            os.makedirs(output_dir, exist_ok=True)
            labels = [i for i in range(self.num_classes) if n_label[i] > 0]
            with open(os.path.join(output_dir, "acc_per_class.csv"), "w", newline="") as f:
                w = csv.writer(f, delimiter=",")
                w.writerow(["Label", "Acc"])
                for key in sorted(str(i) for i in labels):                  # the reference sorts the labels as strings
                    w.writerow([key, self.per_class_accuracy[int(key)]])
            with open(os.path.join(output_dir, "f1_per_class.csv"), "w", newline="") as f:
                w = csv.writer(f, delimiter=",")
                w.writerow(["Label", "F1"])
                for item_id, i in enumerate(labels):
                    w.writerow([item_id, self.per_class_f1[i]])
        return res
