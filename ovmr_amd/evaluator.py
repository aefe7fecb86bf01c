"""On-device classification evaluator.

Arithmetic of Dassl's `Classification` evaluator (Dassl.pytorch/dassl/evaluation/evaluator.py:50-138): top-1 accuracy,
error rate, macro-F1 and per-class accuracy / F1, accumulated as three histograms on the GPU by ONE launch per batch of
the library's own row-argmax-and-count kernel (`ovmr_eval_counts`, csrc/fusion_head.hip: the cross-validation step's
counting, on the [B, C] outputs) -- the reference does `.item()` and `.cpu().numpy()` per batch (:59-67), i.e. one host
sync per batch of 256 images; here the host reads 3C + 1 integers once, in evaluate().

Outputs that live on the GPU go through the kernel and nothing else: without libovmr_hip.so `process` raises.  Host
tensors (an evaluator built with device="cpu": host-side callers, the CPU tests of evaluate()'s arithmetic) are counted
on the host.
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
        # int32 [3][C] = tp, n_pred, n_label, + one slot counting rows whose label is outside [0, C) (include/ovmr_hip.h: ovmr_eval_counts)
        self._counts = torch.zeros(3 * self.num_classes + 1, dtype=torch.int32, device=self.device)

    @torch.no_grad()
    def process(self, mo: torch.Tensor, gt: torch.Tensor):
        """mo: [B, C] model output (fp32 probabilities of CustomCLIP.forward, or fp16 zero-shot logits), gt: [B] labels
        (evaluator.py:50-67).  One kernel launch on the current stream, no host synchronisation."""
        C = self.num_classes
        if mo.dim() != 2 or mo.shape[1] != C or gt.shape[0] != mo.shape[0]:
            raise ValueError(f"outputs {tuple(mo.shape)} / labels {tuple(gt.shape)} do not fit {C} classes")
        if self.device.type == "cpu":
            self._process_host(mo, gt)
            return
        from . import runtime
        lib = runtime.load_library()                                      # raises without the HIP library: no fallback for device tensors
        mo = mo.to(self.device)
        if mo.dtype not in (torch.float16, torch.float32):
            mo = mo.float()
        if mo.stride(1) != 1:
            mo = mo.contiguous()
        gt = gt.to(self.device, non_blocking=True)
        if gt.dtype != torch.int64 or not gt.is_contiguous():
            gt = gt.long().contiguous()
        rc = lib.ovmr_eval_counts(runtime._ptr(mo), runtime.F32 if mo.dtype == torch.float32 else runtime.F16, mo.stride(0),
                                  runtime._ptr(gt), mo.shape[0], C, runtime._ptr(self._counts), runtime._stream())
        if rc != 0:
            raise runtime.OvmrError(f"ovmr_eval_counts failed with {rc}")

    def _process_host(self, mo, gt):
        """The same three histograms for host tensors (mo.max(1)[1]: lowest column on ties)."""
        C = self.num_classes
        pred = mo.float().argmax(dim=1)
        gt = gt.long()
        bad = (gt < 0) | (gt >= C)
        ok = ~bad
        c = self._counts
        c[3 * C] += int(bad.sum())
        c[2 * C:3 * C] += torch.bincount(gt[ok], minlength=C).int()
        c[C:2 * C] += torch.bincount(pred[ok], minlength=C).int()
        c[:C] += torch.bincount(gt[ok & (pred == gt)], minlength=C).int()

    def counts(self):
        """(tp, n_pred, n_label) as int64 host tensors [C]; raises if a label outside [0, C) was seen."""
        c = self._counts.cpu().long()
        C = self.num_classes
        if int(c[3 * C]):
            raise ValueError(f"{int(c[3 * C])} test label(s) outside [0, {C})")
        return c[:C], c[C:2 * C], c[2 * C:3 * C]

    def evaluate(self, output_dir: Optional[str] = None) -> "OrderedDict[str, float]":
        tp, n_pred, n_label = (t.double() for t in self.counts())
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
