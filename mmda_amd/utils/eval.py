"""Evaluation metrics with the reference's call surface (src/utils/eval.py): ``get_accuracy(y, y_pre)`` and
``get_metrics(y, y_pre)`` on host arrays, plus ``DeviceEval``, which accumulates the same quantities on the GPU across the
batches of an evaluation pass (native kernel ``mmda_eval_accumulate``) and needs one read-back at the end.

get_metrics: the reference calls sklearn.metrics f1/precision/recall with average macro / micro / weighted on multilabel
indicator matrices; all of them are functions of the per-class tp / fp / fn (zero when a denominator is zero), which is what is
counted here -- sklearn itself is not needed at run time.
"""
from __future__ import annotations

import numpy as np
import torch

from .. import _lib

KEYS = ["acc", "f1", "precision", "recall", "micro_f1", "micro_precision", "micro_recall", "weighted_f1", "weighted_precision",
        "weighted_recall"]


def _div(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return np.where(b > 0, a / np.where(b > 0, b, 1.0), 0.0)


def metrics_from_counts(tp, fp, fn, acc):
    """tp, fp, fn: per-class counts; acc: the Jaccard-style accuracy.  Returns the reference's get_metrics dict."""
    tp, fp, fn = (np.asarray(x, dtype=np.float64) for x in (tp, fp, fn))
    prec, rec, f1 = _div(tp, tp + fp), _div(tp, tp + fn), _div(2 * tp, 2 * tp + fp + fn)
    sup = tp + fn
    w = _div(sup, sup.sum()) if sup.sum() > 0 else np.zeros_like(sup)
    TP, FP, FN = tp.sum(), fp.sum(), fn.sum()
    return {"acc": acc, "f1": float(f1.mean()), "precision": float(prec.mean()), "recall": float(rec.mean()),
            "micro_f1": float(_div(2 * TP, 2 * TP + FP + FN)), "micro_precision": float(_div(TP, TP + FP)),
            "micro_recall": float(_div(TP, TP + FN)),
            "weighted_f1": float((w * f1).sum()), "weighted_precision": float((w * prec).sum()),
            "weighted_recall": float((w * rec).sum())}


def get_accuracy(y, y_pre):
    """reference utils/eval.py:14-31, vectorised: mean_i |y_i & p_i| / max(|y_i | p_i|, 1), rounded to 4 places."""
    y = np.asarray(y) > 0
    p = np.asarray(y_pre) > 0
    if y.ndim == 1:
        y, p = y[None], p[None]
    inter = (y & p).sum(axis=1).astype(np.float64)
    union = (y | p).sum(axis=1).astype(np.float64)
    union[union <= 0] = 1.0
    return round(float((inter / union).sum() / y.shape[0]), 4)


def get_metrics(y, y_pre):
    """reference utils/eval.py:33-65."""
    yb = np.asarray(y) > 0
    pb = np.asarray(y_pre) > 0
    if yb.ndim == 1:
        yb, pb = yb[None], pb[None]
    tp = (yb & pb).sum(0); fp = (~yb & pb).sum(0); fn = (yb & ~pb).sum(0)
    return metrics_from_counts(tp, fp, fn, get_accuracy(y, y_pre))


class DeviceEval:
    """Accumulates tp / fp / fn / accuracy sum / sample count (and the running sum of per-batch losses) on the device."""

    def __init__(self, num_classes: int, device):
        self.C = int(num_classes)
        self.state = torch.zeros(3 * self.C + 2, dtype=torch.float64, device=device)
        self.loss_sum = torch.zeros(1, dtype=torch.float32, device=device)
        self.batches = 0
        self._lib = _lib.load()

    def update(self, pred_labels: torch.Tensor, truth: torch.Tensor):
        """pred_labels, truth: (N, C) fp32 on the device; no synchronisation."""
        p = pred_labels.detach().to(torch.float32).contiguous()
        t = truth.detach().to(torch.float32).contiguous()
        if not (p.is_cuda and t.is_cuda):
            raise _lib.MMDAError("DeviceEval.update needs device tensors (the HIP path has no CPU fallback)")
        _lib.check(self._lib.mmda_eval_accumulate(p.data_ptr(), t.data_ptr(), p.shape[0], self.C, self.state.data_ptr(),
                                                  _lib.stream_ptr()), "mmda_eval_accumulate")

    def add_cls_loss(self, scores: torch.Tensor, truth: torch.Tensor):
        """adds this batch's classification loss (solver.py:373-385) to the running sum, on the device"""
        s = scores.detach().to(torch.float32).contiguous(); t = truth.detach().to(torch.float32).contiguous()
        _lib.check(self._lib.mmda_loss_cls(s.data_ptr(), t.data_ptr(), s.shape[0], s.shape[1], 1.0, self.loss_sum.data_ptr(), None,
                                           _lib.stream_ptr()), "mmda_loss_cls")
        self.batches += 1

    def result(self):
        """one device->host read: (mean batch loss, accuracy, metrics dict)"""
        st = self.state.cpu().numpy()
        C = self.C
        n = max(st[3 * C + 1], 1.0)
        acc = round(float(st[3 * C] / n), 4)
        loss = float(self.loss_sum.item()) / max(self.batches, 1)
        return loss, acc, metrics_from_counts(st[:C], st[C:2 * C], st[2 * C:3 * C], acc)
