"""CPU restatement of the reference's evaluation metrics (TEST INFRASTRUCTURE: only tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg may import anything under oracle/).

  get_accuracy  -- reference src/utils/eval.py:14-31: mean over samples of |y & p| / max(|y | p|, 1), rounded to 4 places.
  get_metrics   -- reference src/utils/eval.py:33-65: sklearn.metrics f1/precision/recall with average in
                   {macro, micro, weighted} on multilabel indicator matrices.  sklearn (1.7.2 in the build container; the
                   reference pins no version) is not restated line by line; its published definitions are:
                     per class c: P_c = tp/(tp+fp), R_c = tp/(tp+fn), F_c = 2 tp/(2 tp+fp+fn), each 0 when its denominator is 0
                     macro = mean_c, weighted = sum_c support_c * x_c / sum_c support_c (support = tp+fn),
                     micro = the same formulas on the counts summed over classes.
Pinned by tests/golden/eval_metrics.npz, generated from the reference's own eval.py (tests/golden/gen_golden_eval.py).
"""
import numpy as np

KEYS = ["acc", "f1", "precision", "recall", "micro_f1", "micro_precision", "micro_recall", "weighted_f1", "weighted_precision",
        "weighted_recall"]


def get_accuracy(y, y_pre):
    y = np.asarray(y) > 0
    p = np.asarray(y_pre) > 0
    count = 0.0
    for i in range(y.shape[0]):                      # the reference's loop, row by row
        inter = int((y[i] & p[i]).sum())
        union = int((y[i] | p[i]).sum())
        if union <= 0:
            union = 1
        count += float(inter) / float(union)
    return round(count / float(y.shape[0]), 4)


def counts(y, y_pre):
    y = np.asarray(y) > 0
    p = np.asarray(y_pre) > 0
    tp = (y & p).sum(0).astype(np.float64)
    fp = (~y & p).sum(0).astype(np.float64)
    fn = (y & ~p).sum(0).astype(np.float64)
    return tp, fp, fn


def _div(a, b):
    a = np.asarray(a, dtype=np.float64); b = np.asarray(b, dtype=np.float64)
    return np.where(b > 0, a / np.where(b > 0, b, 1.0), 0.0)


def metrics_from_counts(tp, fp, fn, acc):
    prec, rec, f1 = _div(tp, tp + fp), _div(tp, tp + fn), _div(2 * tp, 2 * tp + fp + fn)
    sup = tp + fn
    w = _div(sup, sup.sum()) if sup.sum() > 0 else np.zeros_like(sup)
    TP, FP, FN = tp.sum(), fp.sum(), fn.sum()
    return {"acc": acc, "f1": float(f1.mean()), "precision": float(prec.mean()), "recall": float(rec.mean()),
            "micro_f1": float(_div(2 * TP, 2 * TP + FP + FN)), "micro_precision": float(_div(TP, TP + FP)),
            "micro_recall": float(_div(TP, TP + FN)),
            "weighted_f1": float((w * f1).sum()), "weighted_precision": float((w * prec).sum()),
            "weighted_recall": float((w * rec).sum())}


def get_metrics(y, y_pre):
    tp, fp, fn = counts(y, y_pre)
    return metrics_from_counts(tp, fp, fn, get_accuracy(y, y_pre))
