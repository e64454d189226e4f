#!/usr/bin/env python3
"""Generate tests/golden/eval_metrics.npz from the REFERENCE's own src/utils/eval.py (get_accuracy, get_metrics), run in the
build container only.  The file is loaded by path (the reference's `utils` package __init__ is not needed).

Usage:  python tests/golden/gen_golden_eval.py
"""
import importlib.util
import os
import warnings

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
spec = importlib.util.spec_from_file_location("ref_eval", "/root/reference/src/utils/eval.py")
ref = importlib.util.module_from_spec(spec)
spec.loader.exec_module(ref)

KEYS = ["acc", "f1", "precision", "recall", "micro_f1", "micro_precision", "micro_recall", "weighted_f1", "weighted_precision",
        "weighted_recall"]
rng = np.random.default_rng(7)
cases = {}
def add(name, y, p):
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")               # sklearn's zero-division warnings: the value (0.0) is what is pinned
        m = ref.get_metrics(y, p)
        acc = ref.get_accuracy(y, p)
    assert abs(acc - m["acc"]) < 1e-12
    cases[name] = (y.astype(np.float32), p.astype(np.float32), np.array([m[k] for k in KEYS], dtype=np.float64))

add("dense_187x6", (rng.random((187, 6)) > 0.5), (rng.random((187, 6)) > 0.4))
add("sparse_1871x6", (rng.random((1871, 6)) > 0.8), (rng.random((1871, 6)) > 0.75))
y = rng.random((64, 6)) > 0.6; p = rng.random((64, 6)) > 0.5
y[:, 2] = False                                       # a class with no positives
p[:, 4] = False                                       # a class that is never predicted
y[:5] = False; p[:5] = False                          # rows with empty union
add("degenerate_64x6", y, p)
add("perfect_32x6", (yy := rng.random((32, 6)) > 0.5), yy.copy())
add("one_row", np.array([[1, 0, 1, 0, 0, 1]], bool), np.array([[1, 1, 0, 0, 0, 1]], bool))
out = {}
for k, (y, p, m) in cases.items():
    out[k + "/y"] = y; out[k + "/pred"] = p; out[k + "/metrics"] = m
out["keys"] = np.array(KEYS)
np.savez_compressed(os.path.join(HERE, "eval_metrics.npz"), **out)
print("wrote", len(cases), "cases")
