"""CPU: the evaluation-metric oracle against the golden vectors produced by the reference's own src/utils/eval.py."""
import os

import numpy as np

from oracle import eval_oracle as ev

G = np.load(os.path.join(os.path.dirname(__file__), "golden", "eval_metrics.npz"))
CASES = sorted({k.split("/")[0] for k in G.files if "/" in k})


def test_metric_oracle_matches_reference_outputs():
    assert list(G["keys"]) == ev.KEYS
    for c in CASES:
        got = ev.get_metrics(G[c + "/y"], G[c + "/pred"])
        ref = G[c + "/metrics"]
        for k, r in zip(ev.KEYS, ref):
            assert abs(got[k] - r) < 1e-12, (c, k, got[k], r)
        assert ev.get_accuracy(G[c + "/y"], G[c + "/pred"]) == ref[0]
