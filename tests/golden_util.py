"""Helpers to read tests/golden/*.npz (written by tests/golden/gen_golden.py from the reference)."""
import glob
import json
import os
from types import SimpleNamespace

import numpy as np
import torch

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SAMPLE_TARGET = 2048

SIDE = ["utt_t_orig", "utt_v_orig", "utt_a_orig", "utt_private_t", "utt_private_v", "utt_private_a",
        "utt_shared_t", "utt_shared_v", "utt_shared_a", "utt_t_recon", "utt_v_recon", "utt_a_recon"]


LARGE = ("real_b32_t500_ragged", "real_b256_t6_full")     # BASELINE configs[3] / configs[2] shapes: tests of their own


def case_names(large=False):
    # the model fixtures of gen_golden.py (eval_metrics.npz belongs to gen_golden_eval.py and has its own tests); the large cases
    # (no stored inputs, a CPU oracle backward of a minute at T = 500) only on request
    names = sorted(os.path.basename(p)[:-4] for p in glob.glob(os.path.join(GOLDEN_DIR, "*.npz"))
                   if os.path.basename(p) != "eval_metrics.npz")
    return [n for n in names if large or n not in LARGE]


def sample_idx(n):
    stride = max(1, n // SAMPLE_TARGET)
    return np.arange(0, n, stride)


def load_case(name):
    z = np.load(os.path.join(GOLDEN_DIR, name + ".npz"), allow_pickle=False)
    meta = json.loads(bytes(z["meta"]).decode())
    cfg = SimpleNamespace(**meta["cfg"])
    return z, meta, cfg


def batch_of(z):
    if "in::t" in z.files:
        return {k: torch.from_numpy(z["in::" + k]) for k in ("t", "v", "a", "l", "emo")}
    # large cases store no inputs: they are rebuilt from (cfg, B, T, seed) exactly as the generator built them
    from oracle import misa_oracle as orc
    meta = json.loads(bytes(z["meta"]).decode())
    b = orc.synth_batch(SimpleNamespace(**meta["cfg"]), meta["B"], meta["T"], meta["seed"], meta["ragged"])
    assert float(b["v"].double().sum()) == float(z["insum::v"]) and int(b["t"].sum()) == int(z["insum::t"])
    return b
