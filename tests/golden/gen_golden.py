#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE itself (run in the build container only;
/root/reference does not exist on the GPU box and nothing at test time reads it).

What runs the reference's own code: ``models.MISA`` (src/models.py), ``utils.DiffLoss`` and
``utils.CMD`` (src/utils/functions.py).  ``src/solver.py`` cannot be imported offline (it fetches
``bert-base-uncased`` at import and needs gensim/wandb/hypertune), so its loop body
(solver.py:139-186) and its six getters (solver.py:373-462) are driven here with the same torch.nn
criteria objects the solver constructs (solver.py:108-118), one class at a time like the solver
does.  The oracle (oracle/misa_oracle.py) states the same losses in closed form; the two must agree.

Usage:  python tests/golden/gen_golden.py [substring]   (writes next to this file; with an argument only the cases whose
                                                         name contains it are regenerated)
"""
import json
import os
import sys
from types import SimpleNamespace

import numpy as np
import torch
import torch.nn as nn

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
ONLY = sys.argv[1] if len(sys.argv) > 1 else ""
sys.argv = sys.argv[:1]
sys.path.insert(0, "/root/reference/src")

import models as ref_models                      # noqa: E402  (the reference)
from utils import DiffLoss, CMD                  # noqa: E402  (the reference)
from oracle import misa_oracle as orc            # noqa: E402  (only for synth_params/synth_batch/default_config)

ACT = {"leakyrelu": nn.LeakyReLU, "relu": nn.ReLU, "tanh": nn.Tanh, "elu": nn.ELU, "prelu": nn.PReLU, "rrelu": nn.RReLU}
SAMPLE_TARGET = 2048


def sample_idx(n):
    stride = max(1, n // SAMPLE_TARGET)
    return np.arange(0, n, stride)


def build_reference(cfg, params):
    rc = SimpleNamespace(**vars(cfg))
    rc.activation = ACT[cfg.activation]
    rc.word2id = list(range(cfg.vocab_size))
    rc.extractor = "lstm"
    rc.use_bert = False
    m = ref_models.MISA(rc)
    missing = m.load_state_dict(params, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    assert list(m.state_dict().keys()) == list(params.keys()), "state_dict order differs from oracle.param_shapes"
    m.train()
    for mod in m.modules():                       # parity is defined dropout-off
        if isinstance(mod, nn.Dropout):
            mod.p = 0.0
        if isinstance(mod, nn.RReLU):             # random slopes are a regulariser like dropout: its evaluation form (the mean slope)
            mod.eval()
    m.transformer_encoder.layers[0].self_attn.dropout = 0.0
    return m


class SolverLosses:
    """The criteria the reference solver builds (solver.py:108-118) applied as its getters do."""

    def __init__(self, cfg):
        self.cfg = cfg
        self.bce = nn.BCELoss(reduction="mean")
        self.dom = nn.CrossEntropyLoss(reduction="mean")
        self.diff = DiffLoss()
        self.mse = nn.MSELoss(reduction="mean")
        self.cmd = CMD()
        self.mcp = nn.CrossEntropyLoss(reduction="mean")
        self.tcp = nn.MSELoss(reduction="mean")

    def __call__(self, m, scores, emo):
        cfg = self.cfg
        ncls = emo.shape[1]
        out = {}
        out["cls"] = sum(self.bce(scores[:, c], emo[:, c]) for c in range(ncls))
        pt, pv, pa = m.utt_private_t, m.utt_private_v, m.utt_private_a
        st, sv, sa = m.utt_shared_t, m.utt_shared_v, m.utt_shared_a
        out["diff"] = (self.diff(pt, st) + self.diff(pv, sv) + self.diff(pa, sa)
                       + self.diff(pa, pt) + self.diff(pa, pv) + self.diff(pt, pv))
        out["recon"] = (self.mse(m.utt_t_recon, m.utt_t_orig) + self.mse(m.utt_v_recon, m.utt_v_orig)
                        + self.mse(m.utt_a_recon, m.utt_a_orig)) / 3.0
        if cfg.use_cmd_sim:
            out["sim"] = (self.cmd(st, sv, 5) + self.cmd(st, sa, 5) + self.cmd(sa, sv, 5)) / 3.0
        else:
            B = scores.shape[0]
            pred = torch.cat((m.domain_label_t, m.domain_label_v, m.domain_label_a), dim=0)
            true = torch.cat((torch.full((B,), 0), torch.full((B,), 1), torch.full((B,), 2))).long()
            out["sim"] = self.dom(pred, true)
        conf = 0.0
        for c in range(ncls):
            nz = torch.count_nonzero(emo[:, c])
            conf = conf + self.tcp(m.tcp[:, c], emo[:, c] * scores[:, c]) / nz
            conf = conf + self.mcp(scores[:, c], emo[:, c]) / nz
        out["conf"] = conf
        total = out["cls"] + cfg.diff_weight * out["diff"] + cfg.sim_weight * out["sim"] + cfg.recon_weight * out["recon"]
        if cfg.use_confidNet:
            total = total + cfg.conf_weight * out["conf"]
        out["total"] = total
        return out


SIDE = ["utt_t_orig", "utt_v_orig", "utt_a_orig", "utt_private_t", "utt_private_v", "utt_private_a",
        "utt_shared_t", "utt_shared_v", "utt_shared_a", "utt_t_recon", "utt_v_recon", "utt_a_recon"]


def run_case(name, cfg, B, T, seed, ragged, full_tensors, steps=3, store_inputs=True):
    if ONLY not in name:
        return
    params = orc.synth_params(cfg, seed)
    model = build_reference(cfg, params)
    losses = SolverLosses(cfg)
    opt = torch.optim.Adam([p for p in model.parameters() if p.requires_grad], lr=cfg.learning_rate)
    rec = {}
    meta = dict(name=name, cfg={k: v for k, v in vars(cfg).items()}, B=B, T=T, seed=seed, ragged=ragged,
                full_tensors=full_tensors, steps=steps, torch=torch.__version__, store_inputs=store_inputs)
    for step in range(steps):
        batch = orc.synth_batch(cfg, B, T, seed + step, ragged)
        model.zero_grad()
        scores, labels = model(batch["t"], batch["v"], batch["a"], batch["l"], None, None, None)
        L = losses(model, scores, batch["emo"])
        L["total"].backward()
        if step == 0:
            if store_inputs:          # large cases: the test rebuilds the batch from (cfg, B, T, seed) with oracle.synth_batch
                for k in ("t", "v", "a", "l", "emo"):
                    rec["in::" + k] = batch[k].numpy()
            else:
                rec["insum::v"] = np.float64(batch["v"].double().sum().item())      # guards the regeneration
                rec["insum::t"] = np.int64(batch["t"].sum().item())
            rec["out::scores"] = scores.detach().numpy()
            rec["out::labels"] = labels.detach().numpy()
            rec["out::tcp"] = model.tcp.detach().numpy()
            for s in SIDE:
                rec["out::" + s] = getattr(model, s).detach().numpy()
            if not cfg.use_cmd_sim:
                for m_ in "tva":
                    rec[f"out::domain_label_{m_}"] = getattr(model, f"domain_label_{m_}").detach().numpy()
            for k, v in L.items():
                rec["loss::" + k] = np.float64(v.item())
            none_grads = []
            for k, p in model.named_parameters():
                if p.grad is None:
                    none_grads.append(k)
                    continue
                g = p.grad.detach().numpy().ravel()
                rec["gnorm::" + k] = np.float64(np.sqrt((g.astype(np.float64) ** 2).sum()))
                rec["gsum::" + k] = np.float64(g.astype(np.float64).sum())
                if full_tensors:
                    rec["grad::" + k] = p.grad.detach().numpy().copy()
                else:
                    rec["gsample::" + k] = g[sample_idx(g.size)].copy()
            meta["none_grads"] = none_grads
        torch.nn.utils.clip_grad_value_([p for p in model.parameters() if p.requires_grad], cfg.clip)
        opt.step()
        rec[f"loss_step{step}::total"] = np.float64(L["total"].item())
    for k, p in model.state_dict().items():
        a = p.detach().numpy()
        if full_tensors:
            rec["param0::" + k] = params[k].numpy()
            rec[f"param{steps}::" + k] = a.copy()
        else:
            rec[f"psample{steps}::" + k] = a.ravel()[sample_idx(a.size)].copy()
    rec["meta"] = np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **rec)
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB  total={rec['loss::total']:.6f}")


def main():
    torch.manual_seed(0)
    torch.set_num_threads(8)
    tiny = dict(embedding_size=12, visual_size=5, acoustic_size=7, hidden_size=16, vocab_size=30)
    run_case("tiny_cmd_ragged", orc.default_config(**tiny), B=4, T=7, seed=1, ragged=True, full_tensors=True)
    run_case("tiny_adv_confid_full", orc.default_config(use_cmd_sim=False, use_confidNet=True, **tiny),
             B=5, T=6, seed=2, ragged=False, full_tensors=True)
    run_case("tiny_relu_confid_ragged", orc.default_config(activation="relu", use_confidNet=True, **tiny),
             B=6, T=9, seed=3, ragged=True, full_tensors=True)
    real = dict(vocab_size=64)
    run_case("real_b8_t12_ragged", orc.default_config(**real), B=8, T=12, seed=4, ragged=True, full_tensors=False)
    run_case("real_b32_t50_full", orc.default_config(vocab_size=512), B=32, T=50, seed=5, ragged=False,
             full_tensors=False, steps=3)
    run_case("real_b16_t20_adv_confid", orc.default_config(use_cmd_sim=False, use_confidNet=True, **real),
             B=16, T=20, seed=6, ragged=True, full_tensors=False)
    # config.activation = prelu / rrelu (config.py:25-27): the learned slope is ONE nn.PReLU shared by the projections and the discriminator
    run_case("tiny_prelu_adv_ragged", orc.default_config(activation="prelu", use_cmd_sim=False, **tiny), B=6, T=7, seed=12, ragged=True,
             full_tensors=True)
    run_case("tiny_rrelu_ragged", orc.default_config(activation="rrelu", **tiny), B=5, T=6, seed=13, ragged=True, full_tensors=True)
    # BASELINE.json configs[3] (seq_len = 500) and the per-GPU batch of configs[2] (B = 256), one step each
    run_case("real_b32_t500_ragged", orc.default_config(**real), B=32, T=500, seed=10, ragged=True, full_tensors=False, steps=1,
             store_inputs=False)
    run_case("real_b256_t6_full", orc.default_config(**real), B=256, T=6, seed=11, ragged=False, full_tensors=False, steps=1,
             store_inputs=False)
    # config.rnncell != 'lstm' -> nn.GRU encoders (reference models.py:39)
    run_case("tiny_gru_ragged", orc.default_config(rnncell="gru", use_confidNet=True, **tiny), B=5, T=8, seed=7, ragged=True,
             full_tensors=True)
    run_case("real_gru_b8_t12_ragged", orc.default_config(rnncell="gru", **real), B=8, T=12, seed=8, ragged=True, full_tensors=False)
    run_case("real_gru_b16_t20_adv", orc.default_config(rnncell="gru", use_cmd_sim=False, **real), B=16, T=20, seed=9, ragged=False,
             full_tensors=False)


if __name__ == "__main__":
    main()
