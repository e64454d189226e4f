"""CPU oracle for the MISA tri-modal training hot path.  TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only ``tests/``, ``__graft_entry__.smoke()``
and ``bench.py``'s ``cpu_baseline`` leg may import it.  ``mmda_amd`` (the product) never does.

What it is: a plain-PyTorch fp32 CPU restatement of the reference's algorithm, written as
pure functions over a ``{state_dict key: tensor}`` dict (the reference is an ``nn.Module``
with side-channel attributes; this is deliberately a different shape so nothing is copied):

  * model forward  ........ reference ``src/models.py:163-180`` (extract_features),
                            ``:182-250`` (alignment), ``:254-262`` (reconstruct),
                            ``:265-279`` (shared_private)
  * loss getters  ......... reference ``src/solver.py:373-462`` and
                            ``src/utils/functions.py:49-109`` (DiffLoss, CMD)
  * loss combine/step ..... reference ``src/solver.py:170-186`` (weights ``src/config.py:134-142``),
                            Adam ``src/solver.py:97-99``

Third-party arithmetic: every FLOP of the reference is PyTorch (``nn.LSTM``,
``nn.TransformerEncoderLayer`` slow path, ``nn.LayerNorm``, ``nn.Linear``, ``BCELoss``,
``MSELoss``, ``CrossEntropyLoss``, ``clip_grad_value_``, ``optim.Adam``).  The reference pins no
torch version; this container and the GPU box carry torch 2.10.0.  The recurrent encoders here
call the same ``nn.LSTM`` packed-sequence kernels the reference calls; the fusion transformer
layer, the losses and Adam are restated in closed form (matmul / softmax / mean) so they are an
independent derivation of what the reference's module calls compute.

Pinning: ``tests/golden/*.npz`` were produced by ``tests/golden/gen_golden.py`` which imports
the reference's own ``models.MISA`` / ``utils.DiffLoss`` / ``utils.CMD`` in the build container
(the reference's ``solver.py`` cannot be imported offline: it fetches ``bert-base-uncased`` at
import and needs gensim/wandb/hypertune; its six getters are composed in the generator from the
reference's importable loss modules + torch.nn criteria exactly as the solver text does).
``tests/test_oracle_golden.py`` checks this oracle against every fixture.  The reference has no
tests or golden vectors of its own.
"""
from __future__ import annotations

import math
from types import SimpleNamespace
from typing import Dict, Optional

import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.utils.rnn import pack_padded_sequence, pad_packed_sequence

Params = Dict[str, torch.Tensor]

PRELU_KEY = "activation.weight"      # first registration of the shared nn.PReLU (self.activation, models.py:30); the rest alias it
FFN_DIM = 2048      # nn.TransformerEncoderLayer default dim_feedforward (reference models.py:160)
NHEAD = 2           # reference models.py:160
LN_EPS = 1e-5       # torch default

_ACTS = {
    "elu": F.elu, "hardshrink": F.hardshrink, "hardtanh": F.hardtanh,
    "leakyrelu": lambda x: F.leaky_relu(x, 0.01), "relu": F.relu, "tanh": torch.tanh,
}


def default_config(**kw) -> SimpleNamespace:
    """Field names follow reference src/config.py:99-170 (only those the hot path reads)."""
    c = SimpleNamespace(
        embedding_size=300, visual_size=35, acoustic_size=74, hidden_size=128, num_classes=6,
        vocab_size=20000, dropout=0.1, activation="leakyrelu", rnncell="lstm",
        use_cmd_sim=True, use_confidNet=False, reverse_grad_weight=1.0, threshold=0.35,
        diff_weight=0.3, sim_weight=0.7, recon_weight=0.7, conf_weight=0.3,
        learning_rate=1e-4, clip=1.0)
    for k, v in kw.items():
        setattr(c, k, v)
    return c


# ----------------------------------------------------------------------------- parameters
def param_shapes(cfg) -> Dict[str, tuple]:
    """state_dict key -> shape, in the reference's registration order (models.py:47-161)."""
    dt, dv, da, hs = cfg.embedding_size, cfg.visual_size, cfg.acoustic_size, cfg.hidden_size
    s: Dict[str, tuple] = {}
    if getattr(cfg, "activation", "leakyrelu") == "prelu":
        s[PRELU_KEY] = (1,)
    s["embed.weight"] = (cfg.vocab_size, dt)

    ng = 4 if getattr(cfg, "rnncell", "lstm") == "lstm" else 3      # models.py:39: nn.LSTM if rnncell == 'lstm' else nn.GRU

    def rnn(prefix, din, h):
        for sfx in ("", "_reverse"):
            s[f"{prefix}.weight_ih_l0{sfx}"] = (ng * h, din)
            s[f"{prefix}.weight_hh_l0{sfx}"] = (ng * h, h)
            s[f"{prefix}.bias_ih_l0{sfx}"] = (ng * h,)
            s[f"{prefix}.bias_hh_l0{sfx}"] = (ng * h,)

    rnn("trnn1", dt, dt); rnn("trnn2", 2 * dt, dt)
    rnn("vrnn1", dv, dv); rnn("vrnn2", 2 * dv, dv)
    rnn("arnn1", da, da); rnn("arnn2", 2 * da, da)
    prelu = getattr(cfg, "activation", "leakyrelu") == "prelu"     # ONE nn.PReLU() instance shared by every use (models.py:30): aliases
    for m, d in (("t", dt), ("v", dv), ("a", da)):
        s[f"project_{m}.project_{m}.weight"] = (hs, 4 * d)
        s[f"project_{m}.project_{m}.bias"] = (hs,)
        if prelu:
            s[f"project_{m}.project_{m}_activation.weight"] = (1,)
        s[f"project_{m}.project_{m}_layer_norm.weight"] = (hs,)
        s[f"project_{m}.project_{m}_layer_norm.bias"] = (hs,)
    for name in ("private_t.private_t_1", "private_v.private_v_1", "private_a.private_a_3",
                 "shared.shared_1", "recon_t.recon_t_1", "recon_v.recon_v_1", "recon_a.recon_a_1"):
        s[name + ".weight"] = (hs, hs)
        s[name + ".bias"] = (hs,)
    if not cfg.use_cmd_sim:
        s["discriminator.discriminator_layer_1.weight"] = (hs, hs)
        s["discriminator.discriminator_layer_1.bias"] = (hs,)
        if prelu:
            s["discriminator.discriminator_layer_1_activation.weight"] = (1,)
        s["discriminator.discriminator_layer_2.weight"] = (3, hs)
        s["discriminator.discriminator_layer_2.bias"] = (3,)
    s["sp_discriminator.sp_discriminator_layer_1.weight"] = (4, hs)
    s["sp_discriminator.sp_discriminator_layer_1.bias"] = (4,)
    s["confidence.confidence_layer_1.weight"] = (6, 6 * hs)
    s["confidence.confidence_layer_1.bias"] = (6,)
    s["classifier.classifier_layer.weight"] = (cfg.num_classes, 6 * hs)
    s["classifier.classifier_layer.bias"] = (cfg.num_classes,)
    for m, d in (("t", dt), ("v", dv), ("a", da)):
        s[f"{m}layer_norm.weight"] = (2 * d,)
        s[f"{m}layer_norm.bias"] = (2 * d,)
    te = "transformer_encoder.layers.0."
    s[te + "self_attn.in_proj_weight"] = (3 * hs, hs)
    s[te + "self_attn.in_proj_bias"] = (3 * hs,)
    s[te + "self_attn.out_proj.weight"] = (hs, hs)
    s[te + "self_attn.out_proj.bias"] = (hs,)
    s[te + "linear1.weight"] = (FFN_DIM, hs)
    s[te + "linear1.bias"] = (FFN_DIM,)
    s[te + "linear2.weight"] = (hs, FFN_DIM)
    s[te + "linear2.bias"] = (hs,)
    for n in ("norm1", "norm2"):
        s[te + n + ".weight"] = (hs,)
        s[te + n + ".bias"] = (hs,)
    return s


def synth_params(cfg, seed: int) -> Params:
    """Deterministic parameters from numpy PCG64 (NOT torch's RNG), so the golden generator,
    the oracle and the HIP model can all rebuild the same weights from (cfg, seed) alone."""
    import numpy as np
    rng = np.random.default_rng(seed)
    out: Params = {}
    shared_prelu = None
    for k, shp in param_shapes(cfg).items():
        if k.endswith("activation.weight"):              # the shared nn.PReLU slope: one tensor under every alias
            if shared_prelu is None:
                shared_prelu = torch.tensor([0.25 + float(rng.uniform(-0.1, 0.1))], dtype=torch.float32)
            out[k] = shared_prelu
            continue
        if k == "embed.weight":
            a = rng.standard_normal(shp) * 0.5
        elif "layer_norm.weight" in k or k.endswith("norm1.weight") or k.endswith("norm2.weight"):
            a = 1.0 + rng.uniform(-0.1, 0.1, shp)
        elif len(shp) == 1:
            a = rng.uniform(-0.1, 0.1, shp)
        else:
            a = rng.uniform(-1.0, 1.0, shp) / math.sqrt(shp[1])
        out[k] = torch.tensor(a, dtype=torch.float32)
    return out


def synth_batch(cfg, B: int, T: int, seed: int, ragged: bool):
    """MOSEI-shaped synthetic batch, time-major like reference data_loader.py:70-72.
    Returns dict(t (T,B) i64, v (T,B,dv), a (T,B,da), l (B,) i64 sorted desc, emo (B,6) f32)."""
    import numpy as np
    rng = np.random.default_rng(10_000 + seed)
    t = rng.integers(2, cfg.vocab_size, (T, B))
    v = rng.standard_normal((T, B, cfg.visual_size))
    a = rng.standard_normal((T, B, cfg.acoustic_size))
    if ragged:
        l = np.sort(rng.integers(1, T + 1, B))[::-1].copy()
        l[0] = T
        if B > 1:
            l[-1] = 1
    else:
        l = np.full((B,), T)
    emo = (rng.random((B, 6)) > 0.6).astype("float32")
    for c in range(6):                      # every class >=1 positive (avoid /nnz==0, solver.py:459)
        if emo[:, c].sum() == 0:
            emo[c % B, c] = 1.0
    return dict(t=torch.tensor(t, dtype=torch.int64), v=torch.tensor(v, dtype=torch.float32),
                a=torch.tensor(a, dtype=torch.float32), l=torch.tensor(l, dtype=torch.int64),
                emo=torch.tensor(emo, dtype=torch.float32))


# ----------------------------------------------------------------------------- encoders
def _bilstm(x, lengths, P: Params, prefix: str, din: int, h: int, cell: str = "lstm"):
    """nn.LSTM / nn.GRU (bidirectional) over a packed sequence (reference models.py:39, 164-178)."""
    rnn = nn.LSTM(din, h, bidirectional=True) if cell == "lstm" else nn.GRU(din, h, bidirectional=True)
    names = [n for n, _ in rnn.named_parameters()]
    packed = pack_padded_sequence(x, lengths, enforce_sorted=False)
    out, hn = torch.func.functional_call(rnn, {n: P[f"{prefix}.{n}"] for n in names}, (packed,))
    return out, (hn[0] if cell == "lstm" else hn)          # models.py:166-169


def encode_modality(x, lengths, P: Params, m: str, d: int, cell: str = "lstm"):
    """Two stacked bidirectional RNNs with LayerNorm between (models.py:163-180) and the utterance vector
    [h1_fwd, h2_fwd, h1_bwd, h2_bwd] per sample (models.py:203)."""
    out1, h1 = _bilstm(x, lengths, P, f"{m}rnn1", x.shape[-1], d, cell)
    padded, _ = pad_packed_sequence(out1)
    normed = F.layer_norm(padded, (2 * d,), P[f"{m}layer_norm.weight"], P[f"{m}layer_norm.bias"], LN_EPS)
    _, h2 = _bilstm(normed, lengths, P, f"{m}rnn2", 2 * d, d, cell)
    B = x.shape[1]
    return torch.cat((h1, h2), dim=2).permute(1, 0, 2).reshape(B, 4 * d)


def lstm_dir_loop(x, lengths, w_ih, w_hh, b_ih, b_hh, reverse: bool):
    """Independent explicit-loop single-direction LSTM with packed-sequence semantics, used by
    tests to cross-check the masking rules the HIP kernel implements (gate order i,f,g,o)."""
    T, B, _ = x.shape
    H = w_hh.shape[1]
    h = x.new_zeros(B, H); c = x.new_zeros(B, H)
    out = x.new_zeros(T, B, H)
    steps = range(T - 1, -1, -1) if reverse else range(T)
    for t in steps:
        g = x[t] @ w_ih.t() + b_ih + h @ w_hh.t() + b_hh
        i, f, gg, o = g.chunk(4, dim=1)
        c2 = torch.sigmoid(f) * c + torch.sigmoid(i) * torch.tanh(gg)
        h2 = torch.sigmoid(o) * torch.tanh(c2)
        m = (t < lengths).to(x.dtype).unsqueeze(1)
        c = m * c2 + (1 - m) * c
        h = m * h2 + (1 - m) * h
        out[t] = m * h2
    return out, h


# ----------------------------------------------------------------------------- fusion
def _linear(x, P, name):
    return x @ P[name + ".weight"].t() + P[name + ".bias"]


def ffn_exact(x, P: Params):
    """linear2(relu(linear1(x))) of the encoder layer (torch/nn/modules/transformer.py _ff_block, dropout off)."""
    te = "transformer_encoder.layers.0."
    return _linear(torch.relu(_linear(x, P, te + "linear1")), P, te + "linear2")


def fusion_layer(x, P: Params, ffn=None):
    """Post-norm transformer encoder layer on (S=6, B, E) (models.py:160-161,243-245);
    restated from torch/nn/modules/transformer.py slow path: x=LN(x+SA(x)); x=LN(x+FF(x)).
    ``ffn``: replacement for the feed-forward block (oracle/fp8_emul.py emulates the block-scaled fp8 products with it)."""
    te = "transformer_encoder.layers.0."
    S, B, E = x.shape
    hd = E // NHEAD
    qkv = x @ P[te + "self_attn.in_proj_weight"].t() + P[te + "self_attn.in_proj_bias"]
    q, k, v = qkv.split(E, dim=-1)

    def heads(z):                                   # (S,B,E) -> (B,NHEAD,S,hd)
        return z.reshape(S, B, NHEAD, hd).permute(1, 2, 0, 3)

    q, k, v = heads(q), heads(k), heads(v)
    att = torch.softmax((q @ k.transpose(-1, -2)) / math.sqrt(hd), dim=-1)
    ctx = (att @ v).permute(2, 0, 1, 3).reshape(S, B, E)
    sa = _linear(ctx, P, te + "self_attn.out_proj")
    x = F.layer_norm(x + sa, (E,), P[te + "norm1.weight"], P[te + "norm1.bias"], LN_EPS)
    ff = (ffn or ffn_exact)(x, P)
    return F.layer_norm(x + ff, (E,), P[te + "norm2.weight"], P[te + "norm2.bias"], LN_EPS)


class _GradReverse(torch.autograd.Function):
    """Gradient-reversal layer (reference utils/functions.py:9-21)."""
    @staticmethod
    def forward(ctx, x, p):
        ctx.p = p
        return x.view_as(x)

    @staticmethod
    def backward(ctx, g):
        return -g * ctx.p, None


def forward(P: Params, cfg, t, v, a, lengths, ffn=None) -> SimpleNamespace:
    """Full model forward with all dropout disabled.  Returns every tensor the reference's
    solver reads off the module (SURVEY.md 8b side channel)."""
    dt, dv, da, hs = cfg.embedding_size, cfg.visual_size, cfg.acoustic_size, cfg.hidden_size
    lengths = lengths.cpu()
    emb = P["embed.weight"][t]                                          # models.py:201
    cell = "lstm" if getattr(cfg, "rnncell", "lstm") == "lstm" else "gru"
    utt = {"t": encode_modality(emb, lengths, P, "t", dt, cell),
           "v": encode_modality(v, lengths, P, "v", dv, cell),
           "a": encode_modality(a, lengths, P, "a", da, cell)}
    return fusion_from_utterances(P, cfg, utt, ffn)


def fusion_from_utterances(P: Params, cfg, utt, ffn=None) -> SimpleNamespace:
    """Everything of ``forward`` behind the encoders (models.py:216-249): projections, private/shared, discriminator,
    reconstruction, the transformer fusion layer and the heads, from the three utterance vectors {"t","v","a"}."""
    hs = cfg.hidden_size
    if cfg.activation == "prelu":        # learned slope, one parameter shared by the three projections and the discriminator
        slope = P[PRELU_KEY]
        act = lambda x: F.prelu(x, slope)
    elif cfg.activation == "rrelu":      # nn.RReLU() in evaluation mode: the mean slope (1/8 + 1/3) / 2; parity is defined dropout-off
        act = lambda x: F.rrelu(x, training=False)
    else:
        act = _ACTS[cfg.activation]
    o = SimpleNamespace()
    o.utterance_t, o.utterance_v, o.utterance_a = utt["t"], utt["v"], utt["a"]
    priv_names = {"t": "private_t.private_t_1", "v": "private_v.private_v_1", "a": "private_a.private_a_3"}
    for m in "tva":
        z = act(_linear(utt[m], P, f"project_{m}.project_{m}"))
        orig = F.layer_norm(z, (hs,), P[f"project_{m}.project_{m}_layer_norm.weight"],
                            P[f"project_{m}.project_{m}_layer_norm.bias"], LN_EPS)
        setattr(o, f"utt_{m}_orig", orig)
        setattr(o, f"utt_private_{m}", torch.sigmoid(_linear(orig, P, priv_names[m])))
        setattr(o, f"utt_shared_{m}", torch.sigmoid(_linear(orig, P, "shared.shared_1")))
    for m in "tva":                                                     # models.py:219-231
        if cfg.use_cmd_sim:
            setattr(o, f"domain_label_{m}", None)
        else:
            r = _GradReverse.apply(getattr(o, f"utt_shared_{m}"), cfg.reverse_grad_weight)
            hdn = act(_linear(r, P, "discriminator.discriminator_layer_1"))
            setattr(o, f"domain_label_{m}", _linear(hdn, P, "discriminator.discriminator_layer_2"))
    for m in "tva":                                                     # models.py:254-262
        s = getattr(o, f"utt_private_{m}") + getattr(o, f"utt_shared_{m}")
        setattr(o, f"utt_{m}_recon", _linear(s, P, f"recon_{m}.recon_{m}_1"))
    x = torch.stack((o.utt_private_t, o.utt_private_v, o.utt_private_a,
                     o.utt_shared_t, o.utt_shared_v, o.utt_shared_a), dim=0)
    hfused = fusion_layer(x, P, ffn)
    o.h = hfused.permute(1, 0, 2).reshape(x.shape[1], 6 * hs)          # == cat(h[0..5], dim=1)
    o.tcp = torch.sigmoid(_linear(o.h, P, "confidence.confidence_layer_1"))
    o.scores = torch.sigmoid(_linear(o.h, P, "classifier.classifier_layer"))
    o.labels = (o.scores > cfg.threshold).to(o.scores.dtype)           # getBinaryTensor
    return o


# ----------------------------------------------------------------------------- losses
def cls_loss(scores, emo):
    """sum_c mean_b BCE(scores[:,c], emo[:,c]) with log clamped at -100 (solver.py:373-385)."""
    lp = torch.clamp(torch.log(scores), min=-100.0)
    lq = torch.clamp(torch.log1p(-scores), min=-100.0)
    return (-(emo * lp + (1 - emo) * lq)).mean(dim=0).sum()


def diff_pair(x1, x2):
    """functions.py:54-78: centre over batch, row-normalise by DETACHED L2 norm (+1e-6),
    mean of squared 128x128 Gram."""
    x1 = torch.nan_to_num(x1); x2 = torch.nan_to_num(x2)
    x1 = x1 - x1.mean(dim=0, keepdim=True)
    x2 = x2 - x2.mean(dim=0, keepdim=True)
    n1 = x1.norm(dim=1, keepdim=True).detach() + 1e-6
    n2 = x2.norm(dim=1, keepdim=True).detach() + 1e-6
    g = (x1 / n1).t() @ (x2 / n2)
    return (g * g).mean()


def diff_loss(o):
    """solver.py:422-441: the six (private,shared)/(private,private) pairs."""
    return (diff_pair(o.utt_private_t, o.utt_shared_t) + diff_pair(o.utt_private_v, o.utt_shared_v)
            + diff_pair(o.utt_private_a, o.utt_shared_a) + diff_pair(o.utt_private_a, o.utt_private_t)
            + diff_pair(o.utt_private_a, o.utt_private_v) + diff_pair(o.utt_private_t, o.utt_private_v))


def cmd_pair(x1, x2, n_moments=5):
    """functions.py:88-109: ||m1-m2||_2 + sum_{k=2..5} ||E[(x1-m1)^k]-E[(x2-m2)^k]||_2."""
    m1, m2 = x1.mean(0), x2.mean(0)
    s1, s2 = x1 - m1, x2 - m2
    tot = ((m1 - m2) ** 2).sum() ** 0.5
    for k in range(2, n_moments + 1):
        tot = tot + (((s1 ** k).mean(0) - (s2 ** k).mean(0)) ** 2).sum() ** 0.5
    return tot


def cmd_loss(o):
    """solver.py:409-420."""
    return (cmd_pair(o.utt_shared_t, o.utt_shared_v) + cmd_pair(o.utt_shared_t, o.utt_shared_a)
            + cmd_pair(o.utt_shared_a, o.utt_shared_v)) / 3.0


def recon_loss(o):
    """solver.py:443-449; note the target (utt_*_orig) is NOT detached."""
    return (((o.utt_t_recon - o.utt_t_orig) ** 2).mean() + ((o.utt_v_recon - o.utt_v_orig) ** 2).mean()
            + ((o.utt_a_recon - o.utt_a_orig) ** 2).mean()) / 3.0


def domain_loss(o):
    """solver.py:388-407: CE over cat(dom_t,dom_v,dom_a) with labels 0/1/2."""
    B = o.domain_label_t.shape[0]
    pred = torch.cat((o.domain_label_t, o.domain_label_v, o.domain_label_a), dim=0)
    lsm = pred - torch.logsumexp(pred, dim=1, keepdim=True)
    tgt = torch.arange(3).repeat_interleave(B)
    return -lsm[torch.arange(3 * B), tgt].mean()


def conf_loss(scores, tcp, emo):
    """solver.py:451-462.  Per class c: MSE_mean(tcp_c, emo_c*score_c)/nnz_c +
    CE(score_c, emo_c)/nnz_c where the CE is nn.CrossEntropyLoss on a 1-D input with a float
    (probability) target: -sum_b emo_b * log_softmax_over_batch(score)_b."""
    nnz = (emo != 0).sum(dim=0).to(scores.dtype)
    tcp_term = ((tcp - emo * scores) ** 2).mean(dim=0) / nnz
    lsm = scores - torch.logsumexp(scores, dim=0, keepdim=True)
    mcp_term = -(emo * lsm).sum(dim=0) / nnz
    return tcp_term.sum() + mcp_term.sum()


def all_losses(o, emo, cfg) -> SimpleNamespace:
    """solver.py:163-181."""
    L = SimpleNamespace()
    L.cls = cls_loss(o.scores, emo)
    L.diff = diff_loss(o)
    L.recon = recon_loss(o)
    L.conf = conf_loss(o.scores, o.tcp, emo)
    if cfg.use_cmd_sim:
        L.sim = cmd_loss(o)
    else:
        L.sim = domain_loss(o)
    L.total = L.cls + cfg.diff_weight * L.diff + cfg.sim_weight * L.sim + cfg.recon_weight * L.recon
    if cfg.use_confidNet:
        L.total = L.total + cfg.conf_weight * L.conf
    return L


# ----------------------------------------------------------------------------- training step
class AdamState:
    """Restated torch.optim.Adam (lr, betas (0.9,0.999), eps 1e-8, no weight decay, no amsgrad);
    parameters whose grad is None are skipped like torch does (solver.py:97-99,186)."""

    def __init__(self, P: Params, lr: float):
        self.lr, self.b1, self.b2, self.eps = lr, 0.9, 0.999, 1e-8
        self.m = {k: torch.zeros_like(v) for k, v in P.items()}
        self.v = {k: torch.zeros_like(v) for k, v in P.items()}
        self.t = {k: 0 for k in P}

    def step(self, P: Params, G: Dict[str, Optional[torch.Tensor]]):
        with torch.no_grad():
            seen = set()
            for k, p in P.items():
                g = G.get(k)
                if g is None:
                    continue
                if p.data_ptr() in seen:                  # an alias of a shared parameter (the one nn.PReLU slope): stepped once
                    continue
                seen.add(p.data_ptr())
                self.t[k] += 1
                t = self.t[k]
                self.m[k].mul_(self.b1).add_(g, alpha=1 - self.b1)
                self.v[k].mul_(self.b2).addcmul_(g, g, value=1 - self.b2)
                bc1 = 1 - self.b1 ** t
                bc2 = 1 - self.b2 ** t
                denom = (self.v[k].sqrt() / math.sqrt(bc2)).add_(self.eps)
                p.addcdiv_(self.m[k], denom, value=-self.lr / bc1)


def loss_and_grads(P: Params, cfg, batch, ffn=None):
    """One forward+backward (solver.py:139-183).  Returns (outputs, losses, grads) where grads
    has None for parameters outside the graph (sp_discriminator.*; confidence.* unless
    use_confidNet) exactly as autograd leaves them in the reference."""
    leaves = {k: p.detach().clone().requires_grad_(True) for k, p in P.items()}
    for k in leaves:                                      # the shared PReLU slope is ONE leaf under all its names
        if k.endswith("activation.weight"):
            leaves[k] = leaves[PRELU_KEY]
    o = forward(leaves, cfg, batch["t"], batch["v"], batch["a"], batch["l"], ffn)
    L = all_losses(o, batch["emo"], cfg)
    L.total.backward()
    grads = {k: (None if p.grad is None else p.grad.detach()) for k, p in leaves.items()}
    return o, L, grads


def train_step(P: Params, opt: AdamState, cfg, batch):
    """solver.py:139-186: fwd, losses, bwd, clip_grad_value_(clip), Adam.  Mutates P in place."""
    o, L, G = loss_and_grads(P, cfg, batch)
    G = {k: (None if g is None else g.clamp(-cfg.clip, cfg.clip)) for k, g in G.items()}
    opt.step(P, G)
    return o, L, G


# ----------------------------------------------------------------------------- nn.Module twin
class ModuleBaseline(nn.Module):
    """Stock-module CPU baseline (nn.LSTM / nn.TransformerEncoderLayer / torch.optim.Adam) with
    dropout ON, i.e. the shape of work the reference's CPU training loop performs per step.
    Used only by bench.py's cpu_baseline leg ("kind": "port").  Pinned: with the reference's parameters
    loaded (`load_reference_params`) and dropout off (`eval()`), its losses and gradients equal the
    reference-generated golden fixtures (tests/test_oracle_golden.py::test_module_baseline_*)."""

    _MODS = "tva"

    def reference_name_map(self):
        """{reference state_dict key (models.py:47-161): this module's parameter name}."""
        m = {"embed.weight": "embed.weight", "shared.shared_1.weight": "shared.0.weight", "shared.shared_1.bias": "shared.0.bias",
             "sp_discriminator.sp_discriminator_layer_1.weight": "sp.weight", "sp_discriminator.sp_discriminator_layer_1.bias": "sp.bias",
             "confidence.confidence_layer_1.weight": "confidence.0.weight", "confidence.confidence_layer_1.bias": "confidence.0.bias",
             "classifier.classifier_layer.weight": "classifier.0.weight", "classifier.classifier_layer.bias": "classifier.0.bias"}
        for i, c in enumerate(self._MODS):
            for layer, mine in ((1, "rnn1"), (2, "rnn2")):
                for sfx in ("", "_reverse"):
                    for w in ("weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0"):
                        m[f"{c}rnn{layer}.{w}{sfx}"] = f"{mine}.{i}.{w}{sfx}"
            for w in ("weight", "bias"):
                m[f"{c}layer_norm.{w}"] = f"ln.{i}.{w}"
                m[f"project_{c}.project_{c}.{w}"] = f"proj.{i}.0.{w}"
                m[f"project_{c}.project_{c}_layer_norm.{w}"] = f"proj.{i}.2.{w}"
                m[f"private_{c}.private_{c}_{3 if c == 'a' else 1}.{w}"] = f"private.{i}.0.{w}"
                m[f"recon_{c}.recon_{c}_1.{w}"] = f"recon.{i}.{w}"
        for k in ("self_attn.in_proj_weight", "self_attn.in_proj_bias", "self_attn.out_proj.weight", "self_attn.out_proj.bias",
                  "linear1.weight", "linear1.bias", "linear2.weight", "linear2.bias", "norm1.weight", "norm1.bias",
                  "norm2.weight", "norm2.bias"):
            m[f"transformer_encoder.layers.0.{k}"] = f"fuse.layers.0.{k}"
        return m

    def load_reference_params(self, P: Params):
        """Copies a reference-keyed parameter dict (synth_params / a reference checkpoint) into the stock modules."""
        mine = dict(self.named_parameters())
        nm = self.reference_name_map()
        assert sorted(nm.values()) == sorted(mine), "name map does not cover the module"
        with torch.no_grad():
            for ref, own in nm.items():
                assert mine[own].shape == P[ref].shape, (ref, own)
                mine[own].copy_(P[ref])
        return self

    def __init__(self, cfg):
        super().__init__()
        self.cfg = cfg
        dt, dv, da, hs = cfg.embedding_size, cfg.visual_size, cfg.acoustic_size, cfg.hidden_size
        self.embed = nn.Embedding(cfg.vocab_size, dt)
        self.lstm = getattr(cfg, "rnncell", "lstm") == "lstm"
        rnn = nn.LSTM if self.lstm else nn.GRU                         # models.py:39
        self.rnn1 = nn.ModuleList([rnn(d, d, bidirectional=True) for d in (dt, dv, da)])
        self.rnn2 = nn.ModuleList([rnn(2 * d, d, bidirectional=True) for d in (dt, dv, da)])
        self.ln = nn.ModuleList([nn.LayerNorm(2 * d) for d in (dt, dv, da)])
        self.proj = nn.ModuleList([nn.Sequential(nn.Linear(4 * d, hs), nn.LeakyReLU(), nn.LayerNorm(hs))
                                   for d in (dt, dv, da)])
        self.private = nn.ModuleList([nn.Sequential(nn.Linear(hs, hs), nn.Sigmoid()) for _ in range(3)])
        self.shared = nn.Sequential(nn.Linear(hs, hs), nn.Sigmoid())
        self.recon = nn.ModuleList([nn.Linear(hs, hs) for _ in range(3)])
        self.sp = nn.Linear(hs, 4)
        assert cfg.activation == "leakyrelu" and cfg.use_cmd_sim, "ModuleBaseline is the headline configuration only"
        self.fuse = nn.TransformerEncoder(nn.TransformerEncoderLayer(d_model=hs, nhead=NHEAD), num_layers=1,
                                          enable_nested_tensor=False)
        self.confidence = nn.Sequential(nn.Linear(6 * hs, 6), nn.Sigmoid())
        self.classifier = nn.Sequential(nn.Linear(6 * hs, cfg.num_classes), nn.Dropout(cfg.dropout), nn.Sigmoid())

    def forward(self, t, v, a, lengths):
        o = SimpleNamespace()
        xs = (self.embed(t), v, a)
        utt = []
        for i, x in enumerate(xs):
            pk = pack_padded_sequence(x, lengths, enforce_sorted=False)
            o1, h1 = self.rnn1[i](pk)
            pad, _ = pad_packed_sequence(o1)
            pk2 = pack_padded_sequence(self.ln[i](pad), lengths, enforce_sorted=False)
            _, h2 = self.rnn2[i](pk2)
            if self.lstm:
                h1, h2 = h1[0], h2[0]
            utt.append(torch.cat((h1, h2), dim=2).permute(1, 0, 2).reshape(x.shape[1], -1))
        for i, m in enumerate("tva"):
            orig = self.proj[i](utt[i])
            setattr(o, f"utt_{m}_orig", orig)
            setattr(o, f"utt_private_{m}", self.private[i](orig))
            setattr(o, f"utt_shared_{m}", self.shared(orig))
        _ = [self.sp(getattr(o, f"utt_private_{m}")) for m in "tva"]
        _ = self.sp((o.utt_shared_t + o.utt_shared_v + o.utt_shared_a) / 3.0)
        for i, m in enumerate("tva"):
            setattr(o, f"utt_{m}_recon", self.recon[i](getattr(o, f"utt_private_{m}") + getattr(o, f"utt_shared_{m}")))
        x = torch.stack((o.utt_private_t, o.utt_private_v, o.utt_private_a,
                         o.utt_shared_t, o.utt_shared_v, o.utt_shared_a), dim=0)
        hf = self.fuse(x)
        o.h = torch.cat(tuple(hf[i] for i in range(6)), dim=1)
        o.tcp = self.confidence(o.h)
        o.scores = self.classifier(o.h)
        for m in "tva":
            setattr(o, f"domain_label_{m}", None)
        return o


def baseline_train_steps(cfg, batch, steps: int, warmup: int = 1):
    """Times `steps` CPU training steps (fwd, six losses, bwd, clip, Adam, six .item() reads as in
    solver.py:139-193) of ModuleBaseline.  Returns seconds for the timed steps."""
    import time
    model = ModuleBaseline(cfg)
    for n, p in model.named_parameters():
        if "weight_hh" in n:
            nn.init.orthogonal_(p)
    model.train()
    opt = torch.optim.Adam(model.parameters(), lr=cfg.learning_rate)
    t0 = None
    for it in range(warmup + steps):
        if it == warmup:
            t0 = time.perf_counter()
        model.zero_grad()
        o = model(batch["t"], batch["v"], batch["a"], batch["l"])
        L = all_losses(o, batch["emo"], cfg)
        L.total.backward()
        torch.nn.utils.clip_grad_value_([p for p in model.parameters() if p.requires_grad], cfg.clip)
        opt.step()
        _ = [x.item() for x in (L.cls, L.diff, L.recon, L.total, L.sim, L.conf)]
    return time.perf_counter() - t0
