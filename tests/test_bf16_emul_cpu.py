"""CPU: pins oracle/bf16_emul.py.  With rounding off, its explicit-loop encoders (forward and hand-derived BPTT) must reproduce
misa_oracle (autograd through nn.LSTM / nn.GRU, itself pinned by the reference-generated golden fixtures) on outputs, losses
and every gradient; with rounding on, it must stay inside the bf16 weight-quantisation floor measured by
test_bf16_floor_cpu.py -- it is the same computation plus q() calls."""
import pytest
import torch

from oracle import misa_oracle as orc
from oracle import bf16_emul as emu
from golden_util import batch_of, load_case


def _grad_errs(G, Gref):
    worst = ("", 0.0)
    for k, ref in Gref.items():
        if ref is None:
            assert G[k] is None, k
            continue
        if k.endswith("in_proj_bias"):
            continue
        e = float((G[k] - ref).norm() / ref.norm().clamp_min(1e-30))
        if e > worst[1]:
            worst = (k, e)
    return worst


@pytest.mark.parametrize("name", ["tiny_cmd_ragged", "tiny_gru_ragged", "tiny_adv_confid_full", "real_b8_t12_ragged",
                                  "real_gru_b8_t12_ragged"])
def test_unrounded_loops_equal_the_fp32_oracle_and_the_golden_gradients(name):
    z, meta, cfg = load_case(name)
    P = orc.synth_params(cfg, meta["seed"])
    batch = batch_of(z)
    o, L, G = orc.loss_and_grads(P, cfg, batch)
    o2, L2, G2 = emu.loss_and_grads(P, cfg, batch, rounding=False)
    assert float((o2.scores - o.scores).abs().max()) < 2e-6
    for k in ("cls", "diff", "sim", "recon", "conf", "total"):
        assert abs(float(getattr(L2, k)) - float(getattr(L, k))) <= 2e-5 * abs(float(getattr(L, k))) + 1e-7, k
    k, e = _grad_errs(G2, G)
    assert e < 2e-4, (k, e)
    # and straight against the reference-produced gradients of the fixture
    if meta["full_tensors"]:
        for k2 in G2:
            if G2[k2] is None or k2.endswith("in_proj_bias"):
                continue
            gold = torch.from_numpy(z["grad::" + k2])
            assert float((G2[k2] - gold).norm() / gold.norm().clamp_min(1e-30)) < 2e-4, k2


@pytest.mark.parametrize("name,hi", [("real_b8_t12_ragged", 1.2e-1), ("real_gru_b8_t12_ragged", 2e-1)])
def test_rounded_loops_stay_within_the_quantisation_floor(name, hi):
    z, meta, cfg = load_case(name)
    P = orc.synth_params(cfg, meta["seed"])
    batch = batch_of(z)
    o, L, G = orc.loss_and_grads(P, cfg, batch)
    oq, Lq, Gq = emu.loss_and_grads(P, cfg, batch, rounding=True, tile_partials=True)
    assert float((oq.scores - o.scores).abs().max()) < 1e-2
    k, e = _grad_errs(Gq, G)
    assert 1e-3 < e < hi, (k, e)
    # the per-tile rounding of the partial dh sums is a second-order effect
    os_, Ls, Gs = emu.loss_and_grads(P, cfg, batch, rounding=True, tile_partials=False)
    k, e2 = _grad_errs(Gs, Gq)
    assert e2 < 3e-2, (k, e2)
