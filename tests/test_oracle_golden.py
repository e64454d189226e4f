"""Pins the CPU oracle (oracle/misa_oracle.py) to golden vectors produced by the reference's own
models.MISA / DiffLoss / CMD (tests/golden/gen_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from oracle import misa_oracle as orc
from golden_util import SIDE, batch_of, case_names, load_case, sample_idx

RTOL, ATOL = 2e-5, 2e-6     # fp32 torch-vs-torch; same kernels for LSTM, closed forms elsewhere


@pytest.mark.parametrize("name", case_names(large=True))
def test_oracle_forward_losses_grads(name):
    z, meta, cfg = load_case(name)
    P = orc.synth_params(cfg, meta["seed"])
    if meta["full_tensors"]:
        for k in P:
            np.testing.assert_array_equal(P[k].numpy(), z["param0::" + k])
    batch = batch_of(z)
    b2 = orc.synth_batch(cfg, meta["B"], meta["T"], meta["seed"], meta["ragged"])
    for k in batch:
        assert torch.equal(batch[k], b2[k]), f"synthetic batch generator drifted for {k}"
    o, L, G = orc.loss_and_grads(P, cfg, batch)
    np.testing.assert_allclose(o.scores.detach().numpy(), z["out::scores"], rtol=RTOL, atol=ATOL)
    np.testing.assert_array_equal(o.labels.detach().numpy(), z["out::labels"])
    np.testing.assert_allclose(o.tcp.detach().numpy(), z["out::tcp"], rtol=RTOL, atol=ATOL)
    for s in SIDE:
        np.testing.assert_allclose(getattr(o, s).detach().numpy(), z["out::" + s], rtol=RTOL, atol=ATOL, err_msg=s)
    if not cfg.use_cmd_sim:
        for m in "tva":
            np.testing.assert_allclose(getattr(o, f"domain_label_{m}").detach().numpy(),
                                       z[f"out::domain_label_{m}"], rtol=RTOL, atol=ATOL)
    for k in ("cls", "diff", "recon", "sim", "conf", "total"):
        np.testing.assert_allclose(getattr(L, k).item(), float(z["loss::" + k]), rtol=1e-5, atol=1e-6, err_msg=k)
    none = set(meta["none_grads"])
    for k, g in G.items():
        if k != orc.PRELU_KEY and k.endswith("activation.weight"):
            assert g is G[orc.PRELU_KEY] or torch.equal(g, G[orc.PRELU_KEY])      # an alias of the one shared nn.PReLU slope
            continue
        if k in none:
            assert g is None, k
            continue
        assert g is not None, k
        gn = g.numpy()
        scale = max(float(z["gnorm::" + k]), 1e-12)
        if meta["full_tensors"]:
            ref = z["grad::" + k]
            assert np.abs(gn - ref).max() <= 3e-5 * max(np.abs(ref).max(), 1e-6) + 1e-7, k
        else:
            ref = z["gsample::" + k]
            got = gn.ravel()[sample_idx(gn.size)]
            assert np.abs(got - ref).max() <= 3e-5 * max(np.abs(ref).max(), 1e-6) + 1e-7, k
        assert abs(np.sqrt((gn.astype(np.float64) ** 2).sum()) - scale) <= 1e-4 * scale + 1e-9, k


@pytest.mark.parametrize("name", case_names())
def test_oracle_three_adam_steps(name):
    z, meta, cfg = load_case(name)
    P = orc.synth_params(cfg, meta["seed"])
    opt = orc.AdamState(P, cfg.learning_rate)
    n = meta["steps"]
    for s in range(n):
        batch = orc.synth_batch(cfg, meta["B"], meta["T"], meta["seed"] + s, meta["ragged"])
        _, L, _ = orc.train_step(P, opt, cfg, batch)
        np.testing.assert_allclose(L.total.item(), float(z[f"loss_step{s}::total"]), rtol=2e-5)
    for k, p in P.items():
        a = p.numpy()
        if meta["full_tensors"]:
            ref = z[f"param{n}::" + k]
            got = a
        else:
            ref = z[f"psample{n}::" + k]
            got = a.ravel()[sample_idx(a.size)]
        # Adam divides by sqrt(v): an element whose |grad| is at fp32 rounding-noise level can move by a
        # visibly different fraction of lr.  Every element moves <= steps*lr = 3e-4 in total; we
        # require agreement to 1e-5 absolute (3 % of that bound) and 99.9 % of elements to 2e-6.
        if k.endswith("self_attn.in_proj_bias"):
            # d(loss)/d(key bias) is exactly 0 in real arithmetic (softmax is shift-invariant along
            # keys); in fp32 it is rounding noise that Adam normalises to +-lr.  Not comparable.
            hs = cfg.hidden_size
            keep = np.ones(3 * hs, bool); keep[hs:2 * hs] = False
            keep = keep[sample_idx(3 * hs)] if not meta["full_tensors"] else keep
            got, ref = got[keep], ref[keep]
        np.testing.assert_allclose(got, ref, rtol=0, atol=1e-5, err_msg=k)
        assert (np.abs(got - ref) <= 2e-6).mean() >= 0.999, k


def test_loop_lstm_matches_packed_lstm():
    """The explicit masked-loop LSTM (the semantics the HIP kernel implements) == nn.LSTM on a
    packed sequence, for ragged lengths, both directions."""
    torch.manual_seed(0)
    T, B, D, H = 9, 5, 7, 6
    x = torch.randn(T, B, D)
    lengths = torch.tensor([9, 7, 4, 2, 1])
    rnn = torch.nn.LSTM(D, H, bidirectional=True)
    pk = torch.nn.utils.rnn.pack_padded_sequence(x, lengths, enforce_sorted=False)
    out, (hn, _) = rnn(pk)
    pad, _ = torch.nn.utils.rnn.pad_packed_sequence(out, total_length=T)
    for d, sfx in enumerate(("", "_reverse")):
        o, h = orc.lstm_dir_loop(x, lengths, getattr(rnn, "weight_ih_l0" + sfx), getattr(rnn, "weight_hh_l0" + sfx),
                                 getattr(rnn, "bias_ih_l0" + sfx), getattr(rnn, "bias_hh_l0" + sfx), reverse=bool(d))
        torch.testing.assert_close(o, pad[:, :, d * H:(d + 1) * H], rtol=1e-5, atol=1e-6)
        torch.testing.assert_close(h, hn[d], rtol=1e-5, atol=1e-6)


@pytest.mark.parametrize("name", ["real_b8_t12_ragged", "real_b32_t50_full"])
def test_module_baseline_matches_reference_fixtures(name):
    """bench.py's cpu_baseline times oracle.ModuleBaseline (stock nn.LSTM / nn.TransformerEncoderLayer modules).
    It is a checked port: with the reference's parameters and dropout off, its outputs, six losses and every gradient equal what
    the reference's own models.MISA produced (the golden fixtures)."""
    z, meta, cfg = load_case(name)
    assert cfg.rnncell == "lstm" and cfg.use_cmd_sim and not cfg.use_confidNet
    P = orc.synth_params(cfg, meta["seed"])
    m = orc.ModuleBaseline(cfg).load_reference_params(P)
    m.eval()                                                    # dropout off, like the generator (gen_golden.py)
    batch = batch_of(z)
    o = m(batch["t"], batch["v"], batch["a"], batch["l"])
    L = orc.all_losses(o, batch["emo"], cfg)
    L.total.backward()
    np.testing.assert_allclose(o.scores.detach().numpy(), z["out::scores"], rtol=RTOL, atol=ATOL)
    for s in SIDE:
        np.testing.assert_allclose(getattr(o, s).detach().numpy(), z["out::" + s], rtol=RTOL, atol=ATOL, err_msg=s)
    for k in ("cls", "diff", "recon", "sim", "conf", "total"):
        np.testing.assert_allclose(getattr(L, k).item(), float(z["loss::" + k]), rtol=1e-5, atol=1e-6, err_msg=k)
    mine = dict(m.named_parameters())
    none = set(meta["none_grads"])
    for ref, own in m.reference_name_map().items():
        g = mine[own].grad
        if ref in none:
            assert g is None or float(g.abs().max()) == 0.0, ref
            continue
        assert g is not None, ref
        gn = g.numpy()
        if ref.endswith("self_attn.in_proj_bias"):
            continue                                            # the key-bias slice is rounding noise (softmax shift invariance)
        exp = z["grad::" + ref] if meta["full_tensors"] else z["gsample::" + ref]
        got = gn if meta["full_tensors"] else gn.ravel()[sample_idx(gn.size)]
        assert np.abs(got - exp).max() <= 5e-5 * max(np.abs(exp).max(), 1e-6) + 1e-7, ref
        scale = max(float(z["gnorm::" + ref]), 1e-12)
        assert abs(np.sqrt((gn.astype(np.float64) ** 2).sum()) - scale) <= 1e-4 * scale + 1e-9, ref
