"""bf16-emulating CPU oracle for the recurrent encoders.  TEST INFRASTRUCTURE ONLY (see misa_oracle.py's header).

``misa_oracle`` is the exact fp32 restatement of the reference (``src/models.py:48-55,163-180``: ``nn.LSTM`` / ``nn.GRU`` on
packed sequences).  The product's bf16 mode keeps fp32 accumulators, fp32 cell state and an fp32 stash but feeds the matrix
cores bf16 operands; this file restates the SAME recurrences as explicit loops (forward and hand-derived BPTT) that round
to bf16 (round-to-nearest-even, ``Tensor.bfloat16()``) at exactly the points the HIP kernels round at, and nowhere else:

  forward   pre   = bf16(x) . bf16(W_ih)^T + b_ih + b_hh                (time-batched input GEMM, gemm_bf16.hip)
            gates = pre[t] + bf16(h_{t-1}) . bf16(W_hh)^T               (lstm_cluster.hip / lstm.hip; h, c themselves stay fp32)
  backward  dG_t in fp32 from the fp32 stash, then                      (lstm_bwd_wave_kernel)
            dh_{t-1} = sum over 16-unit hidden tiles of bf16( bf16(dG_t)[tile's gate rows] . bf16(W_hh)[those rows] )
                       (``tile_partials=32``: over 32-unit tile PAIRS -- the large-batch form of the kernel adds two tiles'
                        fp32 products in LDS before it rounds)
                       -- every producer wave publishes its partial as bf16; ``tile_partials=False`` gives the streaming
                          kernel's single fp32 sum (lstm.hip)
            dX = bf16(dG) . bf16(W_ih);  dW_ih = bf16(dG)^T . bf16(x);  dW_hh = bf16(dG)^T . bf16(hseq shifted);
            db_ih = db_hh = column sums of bf16(dG)                     (bias gradient = virtual ones-column of the bf16 GEMM)

With ``rounding=False`` every q() is the identity and the loops are plain fp32 BPTT: ``tests/test_bf16_emul_cpu.py`` checks
that form against autograd through ``nn.LSTM`` / ``nn.GRU`` (i.e. against misa_oracle, which the golden fixtures pin), so the
hand-written backward is itself pinned; the rounded form then differs from it only by the q() calls listed above.
What is NOT emulated: the order of fp32 additions inside a dot product (MFMA k-order, split-K atomics) and the kernels'
v_exp/v_rcp activations (1 ulp) -- both at the 1e-6 level, far below the 1e-2 bound the tests put on the bf16 path.
"""
from __future__ import annotations

from types import SimpleNamespace
from typing import Dict

import torch
import torch.nn.functional as F

from . import misa_oracle as orc

TILE = 16          # hidden units per producer wave of the backward exchange (lstm_cluster.hip: one wave = one 16-unit tile)


def _q(x: torch.Tensor, on: bool) -> torch.Tensor:
    return x.bfloat16().float() if on else x


def _tile_partial_sum(dg: torch.Tensor, w_hh_q: torch.Tensor, ng: int, H: int, rounding: bool, width: int = TILE) -> torch.Tensor:
    """dh = dG (B, ng*H) . W_hh (ng*H, H), summed the way lstm_bwd_wave_kernel sums it: the product over the gate rows of each
    16-unit hidden tile is rounded to bf16 (the producer wave's published partial), the partials are added in fp32.
    width = 32: the PAIR form of the kernel (four waves per block, B > 128): two neighbouring 16-unit tiles add their fp32 products
    in LDS before the rounding, so one bf16 partial covers 32 hidden units."""
    B = dg.shape[0]
    TILE = width                                             # (shadows the module constant on purpose: same code, wider tiles)
    nt = (H + TILE - 1) // TILE
    Hp = nt * TILE
    d3 = dg.view(B, ng, H)
    w3 = w_hh_q.view(ng, H, H)
    if Hp != H:
        d3 = F.pad(d3, (0, Hp - H))
        w3 = F.pad(w3, (0, 0, 0, Hp - H))
    dt = d3.view(B, ng, nt, TILE).permute(2, 0, 1, 3).reshape(nt, B, ng * TILE)
    wt = w3.view(ng, nt, TILE, H).permute(1, 0, 2, 3).reshape(nt, ng * TILE, H)
    part = _q(torch.bmm(dt, wt), rounding)                   # (nt, B, H)
    out = part[0].clone()
    for p in range(1, nt):                                    # producer 0, 1, 2, ... like the consumer's gather loop
        out = out + part[p]
    return out


class _BiRnn(torch.autograd.Function):
    """One bidirectional LSTM / GRU layer over a padded (T, B, D) input with packed-sequence semantics (a sample stops at its
    own length; the reverse direction starts at len_b - 1; outputs are zero at padded positions), forward and BPTT as explicit
    loops.  Parameters in torch's layout (gate order i,f,g,o / r,z,n).  Returns (out (T,B,2H), h_n (2,B,H))."""

    @staticmethod
    def forward(ctx, x, lengths, cell, rounding, tile_partials, *params):
        T, B, D = x.shape
        w_ih = [params[0], params[4]]; w_hh = [params[1], params[5]]
        b_ih = [params[2], params[6]]; b_hh = [params[3], params[7]]
        H = w_hh[0].shape[1]
        ng = 4 if cell == "lstm" else 3
        xq = _q(x, rounding)
        out = x.new_zeros(T, B, 2 * H)
        hn = x.new_zeros(2, B, H)
        stash = []
        lens = lengths.to(torch.int64)
        for d in range(2):
            wq = _q(w_hh[d], rounding)
            if cell == "lstm":
                pre = xq.reshape(T * B, D) @ _q(w_ih[d], rounding).t() + b_ih[d] + b_hh[d]
            else:       # the hidden-side bias of the candidate gate sits INSIDE r * (.) (nn.GRU): it stays with the recurrent product
                bh = b_hh[d].clone(); bh[2 * H:] = 0
                pre = xq.reshape(T * B, D) @ _q(w_ih[d], rounding).t() + b_ih[d] + bh
            pre = pre.view(T, B, ng * H)
            h = x.new_zeros(B, H); c = x.new_zeros(B, H)
            G = x.new_zeros(T, B, 4 * H)                       # activated gates (LSTM i,f,g,o | GRU r,z,n,q)
            C = x.new_zeros(T, B, H)                           # LSTM: c_t ; GRU: unused
            Hprev = x.new_zeros(T, B, H)                       # h_{t-1} in processing order (fp32)
            for t in (range(T - 1, -1, -1) if d else range(T)):
                m = (t < lens).to(x.dtype).unsqueeze(1)
                rec = _q(h, rounding) @ wq.t()                 # (B, ng*H)
                Hprev[t] = h
                if cell == "lstm":
                    g = pre[t] + rec
                    i, f, gg, o = g.chunk(4, dim=1)
                    i, f, gg, o = torch.sigmoid(i), torch.sigmoid(f), torch.tanh(gg), torch.sigmoid(o)
                    c2 = f * c + i * gg
                    h2 = o * torch.tanh(c2)
                    G[t] = torch.cat((i, f, gg, o), 1) * m
                    C[t] = c2 * m
                    c = m * c2 + (1 - m) * c
                else:
                    r = torch.sigmoid(pre[t][:, :H] + rec[:, :H])
                    z = torch.sigmoid(pre[t][:, H:2 * H] + rec[:, H:2 * H])
                    qh = rec[:, 2 * H:] + b_hh[d][2 * H:]
                    n = torch.tanh(pre[t][:, 2 * H:] + r * qh)
                    h2 = (1 - z) * n + z * h
                    G[t] = torch.cat((r, z, n, qh), 1) * m
                h = m * h2 + (1 - m) * h
                out[t, :, d * H:(d + 1) * H] = m * h2
            hn[d] = h
            stash.append((G, C, Hprev))
        ctx.save_for_backward(x, lens, out, *params, *[t_ for s in stash for t_ in s])
        ctx.meta = (cell, rounding, tile_partials, H)
        return out, hn

    @staticmethod
    def backward(ctx, d_out, d_hn):
        cell, rounding, tile_partials, H = ctx.meta
        saved = ctx.saved_tensors
        x, lens, out = saved[0], saved[1], saved[2]
        params = saved[3:11]
        st = saved[11:]
        T, B, D = x.shape
        ng = 4 if cell == "lstm" else 3
        xq = _q(x, rounding)
        dx = torch.zeros_like(x)
        grads = [None] * 8
        for d in range(2):
            G, C, Hprev = st[3 * d], st[3 * d + 1], st[3 * d + 2]
            w_ih, w_hh = params[4 * d], params[4 * d + 1]
            wq = _q(w_hh, rounding)
            dG = x.new_zeros(T, B, ng * H)                     # gradient w.r.t. the pre-activations, torch gate order
            dQ = x.new_zeros(T, B, H) if cell == "gru" else None   # GRU: gradient w.r.t. q = h W_hn^T + b_hn
            dh_rec = x.new_zeros(B, H); dc = x.new_zeros(B, H)
            for t in (range(T) if d else range(T - 1, -1, -1)):        # reverse of the processing order
                act = (t < lens).to(x.dtype).unsqueeze(1)
                fin = (torch.full_like(lens, t) == 0) if d else (lens - 1 == t)     # the step that produced h_n
                fin = fin.to(x.dtype).unsqueeze(1) * act
                dh = dh_rec + d_out[t, :, d * H:(d + 1) * H] * act + fin * d_hn[d]
                if cell == "lstm":
                    i, f, gg, o = G[t].chunk(4, dim=1)
                    # previous cell state in processing order
                    tp = t + 1 if d else t - 1
                    cp = C[tp] * ((tp < lens) & (tp >= 0)).to(x.dtype).unsqueeze(1) if 0 <= tp < T else torch.zeros_like(dc)
                    tc = torch.tanh(C[t])
                    dct = dc + dh * o * (1 - tc * tc)
                    di = dct * gg * i * (1 - i) * act
                    df = dct * cp * f * (1 - f) * act
                    dg_ = dct * i * (1 - gg * gg) * act
                    do = dh * tc * o * (1 - o) * act
                    dc = act * (dct * f) + (1 - act) * dc
                    dG[t] = torch.cat((di, df, dg_, do), 1)
                    dgq = _q(dG[t], rounding)
                    if tile_partials and rounding:
                        dh_rec = _tile_partial_sum(dgq, wq, 4, H, True, TILE if tile_partials is True else int(tile_partials))
                    else:
                        dh_rec = dgq @ wq
                else:
                    r, z, n, qh = G[t].chunk(4, dim=1)
                    hp = Hprev[t]
                    dh = dh + dc
                    dpn = dh * (1 - z) * (1 - n * n) * act
                    dr = dpn * qh * r * (1 - r)
                    dz = dh * (hp - n) * z * (1 - z) * act
                    dq = dpn * r
                    dc = act * (dh * z) + (1 - act) * dc       # the direct path h_{t-1} -> h_t (carried like the LSTM's dc)
                    dG[t] = torch.cat((dr, dz, dpn), 1)
                    dQ[t] = dq
                    # recurrent product: rows r, z of W_hh see d(pre_r), d(pre_z); rows n see dq
                    dgh = _q(torch.cat((dr, dz, dq), 1), rounding)
                    if tile_partials and rounding:
                        dh_rec = _tile_partial_sum(dgh, wq, 3, H, True, TILE if tile_partials is True else int(tile_partials))
                    else:
                        dh_rec = dgh @ wq
            dGq = _q(dG, rounding).reshape(T * B, ng * H)
            dx = dx + (dGq @ _q(w_ih, rounding)).view(T, B, D)
            g_wih = dGq.t() @ xq.reshape(T * B, D)
            # h_{t-1} in processing order as the kernels read it: the hseq tensor (zero at padded positions) shifted by one step
            hs = _q(out[:, :, d * H:(d + 1) * H], rounding)
            hshift = torch.zeros_like(hs)
            if d:
                hshift[:-1] = hs[1:]
            else:
                hshift[1:] = hs[:-1]
            if cell == "lstm":
                g_whh = dGq.t() @ hshift.reshape(T * B, H)
                g_bih = dGq.sum(0)
                g_bhh = g_bih.clone()
            else:
                dHq = _q(torch.cat((dG[:, :, :2 * H], dQ), 2), rounding).reshape(T * B, 3 * H)
                g_whh = dHq.t() @ hshift.reshape(T * B, H)
                g_bih = dGq.sum(0)
                g_bhh = dHq.sum(0)
            grads[4 * d], grads[4 * d + 1], grads[4 * d + 2], grads[4 * d + 3] = g_wih, g_whh, g_bih, g_bhh
        return (dx, None, None, None, None) + tuple(grads)


_RNN_KEYS = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0", "weight_ih_l0_reverse", "weight_hh_l0_reverse",
             "bias_ih_l0_reverse", "bias_hh_l0_reverse"]


def birnn(x, lengths, P: Dict[str, torch.Tensor], prefix: str, cell: str, rounding: bool, tile_partials: bool = True):
    """(out (T,B,2H) zero at padded positions, h_n (2,B,H)) of one bidirectional layer ``prefix`` (state_dict names)."""
    return _BiRnn.apply(x, lengths, cell, rounding, tile_partials, *[P[f"{prefix}.{k}"] for k in _RNN_KEYS])


def encode_modality(x, lengths, P, m: str, d: int, cell: str, rounding: bool, tile_partials: bool = True):
    """misa_oracle.encode_modality (reference models.py:163-180, 203) on the explicit-loop layers."""
    out1, h1 = birnn(x, lengths, P, f"{m}rnn1", cell, rounding, tile_partials)
    # (pad_packed_sequence + LayerNorm turns padded positions into beta; layer 2 re-packs them away, here they are masked)
    normed = F.layer_norm(out1, (2 * d,), P[f"{m}layer_norm.weight"], P[f"{m}layer_norm.bias"], orc.LN_EPS)
    _, h2 = birnn(normed, lengths, P, f"{m}rnn2", cell, rounding, tile_partials)
    B = x.shape[1]
    return torch.cat((h1, h2), dim=2).permute(1, 0, 2).reshape(B, 4 * d)


def forward(P, cfg, t, v, a, lengths, rounding: bool = True, tile_partials: bool = True) -> SimpleNamespace:
    """misa_oracle.forward with the three encoders on the bf16-emulating loops; the fusion block, heads and losses are the exact
    fp32 ones (the product keeps them on the exact f32 path in bf16 mode as well)."""
    cell = "lstm" if getattr(cfg, "rnncell", "lstm") == "lstm" else "gru"
    lengths = lengths.cpu()
    emb = P["embed.weight"][t]
    utt = {"t": encode_modality(emb, lengths, P, "t", cfg.embedding_size, cell, rounding, tile_partials),
           "v": encode_modality(v, lengths, P, "v", cfg.visual_size, cell, rounding, tile_partials),
           "a": encode_modality(a, lengths, P, "a", cfg.acoustic_size, cell, rounding, tile_partials)}
    return orc.fusion_from_utterances(P, cfg, utt)


def loss_and_grads(P, cfg, batch, rounding: bool = True, tile_partials: bool = True):
    """misa_oracle.loss_and_grads through the emulating encoders."""
    leaves = {k: p.detach().clone().requires_grad_(True) for k, p in P.items()}
    for k in leaves:                                      # the shared PReLU slope is ONE leaf under all its names
        if k.endswith("activation.weight"):
            leaves[k] = leaves[orc.PRELU_KEY]
    o = forward(leaves, cfg, batch["t"], batch["v"], batch["a"], batch["l"], rounding, tile_partials)
    L = orc.all_losses(o, batch["emo"], cfg)
    L.total.backward()
    grads = {k: (None if p.grad is None else p.grad.detach()) for k, p in leaves.items()}
    return o, L, grads
