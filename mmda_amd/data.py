"""Synthetic MOSEI-shaped batches with the reference's 10-tuple contract (reference data_loader.py:59-122).

Real CMU-MOSEI needs the mmsdk ETL and data that are not in this environment, so the loaders
here generate tensors of the same shapes, dtypes and ordering rules: batch sorted by length descending, time-major
``pad_sequence`` layout, PAD id 1 past each sample's length (create_dataset.py:25-27: <unk>=0, <pad>=1), emotion labels
binarised to {0,1} float32, CPU int64 ``lengths``.  The three BERT tensors are returned as zeros of shape (B, T+2):
they are ignored when use_bert=False.
"""
from __future__ import annotations

import torch

PAD, UNK = 1, 0


def synth_batch(config, B: int, T: int, seed: int, ragged: bool = False, device="cpu"):
    g = torch.Generator().manual_seed(1234 + seed)
    V = len(config.word2id)
    if ragged:
        lengths = torch.sort(torch.randint(1, T + 1, (B,), generator=g), descending=True).values
        lengths[0] = T
    else:
        lengths = torch.full((B,), T, dtype=torch.int64)
    t = torch.randint(2, V, (T, B), generator=g)
    v = torch.randn(T, B, config.visual_size, generator=g)
    a = torch.randn(T, B, config.acoustic_size, generator=g)
    mask = torch.arange(T).unsqueeze(1) >= lengths.unsqueeze(0)          # (T,B) True on padding
    t[mask] = PAD
    v[mask] = 0.0
    a[mask] = 0.0
    emo = (torch.rand(B, 6, generator=g) > 0.6).float()
    for c in range(6):                    # every class >= 1 positive so conf-loss' /nnz is finite (solver.py:459)
        if emo[:, c].sum() == 0:
            emo[c % B, c] = 1.0
    y = torch.randn(B, generator=g)
    bert = torch.zeros(B, T + 2, dtype=torch.int64)
    ids = [f"synthetic_{seed}_{i}" for i in range(B)]
    dev = torch.device(device)
    if dev.type != "cpu":
        t, v, a, y, emo = (x.to(dev) for x in (t, v, a, y, emo))
    return t, v, a, y, emo, lengths, bert, bert, bert, ids


class SyntheticLoader:
    """Iterable of pre-generated batches (a stand-in for DataLoader(MSADataset, collate_fn))."""

    def __init__(self, config, n_batches: int, batch_size: int, seq_len: int, seed: int = 0, ragged: bool = True, device="cpu"):
        self.batches = [synth_batch(config, batch_size, seq_len, seed * 1000 + i, ragged, device) for i in range(n_batches)]
        self.dataset = self
        self.batch_size = batch_size

    def __iter__(self):
        return iter(self.batches)

    def __len__(self):
        return len(self.batches)


def get_loader(config, shuffle=True, n_batches=8, seed=0, ragged=True, device="cpu"):
    """Same name as the reference's factory (data_loader.py:50); synthetic data."""
    config.data_len = n_batches * config.batch_size
    return SyntheticLoader(config, n_batches, config.batch_size, getattr(config, "seq_len", 50), seed, ragged, device)
