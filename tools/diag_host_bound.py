"""Diagnostic (GPU box): is the training loop host-bound?  Times the host-side enqueue of N steps (no sync) against the
wall time including the final sync."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mmda_amd import make_config
from mmda_amd.solver import Solver
from mmda_amd.data import synth_batch

dev = torch.device("cuda:0")
torch.manual_seed(0)
cfg = make_config(vocab_size=20000, precision="bf16", device=str(dev), batch_size=32, seq_len=50, use_confidNet=False,
                  pretrained_emb=torch.randn(20000, 300))
solver = Solver(cfg, cfg, cfg, None, None, None, is_train=True).build()
model = solver.model
model.train()
t, v, a, y, emo, lengths, *_ = synth_batch(cfg, 32, 50, seed=0, ragged=False, device=dev)
for _ in range(20):
    model.train_step(t, v, a, lengths, emo, lr=1e-4, clip=1.0)
torch.cuda.synchronize()
for N in (3, 10, 30, 200):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(N):
        model.train_step(t, v, a, lengths, emo, lr=1e-4, clip=1.0)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    print(f"N={N}: host enqueue {1e3 * (t1 - t0) / N:.3f} ms/step; wall with sync {1e3 * (t2 - t0) / N:.3f} ms/step; drain after loop {1e3 * (t2 - t1):.2f} ms")
import cProfile, pstats
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(5):
    model.train_step(t, v, a, lengths, emo, lr=1e-4, clip=1.0)
pr.disable()
torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("cumulative").print_stats(14)
