"""Diagnostic (GPU box): evaluation (forward-only) throughput through Solver.eval's model call at the bench shapes."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mmda_amd import make_config
from mmda_amd.solver import Solver
from mmda_amd.data import synth_batch
dev = torch.device("cuda:0")
cfg = make_config(vocab_size=20000, precision="bf16", device=str(dev), batch_size=32, seq_len=50, pretrained_emb=torch.randn(20000, 300))
solver = Solver(cfg, cfg, cfg, None, None, None, is_train=False).build()
m = solver.model; m.eval()
t, v, a, y, emo, lengths, *_ = synth_batch(cfg, 32, 50, seed=0, ragged=False, device=dev)
with torch.no_grad():
    for _ in range(20): m(t, v, a, lengths)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    N = 300
    for _ in range(N): m(t, v, a, lengths)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"forward-only (evaluation) pass: {1e3 * dt / N:.3f} ms/batch = {32 * N / dt:.0f} samples/s")
