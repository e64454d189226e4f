"""Diagnostic (GPU box): training rate when every batch starts in pageable HOST memory (the reference's DataLoader hand-off) and
reaches the device through mmda_amd.data.DevicePrefetcher (page-lock + async H2D on a copy stream, double buffered).
Never the benchmark's `value` (that one is HBM-resident); quoted in DESIGN.md section 5 as the PCIe-inclusive rate."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mmda_amd import make_config
from mmda_amd.solver import Solver
from mmda_amd.data import synth_batch, DevicePrefetcher
dev = torch.device("cuda:0")
cfg = make_config(vocab_size=20000, precision="bf16", device=str(dev), batch_size=32, seq_len=50, pretrained_emb=torch.randn(20000, 300))
solver = Solver(cfg, cfg, cfg, None, None, None, is_train=True).build()
m = solver.model; m.train()
host = [synth_batch(cfg, 32, 50, seed=i, ragged=False, device="cpu") for i in range(64)]     # distinct pageable host batches

class Cycle:
    def __init__(self, n): self.n = n
    def __len__(self): return self.n
    def __iter__(self):
        for i in range(self.n):
            yield host[i % len(host)]                 # what a DataLoader hands over: already-built pageable host tensors

def run(n, prefetch):
    src = DevicePrefetcher(Cycle(n), dev) if prefetch else Cycle(n)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for (t, v, a, y, emo, l, *_) in src:
        if not prefetch:
            t, v, a, emo = (x.to(dev) for x in (t, v, a, emo))
        m.train_step(t, v, a, l, emo, lr=1e-4, clip=1.0)
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n
run(30, True)
for pf in (True, False):
    dt = run(300, pf)
    print(f"host-resident input, {'DevicePrefetcher' if pf else 'synchronous .to(device)'}: {1e3 * dt:.3f} ms/step = {32 / dt:.0f} samples/s")

import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
run(100, True)
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(12)
