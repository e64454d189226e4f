"""Diagnostic (GPU box): cycles per stage of the fused row-local backward stretch A (workgroup 0), from a few training steps."""
import ctypes, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mmda_amd import make_config, MISA, _lib
from mmda_amd.data import synth_batch
lib = ctypes.CDLL(_lib.LIB_PATH)
cfg = make_config(vocab_size=2000, precision="bf16", device="cuda:0")
m = MISA(cfg)
batch = synth_batch(cfg, 32, 50, seed=1, device="cuda:0")
dbg = torch.zeros(16, dtype=torch.int64, device="cuda:0")
for it in range(4):
    if it == 3: lib.mmda_debug_set_fused_stamps(ctypes.c_void_p(dbg.data_ptr()))
    t, v, a, y, emo, l = batch[:6]
    m.train_step(t, v, a, l, emo, 1e-4, 1.0)
torch.cuda.synchronize()
lib.mmda_debug_set_fused_stamps(None)
v = dbg.cpu().tolist()
names = ["LayerNorm 1 backward", "d_ctx GEMM", "attention backward", "d_x6 chain (6 rounds) + sigmoid'", "d_orig chain (4 rounds)", "projection LayerNorms"]
for i, n in enumerate(names):
    print(f"{n:40s} {(v[i + 1] - v[i]):8d} cycles = {(v[i + 1] - v[i]) / 100.0:7.2f} us at 100 MHz (if the counter is the 100 MHz one; else core clocks)" if v[i + 1] else n)
