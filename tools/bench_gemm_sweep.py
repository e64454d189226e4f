"""Diagnostic (GPU box): the bf16-operand GEMM on large square-ish problems (asymptotic k-loop rate) and a K sweep."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mmda_amd import ops
d = torch.device("cuda:0")
for M, N, K in [(4096, 4096, 4096), (8192, 8192, 1024), (12800, 2400, 4800), (12800, 2400, 64), (12800, 2400, 128), (12800, 2400, 1200), (2048, 2048, 8192)]:
    Ab = torch.randn(M, K, device=d).bfloat16(); Bb = torch.randn(N, K, device=d).bfloat16()
    out = torch.zeros(M, N, device=d)
    prob = [dict(A=Ab, B=Bb, K=K, out=out)]
    for _ in range(3):
        ops.gemm_bf16_grouped(prob)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        ops.gemm_bf16_grouped(prob)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 10
    print(f"M={M:6d} N={N:5d} K={K:6d}: {us:9.1f} us  {2.0*M*N*K/us/1e6:7.1f} TFLOP/s  C-write {M*N*4/us/1e6:6.2f} TB/s")
