"""Diagnostic (GPU box): the bf16-operand GEMM on the text-encoder problems of a B=256, T=50 step, one problem per launch."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mmda_amd import ops
d = torch.device("cuda:0")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
R = 50 * B
shapes = [("fwd L1", R, 2400, 300, False), ("fwd L2", R, 2400, 600, False), ("dX L2", R, 600, 2400, False), ("dX L1", R, 300, 2400, False),
          ("dW_ih L2", 2400, 600, R, True), ("dW_ih L1", 2400, 300, R, True), ("dW_hh", 1200, 300, R - B, True)]
for name, M, N, K, acc in shapes:
    Kp = (K + 7) // 8 * 8
    Ab = torch.randn(M, Kp, device=d).bfloat16(); Bb = torch.randn(N, Kp, device=d).bfloat16()
    out = torch.zeros(M, N, device=d)
    bias = torch.zeros(N, device=d) if not acc else None
    prob = [dict(A=Ab, B=Bb, K=K, out=out, accumulate=acc, **({"bias": bias} if bias is not None else {}))]
    for _ in range(3):
        ops.gemm_bf16_grouped(prob)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.gemm_bf16_grouped(prob)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    io = (M * Kp * 2 + N * Kp * 2 + M * N * 4 * (2 if acc else 1)) / 1e6
    print(f"{name:10s} M={M:6d} N={N:5d} K={K:6d}: {us:8.1f} us  {2.0*M*N*K/us/1e6:7.1f} TFLOP/s   min HBM traffic {io:7.1f} MB -> {io/us*1e6/1e6:6.2f} TB/s")
