"""Diagnostic (GPU box): bf16-operand GEMM time vs K at the forward-gates shape (fixed cost vs per-k-tile cost)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mmda_amd import ops
d = torch.device("cuda:0")
for (M, N) in ((1600, 2400), (2400, 600), (1600, 600)):
    for K in (64, 128, 320, 640, 1280, 2560):
        Ab = torch.randn(M, K, device=d).bfloat16(); Bb = torch.randn(N, K, device=d).bfloat16()
        out = torch.zeros(M, N, device=d)
        prob = [dict(A=Ab, B=Bb, K=K, out=out)]
        for _ in range(3): ops.gemm_bf16_grouped(prob)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        torch.cuda._sleep(4_000_000); e0.record()
        for _ in range(20): ops.gemm_bf16_grouped(prob)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / 20
        print(f"M={M} N={N} K={K:5d}: {us:7.1f} us  {2.0*M*N*K/us/1e6:7.1f} TFLOP/s")

def timed(prob, label):
    for _ in range(3): ops.gemm_bf16_grouped(prob)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    torch.cuda._sleep(4_000_000); e0.record()
    for _ in range(20): ops.gemm_bf16_grouped(prob)
    e1.record(); torch.cuda.synchronize()
    print(f"{label}: {e0.elapsed_time(e1) * 1e3 / 20:7.1f} us")

def P(M, N, K, bias=False, acc=False):
    Kp = (K + 7) // 8 * 8
    d_ = dict(A=torch.randn(M, Kp, device=d).bfloat16(), B=torch.randn(N, Kp, device=d).bfloat16(), K=K, out=torch.zeros(M, N, device=d), accumulate=acc)
    if bias:
        d_["bias"] = torch.randn(N, device=d); d_["bias2"] = torch.randn(N, device=d)
    return d_
R = 1600
t, v, a = P(R, 2400, 300, True), P(R, 280, 35, True), P(R, 592, 74, True)
timed([t], "fwd L1 text alone (bias)"); timed([v], "fwd L1 visual alone"); timed([a], "fwd L1 acoustic alone"); timed([t, v, a], "fwd L1 group of 3")
t2, v2, a2 = P(R, 2400, 600, True), P(R, 280, 70, True), P(R, 592, 148, True)
timed([t2, v2, a2], "fwd L2 group of 3")
dx = [P(R, 600, 2400), P(R, 70, 280), P(R, 148, 592)]
timed(dx, "dX L2 group of 3"); timed([dx[0]], "dX L2 text alone")
dw = [P(2400, 600, R, acc=True), P(1200, 300, R - 32, acc=True), P(1200, 300, R - 32, acc=True)]
timed(dw, "dW L2 text (ih + 2 hh)"); timed([dw[0]], "dW_ih L2 text alone"); timed([dw[1]], "dW_hh text alone")
