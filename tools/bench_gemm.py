"""Diagnostic (GPU box): time the LSTM-sized GEMM shapes of one training step through the C ABI."""
import os, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mmda_amd import ops
d = torch.device("cuda:0")
T, B = 50, 32
R = T * B
shapes = []
for H, D1 in ((300, 300), (74, 74), (35, 35)):
    for D in (D1, 2 * H):
        shapes.append((f"fwd gates H={H} D={D}", dict(M=R, N=8 * H, K=D, tA=False, tB=True)))
        shapes.append((f"dW_ih     H={H} D={D}", dict(M=8 * H, N=D, K=R, tA=True, tB=False)))
        shapes.append((f"dX        H={H} D={D}", dict(M=R, N=D, K=8 * H, tA=False, tB=False)))
    shapes.append((f"dW_hh     H={H}", dict(M=4 * H, N=H, K=R - B, tA=True, tB=False)))
for name, s in shapes:
    M, N, K = s["M"], s["N"], s["K"]
    A = torch.randn((K, M) if s["tA"] else (M, K), device=d)
    Bm = torch.randn((N, K) if s["tB"] else (K, N), device=d)
    out = torch.zeros(M, N, device=d)
    for mode in ("bf16", "fp32"):
        for _ in range(3):
            ops.gemm(A, Bm, mode=mode, transA=s["tA"], transB=s["tB"], out=out, accumulate=True)
        torch.cuda.synchronize()
        e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
        e0.record()
        n = 20
        for _ in range(n):
            ops.gemm(A, Bm, mode=mode, transA=s["tA"], transB=s["tB"], out=out, accumulate=True)
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1e3 / n
        print(f"{name:26s} M={M:5d} N={N:5d} K={K:5d} {mode}: {us:8.1f} us  {2.0*M*N*K/us/1e6:8.1f} TFLOP/s")
    # bf16-operand NT kernel on pre-converted copies (conversion timed separately below)
    Kp = (K + 7) // 8 * 8
    Ab = torch.randn(M, Kp, device=d).bfloat16(); Bb = torch.randn(N, Kp, device=d).bfloat16()
    prob = [dict(A=Ab, B=Bb, K=K, out=out, accumulate=name.startswith("dW"))]   # weight gradients accumulate, the others overwrite
    for _ in range(3):
        ops.gemm_bf16_grouped(prob)
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.gemm_bf16_grouped(prob)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / 20
    print(f"{name:26s} M={M:5d} N={N:5d} K={K:5d} bf16-operands: {us:8.1f} us  {2.0*M*N*K/us/1e6:8.1f} TFLOP/s")
for rows, cols in ((R, 300), (R, 2400), (2400, 300), (R, 600)):
    X = torch.randn(rows, cols, device=d)
    for _ in range(3):
        ops.convert_bf16([(X, None, True, True)])
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        ops.convert_bf16([(X, None, True, True)])
    e1.record(); torch.cuda.synchronize()
    print(f"convert {rows}x{cols} plain+transposed: {e0.elapsed_time(e1) * 1e3 / 20:8.1f} us (includes torch.full fills)")
