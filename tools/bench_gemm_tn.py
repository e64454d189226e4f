"""Diagnostic (GPU box): the weight-gradient GEMM dW = dG^T X in the nt form (on transposed copies) and in the tn form."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mmda_amd import ops
d = torch.device("cuda:0")
def timeit(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
for (M, N, K) in [(2400, 300, 1600), (1200, 300, 1568), (2400, 600, 1600), (2400, 300, 12800), (1200, 300, 12544), (2400, 600, 12800)]:
    At = torch.randn(M, (K + 7) // 8 * 8, device=d).to(torch.bfloat16); Bt = torch.randn(N, (K + 7) // 8 * 8, device=d).to(torch.bfloat16)
    A = torch.randn(K, M, device=d).to(torch.bfloat16); Bm = torch.randn(K, (N + 7) // 8 * 8, device=d).to(torch.bfloat16)
    out = torch.zeros(M, N, device=d)
    t_nt = timeit(lambda: ops.gemm_bf16_grouped([dict(A=At, B=Bt, K=K, out=out, accumulate=True)]))
    t_tn = timeit(lambda: ops.gemm_bf16_grouped([dict(A=A, B=Bm, M=M, N=N, K=K, tn=True, out=out, accumulate=True)]))
    fl = 2.0 * M * N * K
    print(f"M={M} N={N} K={K}: nt {t_nt:7.1f} us ({fl / t_nt / 1e6:6.1f} TFLOP/s)   tn {t_tn:7.1f} us ({fl / t_tn / 1e6:6.1f} TFLOP/s)")
