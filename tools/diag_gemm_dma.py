"""Diagnostic (GPU box): which side of the LDS-DMA GEMM's k-loop sets its pace -- the same launch with the MFMA work removed (mode 1),
with the DMA removed (mode 2), with both (mode 3: barriers and epilogue only)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mmda_amd import ops, _lib
lib = _lib.load()
d = torch.device("cuda:0")
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n
R = 12800
for name, M, N, K in [("dX L2", R, 600, 2400), ("fwd L2", R, 2400, 600), ("square 4096", 4096, 4096, 4096)]:
    A = (torch.randn(M, K, device=d) * 0.5).to(torch.bfloat16); B = (torch.randn(N, K, device=d) * 0.5).to(torch.bfloat16)
    p = dict(A=A, B=B, K=K, out=torch.zeros(M, N, device=d))
    res = []
    for mode in (0, 1, 2, 3):
        _lib.check(lib.mmda_debug_gemm_dma_mode(mode), "mode")
        res.append(timeit(lambda: ops.gemm_bf16_grouped([p])))
    _lib.check(lib.mmda_debug_gemm_dma_mode(0), "mode")
    fl = 2.0 * M * N * K
    print(f"{name:12s} M={M} N={N} K={K}: full {res[0]:7.1f} us ({fl/res[0]/1e6:6.1f} TF/s) | no MFMA {res[1]:7.1f} | no DMA {res[2]:7.1f} | neither {res[3]:7.1f}")
