"""Diagnostic (GPU box): the bf16 GEMMs of the text encoder at B = 32 / 256 (T = 50) -- every problem alone and the grouped launches a
training step issues (forward per layer; backward per layer: dX + dW_ih + 2 x dW_hh) -- timed and checked against torch.matmul on
the same bf16 values.  Environment switches (read once per process): MMDA_GEMM_DMA=0|1, MMDA_GEMM_DMA_STAGES=2|3, MMDA_GEMM_TN=..."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from mmda_amd import ops
d = torch.device("cuda:0")
torch.manual_seed(0)


def timeit(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / n


def nt(M, N, K, acc=False, bias=False):
    Kp = (K + 7) // 8 * 8
    A = torch.zeros(M, Kp, device=d, dtype=torch.bfloat16); A[:, :K] = torch.randn(M, K, device=d) * 0.5
    B = torch.zeros(N, Kp, device=d, dtype=torch.bfloat16); B[:, :K] = torch.randn(N, K, device=d) * 0.5
    p = dict(A=A, B=B, K=K, out=torch.zeros(M, N, device=d), accumulate=acc)
    if bias:
        p["bias"] = torch.randn(N, device=d)
    ref = lambda: A[:, :K].float() @ B[:, :K].float().t() + (p["bias"] if bias else 0)
    return p, ref, 2.0 * M * N * K


def tn(M, N, K, lda=None, ldb=None, bias_grad=True):
    lda = lda or (M + 7) // 8 * 8; ldb = ldb or (N + 7) // 8 * 8
    A = (torch.randn(K, lda, device=d) * 0.5).to(torch.bfloat16); B = (torch.randn(K, ldb, device=d) * 0.5).to(torch.bfloat16)
    p = dict(A=A, B=B, M=M, N=N, K=K, tn=True, out=torch.zeros(M, N, device=d), accumulate=True)
    if bias_grad:
        p["bias_grad"] = torch.zeros(M, device=d)
    ref = lambda: A[:, :M].float().t() @ B[:, :N].float()
    return p, ref, 2.0 * M * N * K


def check(p, ref):
    p["out"].zero_()
    if "bias_grad" in p:
        p["bias_grad"].zero_()
    ops.gemm_bf16_grouped([p])
    r = ref()
    err = float((p["out"] - r).abs().max() / r.abs().max())
    e2 = 0.0
    if "bias_grad" in p:
        rb = p["A"][:, :p["M"]].float().sum(0)
        e2 = float((p["bias_grad"] - rb).abs().max() / rb.abs().max())
    return err, e2


for Bsz in [int(x) for x in (sys.argv[1:] or ["32", "256"])]:
    R = 50 * Bsz
    print(f"==== B={Bsz} R={R}  DMA={os.environ.get('MMDA_GEMM_DMA','1')} stages={os.environ.get('MMDA_GEMM_DMA_STAGES','2')}")
    probs = {
        "fwd L1": nt(R, 2400, 300, bias=True), "fwd L2": nt(R, 2400, 600, bias=True),
        "dX L2": nt(R, 600, 2400), "dX L1": nt(R, 300, 2400),
        "dWih L2": tn(2400, 600, R, 2400, 608), "dWih L1": tn(2400, 300, R, 2400, 304),
        "dWhh f": tn(1200, 300, R - Bsz, 2400, 600, bias_grad=False), "dWhh r": tn(1200, 300, R - Bsz, 2400, 600, bias_grad=False),
        "a fwd L2": nt(R, 592, 148, bias=True), "a dWih L2": tn(592, 148, R, 592, 152),
    }
    for name, (p, ref, fl) in probs.items():
        err, e2 = check(p, ref)
        us = timeit(lambda: ops.gemm_bf16_grouped([p]))
        print(f"{name:10s} M={p['out'].shape[0]:6d} N={p['out'].shape[1]:5d} K={p['K']:6d}: {us:8.1f} us {fl / us / 1e6:7.1f} TF/s   err {err:.1e} bias_grad err {e2:.1e}")
    groups = {"bwd L2 (dX + dWih + 2 dWhh)": ["dX L2", "dWih L2", "dWhh f", "dWhh r"], "bwd L1 (dX + dWih + 2 dWhh)": ["dX L1", "dWih L1", "dWhh f", "dWhh r"]}
    for gname, names in groups.items():
        ps = [probs[n][0] for n in names]
        fl = sum(probs[n][2] for n in names)
        us = timeit(lambda: ops.gemm_bf16_grouped(ps))
        print(f"{gname:30s}: {us:8.1f} us {fl / us / 1e6:7.1f} TF/s")
    # determinism: the same grouped launch twice gives the same bits
    ps = [probs[n][0] for n in groups["bwd L2 (dX + dWih + 2 dWhh)"]]
    outs = []
    for _ in range(2):
        for p in ps:
            p["out"].zero_()
        ops.gemm_bf16_grouped(ps)
        outs.append([p["out"].clone() for p in ps])
    print("bitwise reproducible:", all(torch.equal(a, b) for a, b in zip(*outs)))
