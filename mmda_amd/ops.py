"""Thin per-operator wrappers over the C ABI (one function per entry point of include/mmda_hip.h).

These take/return torch CUDA tensors purely as device-memory handles; all arithmetic is in libmmda_hip.so.
Used by the parity tests (tests/test_gpu_ops.py) and available to callers who want single kernels.
"""
from __future__ import annotations

import ctypes as C

import torch

from . import _lib
from ._lib import ACT, BF16, F32, check, load, ptr, stream_ptr

MODE = {"fp32": F32, "bf16": BF16, F32: F32, BF16: BF16}


def _f(t):
    assert t.is_cuda and t.dtype == torch.float32 and (t.is_contiguous() or (t.dim() == 2 and t.stride(1) == 1)), \
        "expect contiguous (or row-strided 2-D) fp32 CUDA tensors"
    return t


def gemm(A, B, *, mode="fp32", transA=False, transB=True, bias=None, bias2=None, out=None, accumulate=False, act="none",
         A2=None, gather=None, alpha=1.0, drop_p=0.0, seed=0, site=0, gate=None, gate_scale=1.0, bias_grad=None, bias_grad2=None):
    """C = act(alpha * opA(A (+A2)) @ opB(B) + bias + bias2 (+C)).  A,B 2-D or 3-D (batched, uniform strides)."""
    lib = load()
    batched = A.dim() == 3
    Ab = A if batched else A.unsqueeze(0)
    Bb = B if B.dim() == 3 else B.unsqueeze(0)
    nb = max(Ab.shape[0], Bb.shape[0])
    if gather is not None:
        M = gather.numel(); K = Ab.shape[2]
    elif transA:
        K, M = Ab.shape[1], Ab.shape[2]
    else:
        M, K = Ab.shape[1], Ab.shape[2]
    N = Bb.shape[1] if transB else Bb.shape[2]
    if out is None:
        out = torch.zeros((nb, M, N), device=A.device, dtype=torch.float32)
        if not batched and B.dim() == 2:
            out = out[0]
    Cb = out if out.dim() == 3 else out.unsqueeze(0)
    g = _lib.GemmArgs()
    g.mode = MODE[mode]; g.transA = int(transA); g.transB = int(transB); g.M = M; g.N = N; g.K = K; g.batch = nb
    g.A = ptr(_f(Ab)); g.lda = Ab.shape[2]; g.strideA = Ab.stride(0) if Ab.shape[0] > 1 else 0
    g.A2 = ptr(A2); g.gather = ptr(gather)
    g.B = ptr(_f(Bb)); g.ldb = Bb.shape[2]; g.strideB = Bb.stride(0) if Bb.shape[0] > 1 else 0
    g.C = ptr(_f(Cb)); g.ldc = N; g.strideC = Cb.stride(0) if Cb.shape[0] > 1 else 0
    g.bias = ptr(bias); g.bias2 = ptr(bias2)
    g.strideBias = (bias.stride(0) if (bias is not None and bias.dim() == 2) else 0)
    g.accumulate = int(accumulate); g.act = ACT[act]
    g.drop_p = drop_p; g.drop_seed = seed; g.drop_site = site
    g.gate = ptr(gate); g.ldgate = N; g.gate_scale = gate_scale; g.alpha = alpha
    g.bias_grad = ptr(bias_grad); g.bias_grad2 = ptr(bias_grad2)
    if bias_grad is not None and bias_grad.dim() == 2:
        g.strideBias = bias_grad.stride(0)
    check(lib.mmda_gemm(C.byref(g), stream_ptr()), "mmda_gemm")
    return out



def convert_bf16(jobs):
    """jobs: list of (src[rows, cols] fp32 or bf16, gather or None, want_plain, want_transposed) -> list of (plain, transposed) bf16 tensors.

    plain is [rows, round_up(cols, 8)], transposed is [cols, round_up(rows, 8)], both zero padded; one launch for all jobs."""
    lib = load()
    arr = (_lib.ConvertJob * len(jobs))()
    outs = []
    for j, (src, gather, want_p, want_t) in zip(arr, jobs):
        rows = gather.numel() if gather is not None else src.shape[0]
        cols = src.shape[1]
        ldp, ldt = (cols + 7) // 8 * 8, (rows + 7) // 8 * 8
        # filled with NaN patterns so that a padding element the kernel forgot shows up in the tests
        P = torch.full((rows, ldp), float("nan"), device=src.device, dtype=torch.bfloat16) if want_p else None
        T = torch.full((cols, ldt), float("nan"), device=src.device, dtype=torch.bfloat16) if want_t else None
        if src.dtype == torch.bfloat16:                 # re-layout of an existing bf16 matrix (row stride = its leading dimension)
            assert src.is_cuda and src.stride(1) == 1
            j.src = ptr(src); j.ld = src.stride(0); j.src_bf16 = 1
        else:
            j.src = ptr(_f(src)); j.ld = src.shape[1]
        j.rows = rows; j.cols = cols; j.gather = ptr(gather)
        j.plain = ptr(P); j.ldp = ldp; j.transposed = ptr(T); j.ldt = ldt
        outs.append((P, T))
    check(lib.mmda_convert_bf16(arr, len(jobs), stream_ptr()), "mmda_convert_bf16")
    return outs


def gemm_bf16_grouped(problems):
    """problems: list of dicts A[M, lda] bf16, B[N, ldb] bf16 (both K-major), K, optional out/bias/bias2/bias_grad/bias_grad2/accumulate/alpha.
    C = alpha * A[:, :K] @ B[:, :K]^T + bias + bias2 (+C); bias_grad[m] += sum_k A[m, k].  One launch per 16 problems."""
    lib = load()
    arr = (_lib.GemmBf16Args * len(problems))()
    outs = []
    for g, p in zip(arr, problems):
        A, B = p["A"], p["B"]
        tn = bool(p.get("tn", False))
        assert A.dtype == torch.bfloat16 and B.dtype == torch.bfloat16 and (tn or (A.is_contiguous() and B.is_contiguous()))
        # tn: A is (K, lda) with M <= lda columns in use, B is (K, ldb): C = A[:K, :M]^T @ B[:K, :N]
        M, N, K = (p["M"], p["N"], p["K"]) if tn else (A.shape[0], B.shape[0], p["K"])
        g.tn = int(tn); g.perm_m_H = int(p.get("perm_m_H", 0))
        out = p.get("out")
        if out is None:
            out = torch.zeros((M, N), device=A.device, dtype=torch.float32)
        g.M = M; g.N = N; g.K = K; g.A = ptr(A); g.lda = A.stride(0) if tn else A.shape[1]; g.B = ptr(B); g.ldb = B.stride(0) if tn else B.shape[1]
        g.C = ptr(_f(out)); g.ldc = N
        g.bias = ptr(p.get("bias")); g.bias2 = ptr(p.get("bias2"))
        g.bias_grad = ptr(p.get("bias_grad")); g.bias_grad2 = ptr(p.get("bias_grad2"))
        g.accumulate = int(p.get("accumulate", False)); g.alpha = p.get("alpha", 1.0)
        outs.append(out)
    check(lib.mmda_gemm_bf16_grouped(arr, len(problems), stream_ptr()), "mmda_gemm_bf16_grouped")
    return outs


def gemm_skinny(problems):
    """Row-skinny exact-f32 GEMMs, up to 8 problems per launch.  Each problem is a dict: A[M,K], B ([N,K] if transB else [K,N]),
    optional A2, second product (A_2nd, B_2nd), out/out2, bias, accumulate, alpha, act, drop_p/seed/site, gate/gate_scale,
    dsig/dsig2.  Returns the list of outputs (out2 tensors are updated in place)."""
    lib = load()
    arr = (_lib.SkinnyArgs * len(problems))()
    outs = []
    for g, p in zip(arr, problems):
        A, B = _f(p["A"]), _f(p["B"])
        tb = bool(p.get("transB", True))
        M, K = A.shape
        N = B.shape[0] if tb else B.shape[1]
        out = p.get("out")
        if out is None:
            out = torch.zeros((M, N), device=A.device, dtype=torch.float32)
        g.M = M; g.N = N; g.K = K; g.transB = int(tb)
        g.A = ptr(A); g.A2 = ptr(p.get("A2")); g.lda = A.stride(0); g.B = ptr(B); g.ldb = B.stride(0)
        if p.get("A_2nd") is not None:
            A2n, B2n = p["A_2nd"], p["B_2nd"]
            g.K2 = A2n.shape[1]; g.A_2nd = ptr(A2n); g.lda_2nd = A2n.stride(0); g.B_2nd = ptr(B2n); g.ldb_2nd = B2n.stride(0)
        g.C = ptr(out); g.ldc = out.stride(0)
        if p.get("out2") is not None:
            g.C2 = ptr(p["out2"]); g.ldc2 = p["out2"].stride(0)
        g.bias = ptr(p.get("bias")); g.accumulate = int(p.get("accumulate", False)); g.alpha = p.get("alpha", 1.0)
        g.act = ACT[p.get("act", "none")]
        g.drop_p = p.get("drop_p", 0.0); g.drop_seed = p.get("seed", 0); g.drop_site = p.get("site", 0)
        if p.get("gate") is not None:
            g.gate = ptr(p["gate"]); g.ldgate = p["gate"].stride(0); g.gate_scale = p.get("gate_scale", 1.0)
        if p.get("dsig") is not None:
            g.dsig = ptr(p["dsig"]); g.lddsig = p["dsig"].stride(0)
        g.dsig2 = ptr(p.get("dsig2"))
        outs.append(out)
    check(lib.mmda_gemm_skinny(arr, len(problems), stream_ptr()), "mmda_gemm_skinny")
    return outs


def transpose_f32(mats):
    """fp32 transposes of several matrices in one launch."""
    lib = load()
    arr = (_lib.TransposeJob * len(mats))()
    outs = []
    for j, x in zip(arr, mats):
        r, c = x.shape
        o = torch.full((c, r), float("nan"), device=x.device)
        j.src = ptr(_f(x)); j.rows = r; j.cols = c; j.ld = x.stride(0); j.dst = ptr(o); j.ldd = r
        outs.append(o)
    check(lib.mmda_transpose_f32(arr, len(mats), stream_ptr()), "mmda_transpose_f32")
    return outs

def colsum(X, out=None, out2=None):
    lib = load()
    M, N = X.shape
    if out is None:
        out = torch.zeros(N, device=X.device)
    check(lib.mmda_colsum(ptr(_f(X)), N, M, N, ptr(out), ptr(out2), stream_ptr()), "mmda_colsum")
    return out


def embed_gather(W, ids):
    lib = load()
    rows, dim = ids.numel(), W.shape[1]
    out = torch.empty(tuple(ids.shape) + (dim,), device=W.device)
    check(lib.mmda_embed_gather(ptr(_f(W)), ptr(ids.contiguous()), rows, dim, ptr(out), stream_ptr()), "embed_gather")
    return out


def embed_scatter_add(dW, ids, dX):
    lib = load()
    check(lib.mmda_embed_scatter_add(ptr(_f(dW)), ptr(ids.contiguous()), ids.numel(), dW.shape[1], ptr(_f(dX)), stream_ptr()),
          "embed_scatter_add")
    return dW


def mx8_quant(x):
    """fp32 (rows, K) -> (element bytes (rows * K,) uint8 in MFMA operand order, scale bytes (rows * K / 32,) uint8): mmda_mx8_quant."""
    lib = load()
    rows, K = x.shape
    q = torch.empty(rows * K, dtype=torch.uint8, device=x.device)
    s = torch.empty(rows * K // 32 + 16, dtype=torch.uint8, device=x.device)
    j = (_lib.Mx8QuantJob * 1)()
    j[0].src = ptr(_f(x)); j[0].ld = K; j[0].rows = rows; j[0].K = K; j[0].q = ptr(q); j[0].s = ptr(s)
    check(lib.mmda_mx8_quant(j, 1, stream_ptr()), "mmda_mx8_quant")
    return q, s


def gemm_mx8(A, B, *, bias=None, act="none", drop_p=0.0, seed=0, site=0):
    """act(A (M, K) @ B (N, K)^T + bias) * dropout with both operands quantised to block-scaled fp8 (OCP MX e4m3): mmda_gemm_mx8."""
    lib = load()
    M, K = A.shape
    N = B.shape[0]
    Aq, As = mx8_quant(A)
    Bq, Bs = mx8_quant(B)
    out = torch.empty(M, N, dtype=torch.float32, device=A.device)
    g = _lib.Mx8Args()
    g.M = M; g.N = N; g.K = K; g.Aq = ptr(Aq); g.As = ptr(As); g.Bq = ptr(Bq); g.Bs = ptr(Bs); g.C = ptr(out); g.ldc = N
    g.bias = ptr(bias); g.act = ACT[act]; g.drop_p = drop_p; g.drop_seed = seed; g.drop_site = site
    check(lib.mmda_gemm_mx8(C.byref(g), stream_ptr()), "mmda_gemm_mx8")
    return out


def embed_segment_sum(dW, ids, rows):
    """dW[id] = list-order sum of rows[p] over ids[p] == id (overwrites those rows; ids < 0 skipped): mmda_embed_segment_sum."""
    lib = load()
    n, D = rows.shape
    need = int(lib.mmda_embed_segment_sum_work_bytes(n, D))
    work = torch.empty(max(need, 256), dtype=torch.uint8, device=dW.device)
    check(lib.mmda_embed_segment_sum(ptr(_f(dW)), ptr(ids), n, D, ptr(_f(rows)), ptr(work), work.numel(), stream_ptr()), "mmda_embed_segment_sum")
    return dW


def layernorm_fwd(x, gamma, beta, *, res=None, act="none", drop_p=0.0, seed=0, site=0, permute=None, eps=1e-5):
    lib = load()
    n = x.shape[-1]
    rows = x.numel() // n
    y = torch.empty_like(x)
    mean = torch.empty(rows, device=x.device); rstd = torch.empty(rows, device=x.device)
    a = _lib.LnArgs()
    a.rows = rows; a.n = n; a.x = ptr(_f(x)); a.res = ptr(res); a.gamma = ptr(gamma); a.beta = ptr(beta)
    a.y = ptr(y); a.mean = ptr(mean); a.rstd = ptr(rstd); a.act = ACT[act]
    a.drop_p = drop_p; a.drop_seed = seed; a.drop_site = site
    a.permute_S, a.permute_B = permute if permute else (0, 0)
    a.eps = eps
    check(lib.mmda_layernorm_fwd(C.byref(a), stream_ptr()), "layernorm_fwd")
    if permute:
        y = y.view(permute[1], permute[0], n)
    return y, mean, rstd


def layernorm_bwd(dy, x, gamma, mean, rstd, *, res=None, act="none", drop_p=0.0, seed=0, site=0, permute=None,
                  want_dres=False):
    lib = load()
    n = x.shape[-1]
    rows = x.numel() // n
    dx = torch.empty_like(x)
    dres = torch.empty_like(x) if want_dres else None
    dg = torch.zeros(n, device=x.device); db = torch.zeros(n, device=x.device)
    a = _lib.LnBwdArgs()
    a.rows = rows; a.n = n; a.dy = ptr(_f(dy)); a.x = ptr(_f(x)); a.res = ptr(res); a.gamma = ptr(gamma)
    a.mean = ptr(mean); a.rstd = ptr(rstd); a.d_x = ptr(dx); a.accumulate_dx = 0; a.d_res = ptr(dres)
    a.dgamma = ptr(dg); a.dbeta = ptr(db); a.act = ACT[act]
    a.drop_p = drop_p; a.drop_seed = seed; a.drop_site = site
    a.permute_S, a.permute_B = permute if permute else (0, 0)
    check(lib.mmda_layernorm_bwd(C.byref(a), stream_ptr()), "layernorm_bwd")
    return dx, dres, dg, db



def layernorm_multi(xs, gammas, betas, dys):
    """Several LayerNorms through the multi-problem launches: forward, input gradients (no parameter gradients) and the
    separate parameter-gradient pass.  Returns per problem (y, dx, dgamma, dbeta)."""
    lib = load()
    k = len(xs)
    fa = (_lib.LnArgs * k)(); ba = (_lib.LnBwdArgs * k)()
    keep = []
    for a, b, x, g, be, dy in zip(fa, ba, xs, gammas, betas, dys):
        rows, n = x.shape
        y = torch.empty_like(x); mean = torch.empty(rows, device=x.device); rstd = torch.empty(rows, device=x.device)
        dx = torch.empty_like(x); dg = torch.zeros(n, device=x.device); db = torch.zeros(n, device=x.device)
        a.rows = rows; a.n = n; a.x = ptr(_f(x)); a.gamma = ptr(g); a.beta = ptr(be); a.y = ptr(y); a.mean = ptr(mean); a.rstd = ptr(rstd)
        a.eps = 1e-5
        b.rows = rows; b.n = n; b.dy = ptr(_f(dy)); b.x = ptr(x); b.gamma = ptr(g); b.mean = ptr(mean); b.rstd = ptr(rstd); b.d_x = ptr(dx)
        keep.append((y, dx, dg, db, mean, rstd))
    check(lib.mmda_layernorm_fwd_multi(fa, k, stream_ptr()), "layernorm_fwd_multi")
    check(lib.mmda_layernorm_bwd_multi(ba, k, stream_ptr()), "layernorm_bwd_multi")      # dgamma/dbeta NULL: dx only
    for b, (y, dx, dg, db, mean, rstd) in zip(ba, keep):
        b.dgamma = ptr(dg); b.dbeta = ptr(db); b.d_x = None
    check(lib.mmda_layernorm_param_grads(ba, k, stream_ptr()), "layernorm_param_grads")
    return [(y, dx, dg, db) for (y, dx, dg, db, _, _) in keep]

def lstm_pack(whh, mode):
    """Returns (packed_fwd, packed_bwd) byte tensors for one direction's W_hh (4H,H)."""
    lib = load()
    H = whh.shape[1]
    md = MODE[mode]
    pf = torch.empty(lib.mmda_lstm_packed_bytes(md, H, 0), dtype=torch.uint8, device=whh.device)
    pb = torch.empty(lib.mmda_lstm_packed_bytes(md, H, 1), dtype=torch.uint8, device=whh.device)
    check(lib.mmda_lstm_pack_whh(md, H, ptr(_f(whh)), ptr(pf), ptr(pb), stream_ptr()), "lstm_pack")
    return pf, pb


def lstm_pack_cluster(whh):
    """Cluster-backward packing (bf16) of one direction's W_hh."""
    lib = load()
    H = whh.shape[1]
    pc = torch.empty(lib.mmda_lstm_packed_bytes(BF16, H, 2), dtype=torch.uint8, device=whh.device)
    check(lib.mmda_lstm_pack_whh_cluster(H, ptr(_f(whh)), ptr(pc), stream_ptr()), "lstm_pack_cluster")
    return pc


def _desc(H, gates, cstash, hseq, wp0, wp1, utt, layer, d_hseq=None, xchg=None, epoch_base=0, wc0=None, wc1=None):
    d = _lib.LstmDesc()
    d.H = H; d.gates = ptr(gates); d.cstash = ptr(cstash); d.hseq = ptr(hseq)
    d.wpack[0] = ptr(wp0); d.wpack[1] = ptr(wp1)
    d.wpack_c[0] = ptr(wc0); d.wpack_c[1] = ptr(wc1)
    d.utt = ptr(utt); d.layer = layer; d.d_hseq = ptr(d_hseq)
    d.xchg = ptr(xchg); d.epoch_base = epoch_base
    return d


def lstm_xchg(H, B, device):
    """Zeroed cluster-exchange buffer (None when the shape has no resident-weights plan)."""
    n = load().mmda_lstm_xchg_bytes(H, B)
    return torch.zeros(n, dtype=torch.uint8, device=device) if n > 0 else None


def _to_gate_minor(x, H):
    """(..., 4H) [gate][unit] -> [unit][gate] (the resident-weights kernels' 16-byte layout)"""
    return x.reshape(*x.shape[:-1], 4, H).transpose(-1, -2).reshape(*x.shape[:-1], 4 * H).contiguous()


def _from_gate_minor(x, H):
    return x.reshape(*x.shape[:-1], H, 4).transpose(-1, -2).reshape(*x.shape[:-1], 4 * H).contiguous()


def gru_pad(rnn_params, H, D, device):
    """torch nn.GRU(bidirectional) parameters -> the four-slot layout (mmda_gru_pad_params).  rnn_params: dict with
    weight_ih_l0, weight_hh_l0, bias_ih_l0, bias_hh_l0 and their _reverse twins.  Returns (job, padded dict); the job can be
    handed to gru_unpad_grads with gradient tensors of the same shapes."""
    lib = load()
    t = {k: _f(v.detach().to(device)) for k, v in rnn_params.items()}
    pad = dict(w_ih=torch.full((8 * H, D), float("nan"), device=device), w_hh_f=torch.full((4 * H, H), float("nan"), device=device),
               w_hh_r=torch.full((4 * H, H), float("nan"), device=device), b_ih=torch.full((8 * H,), float("nan"), device=device),
               b_hh=torch.full((8 * H,), float("nan"), device=device))
    check(lib.mmda_gru_pad_params((_lib.GruPadJob * 1)(_gru_job(t, pad, H, D)), 1, stream_ptr()), "gru_pad_params")
    return pad, t


def _gru_job(t, pad, H, D):
    j = _lib.GruPadJob()
    j.H, j.D = H, D
    for d, sfx in enumerate(("", "_reverse")):
        j.w_ih[d] = ptr(t["weight_ih_l0" + sfx]); j.w_hh[d] = ptr(t["weight_hh_l0" + sfx])
        j.b_ih[d] = ptr(t["bias_ih_l0" + sfx]); j.b_hh[d] = ptr(t["bias_hh_l0" + sfx])
    j.pw_ih = ptr(pad["w_ih"]); j.pw_hh[0] = ptr(pad["w_hh_f"]); j.pw_hh[1] = ptr(pad["w_hh_r"])
    j.pb_ih = ptr(pad["b_ih"]); j.pb_hh = ptr(pad["b_hh"]) if pad.get("b_hh") is not None else None
    return j


def gru_unpad_grads(grads, pad_grads, H, D):
    """grads (torch-layout dict, accumulated into) += pad_grads (four-slot: w_ih, w_hh_f, w_hh_r, b_ih = gate-gradient column
    sums); pad_grads are zeroed."""
    lib = load()
    check(lib.mmda_gru_unpad_grads((_lib.GruPadJob * 1)(_gru_job(grads, dict(pad_grads, b_hh=None), H, D)), 1, stream_ptr()),
          "gru_unpad_grads")


def lstm_bidir_fwd(pre, whh_f, whh_r, lengths, *, mode="fp32", layer=0, utt=None, resident=False, gate_minor=False, cell="lstm"):
    """pre: (T,B,2,4H) = x W_ih^T + b_ih + b_hh per direction.  Returns dict(hseq, gates, cstash, utt, packs).
    gate_minor (resident only): the kernels see `gates` as [dir][unit][gate]; inputs/outputs here stay in torch's order.
    cell="gru": pre / whh_* are in the four-slot layout (gru_pad)."""
    lib = load()
    T, B, _, G4 = pre.shape
    H = G4 // 4
    gates = _to_gate_minor(pre, H) if gate_minor else pre.clone().contiguous()
    # torch layout (T,B,2,H); the gate-minor kernels keep it batch-minor-by-4, (T, ceil(B/4), 2, H, 4): rows rounded up
    cst = torch.zeros(T, (B + 3) // 4, 2, H, 4, device=pre.device) if gate_minor else torch.zeros(T, B, 2, H, device=pre.device)
    hseq = torch.full((T, B, 2 * H), float("nan"), device=pre.device)
    if utt is None:
        utt = torch.zeros(B, 4 * H, device=pre.device)
    pf0, pb0 = lstm_pack(whh_f, mode)
    pf1, pb1 = lstm_pack(whh_r, mode)
    len_dev = lengths.to(device=pre.device, dtype=torch.int32)
    xchg = lstm_xchg(H, B, pre.device) if resident else None
    pcs = (lstm_pack_cluster(whh_f), lstm_pack_cluster(whh_r)) if (resident and MODE[mode] == BF16) else (None, None)
    d = (_lib.LstmDesc * 1)(_desc(H, gates, cst, hseq, pf0, pf1, utt, layer, None, xchg, 0))
    d[0].gate_minor = int(gate_minor)
    d[0].cell = _lib.CELL[cell]
    check(lib.mmda_lstm_fwd(MODE[mode], 1, d, B, T, ptr(len_dev), stream_ptr()), "lstm_fwd")
    return dict(hseq=hseq, gates=gates, cstash=cst, utt=utt, packs=(pf0, pb0, pf1, pb1), len_dev=len_dev, xchg=xchg, T=T, pcs=pcs,
                gate_minor=bool(gate_minor), H=H, cell=cell)


def lstm_bidir_bwd(fw, d_utt, d_hseq, *, mode="fp32", layer=0, dg_bf16_only=False):
    """Consumes the dict returned by lstm_bidir_fwd; returns dG (T,B,2,4H) (overwrites fw['gates'])."""
    lib = load()
    gates = fw["gates"]
    T, B, _, G4 = gates.shape
    H = G4 // 4
    pf0, pb0, pf1, pb1 = fw["packs"]
    pc0, pc1 = fw.get("pcs", (None, None))
    d = (_lib.LstmDesc * 1)(_desc(H, gates, fw["cstash"], fw["hseq"], pb0, pb1, d_utt, layer, d_hseq, fw.get("xchg"), T + 2, pc0, pc1))
    d[0].gate_minor = int(fw.get("gate_minor", False))
    d[0].cell = _lib.CELL[fw.get("cell", "lstm")]
    if MODE[mode] == BF16 and lib.mmda_lstm_bwd_emits_dg_bf16(BF16, 1, d, B, T):
        # the resident gate-minor kernel also emits dG rounded to bf16 (kernel column order); NaN-filled to expose gaps
        fw["dg_bf16"] = torch.full((T * B, 2 * G4), float("nan"), device=gates.device, dtype=torch.bfloat16)
        d[0].dg_bf16 = ptr(fw["dg_bf16"])
        d[0].dg_bf16_only = int(dg_bf16_only)
    check(lib.mmda_lstm_bwd(MODE[mode], 1, d, B, T, ptr(fw["len_dev"]), stream_ptr()), "lstm_bwd")
    return _from_gate_minor(gates, H) if fw.get("gate_minor") else gates


def lstm_aborted(fw):
    """True if a cluster exchange timed out (sticky abort word at the start of the exchange buffer)."""
    x = fw.get("xchg")
    return False if x is None else bool(x[:4].view(torch.int32).item() != 0)


def attn_fwd(qkv, S, B, E, nhead, drop_p=0.0, seed=0, site=0):
    lib = load()
    ctx = torch.empty(S * B, E, device=qkv.device)
    probs = torch.empty(B, nhead, S, S, device=qkv.device)
    check(lib.mmda_attn_fwd(ptr(_f(qkv)), S, B, E, nhead, ptr(ctx), ptr(probs), drop_p, seed, site, stream_ptr()), "attn_fwd")
    return ctx, probs


def attn_bwd(qkv, probs, dctx, S, B, E, nhead, drop_p=0.0, seed=0, site=0):
    lib = load()
    dqkv = torch.empty_like(qkv)
    check(lib.mmda_attn_bwd(ptr(_f(qkv)), ptr(_f(probs)), ptr(_f(dctx)), S, B, E, nhead, ptr(dqkv), drop_p, seed, site,
                            stream_ptr()), "attn_bwd")
    return dqkv


def clamp_adam(p, g, m, v, lr, step, clip=float("inf"), grad_scale=1.0, betas=(0.9, 0.999), eps=1e-8):
    lib = load()
    check(lib.mmda_clamp_adam(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, betas[0], betas[1], eps, clip, grad_scale, step,
                              stream_ptr()), "clamp_adam")


def mark_rows(ids, rows):
    """uint8 mask of `rows` bytes: 1 where the row id occurs in `ids` (int64, device)."""
    lib = load()
    mask = torch.empty(rows, dtype=torch.uint8, device=ids.device)
    check(lib.mmda_mark_rows(ptr(mask), rows, ptr(ids), ids.numel(), stream_ptr()), "mark_rows")
    return mask


def clamp_adam_rows(p, g, m, v, mask, want, lr, step, clip=float("inf"), grad_scale=1.0, betas=(0.9, 0.999), eps=1e-8):
    """clip + Adam over the rows of the (rows, dim) table p whose mask byte equals `want` (in place)."""
    lib = load()
    rows, dim = p.shape
    check(lib.mmda_clamp_adam_rows(ptr(p), ptr(g), ptr(m), ptr(v), rows, dim, ptr(mask), int(want), lr, betas[0], betas[1], eps, clip,
                                   grad_scale, step, stream_ptr()), "clamp_adam_rows")


def heads_fwd(logits, ncls, threshold=0.35, drop_p=0.0, seed=0, site=0):
    lib = load()
    B = logits.shape[0]
    tcp = torch.empty(B, 6, device=logits.device); sc = torch.empty(B, ncls, device=logits.device)
    lab = torch.empty(B, ncls, device=logits.device)
    check(lib.mmda_heads_fwd(ptr(_f(logits)), B, ncls, threshold, ptr(tcp), ptr(sc), ptr(lab), drop_p, seed, site, stream_ptr()),
          "heads_fwd")
    return tcp, sc, lab


def dropout_mask_via_act(n, p, seed, site, device):
    """Exposes the counter-based dropout multipliers (for the statistical tests): h = dropout(identity(1))."""
    lib = load()
    z = torch.ones(n, device=device); h = torch.empty(n, device=device)
    check(lib.mmda_act_dropout_fwd(ptr(z), ptr(h), n, ACT["none"], p, seed, site, stream_ptr()), "act_dropout_fwd")
    return h
