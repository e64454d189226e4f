"""Loss modules of the reference (src/utils/functions.py) backed by the HIP loss kernels.

Each kernel returns the loss value and d(loss)/d(inputs) in one pass; the autograd Functions here stash that gradient
in forward and scale it by the incoming grad_output in backward.  ``DiffLoss``, ``CMD``, ``ReverseLayerF`` and
``getBinaryTensor`` keep the reference's names and call signatures (functions.py:9-21, 49-115).
"""
from __future__ import annotations

import ctypes as C

import torch
import torch.nn as nn
from torch.autograd import Function

from .. import _lib


def _need_cuda(*ts):
    for t in ts:
        if t is not None and t.device.type != "cuda":
            raise _lib.MMDAError("HIP loss kernels need CUDA(ROCm) tensors; there is no CPU fallback")


def _pairs(lst):
    flat = [int(i) for p in lst for i in p]
    return (C.c_int * len(flat))(*flat), len(lst)


class _GradInForward(Function):
    """Base: subclasses implement ``run(*inputs) -> (loss (0-d tensor), grads tuple aligned with inputs)``."""

    @staticmethod
    def forward(ctx, fn, *inputs):
        loss, grads = fn(*[x.detach() for x in inputs])
        ctx.save_for_backward(*[g for g in grads if g is not None])
        ctx.mask = [g is not None for g in grads]
        return loss

    @staticmethod
    def backward(ctx, gout):
        saved = list(ctx.saved_tensors)
        outs = []
        for has in ctx.mask:
            outs.append(saved.pop(0) * gout if has else None)
        return (None,) + tuple(outs)


def _stack(ts):
    x = torch.stack([t.contiguous().float() for t in ts], dim=0).contiguous()
    return x, x.shape[1], x[0].numel() // x.shape[1]


def diff_loss_multi(tensors, pairs):
    """sum over `pairs` of DiffLoss(tensors[i], tensors[j]) in one fused pass (functions.py:54-78)."""
    _need_cuda(*tensors)
    lib = _lib.load()

    def run(*xs):
        x, B, D = _stack([t.reshape(t.shape[0], -1) for t in xs])
        nt = x.shape[0]
        loss = torch.zeros((), device=x.device)
        dx = torch.zeros_like(x)
        work = torch.empty(lib.mmda_loss_diff_work_floats(B, D), device=x.device)
        arr, n_pairs = _pairs(pairs)
        _lib.check(lib.mmda_loss_diff_pairs(x.data_ptr(), B * D, nt, n_pairs, arr, B, D, 1.0, loss.data_ptr(), dx.data_ptr(),
                                            work.data_ptr(), _lib.stream_ptr()), "mmda_loss_diff_pairs")
        return loss, tuple(dx[i].view_as(xs[i]) for i in range(nt))

    return _GradInForward.apply(run, *tensors)


def cmd_loss_multi(tensors, pairs, n_moments=5, value_scale=1.0):
    """value_scale * sum over `pairs` of CMD(tensors[i], tensors[j], n_moments) (functions.py:88-109)."""
    _need_cuda(*tensors)
    lib = _lib.load()

    def run(*xs):
        x, B, D = _stack(xs)
        nt = x.shape[0]
        loss = torch.zeros((), device=x.device)
        dx = torch.zeros_like(x)
        arr, n_pairs = _pairs(pairs)
        _lib.check(lib.mmda_loss_cmd_pairs(x.data_ptr(), B * D, nt, n_pairs, arr, int(n_moments), B, D, 1.0, float(value_scale),
                                           loss.data_ptr(), dx.data_ptr(), _lib.stream_ptr()), "mmda_loss_cmd_pairs")
        return loss, tuple(dx[i] for i in range(nt))

    return _GradInForward.apply(run, *tensors)


class DiffLoss(nn.Module):
    """reference functions.py:49-78"""

    def forward(self, input1, input2):
        return diff_loss_multi([input1, input2], [(0, 1)])


class CMD(nn.Module):
    """reference functions.py:80-109 (central moment discrepancy)"""

    def forward(self, x1, x2, n_moments):
        return cmd_loss_multi([x1, x2], [(0, 1)], n_moments)


def bce_sum_over_classes(scores, emo):
    """sum_c BCELoss_mean(scores[:,c], emo[:,c])   (solver.py:373-385)"""
    _need_cuda(scores, emo)
    lib = _lib.load()

    def run(s, y):
        s = s.contiguous().float(); y = y.contiguous().float()
        B, nc = s.shape
        loss = torch.zeros((), device=s.device)
        ds = torch.zeros_like(s)
        _lib.check(lib.mmda_loss_cls(s.data_ptr(), y.data_ptr(), B, nc, 1.0, loss.data_ptr(), ds.data_ptr(), _lib.stream_ptr()),
                   "mmda_loss_cls")
        return loss, (ds, None)

    return _GradInForward.apply(run, scores, emo)


def conf_loss(scores, tcp, emo):
    """tcp_loss + mcp_loss of solver.py:451-462 (the target emo*score is NOT detached there, nor here)."""
    _need_cuda(scores, tcp, emo)
    lib = _lib.load()

    def run(s, t, y):
        s = s.contiguous().float(); t = t.contiguous().float(); y = y.contiguous().float()
        B, nc = s.shape
        loss = torch.zeros((), device=s.device)
        ds = torch.zeros_like(s); dt = torch.zeros_like(t)
        _lib.check(lib.mmda_loss_conf(s.data_ptr(), t.data_ptr(), y.data_ptr(), B, nc, 1.0, loss.data_ptr(), ds.data_ptr(),
                                      dt.data_ptr(), _lib.stream_ptr()), "mmda_loss_conf")
        return loss, (ds, dt, None)

    return _GradInForward.apply(run, scores, tcp, emo)


def recon_loss(recons, origs):
    """mean over the three modalities of MSE_mean(recon, orig) (solver.py:443-449); gradients flow to both sides."""
    _need_cuda(*recons, *origs)
    lib = _lib.load()

    def run(*xs):
        r, B, D = _stack(xs[:3])
        o, _, _ = _stack(xs[3:])
        loss = torch.zeros((), device=r.device)
        dr = torch.zeros_like(r); do = torch.zeros_like(o)
        _lib.check(lib.mmda_loss_recon(r.data_ptr(), o.data_ptr(), B * D, B, D, 1.0, loss.data_ptr(), dr.data_ptr(), do.data_ptr(),
                                       _lib.stream_ptr()), "mmda_loss_recon")
        return loss, tuple(dr[i] for i in range(3)) + tuple(do[i] for i in range(3))

    return _GradInForward.apply(run, *recons, *origs)


def domain_loss(dom_t, dom_v, dom_a):
    """CrossEntropy_mean(cat(dom_t,dom_v,dom_a), [0]*B+[1]*B+[2]*B) (solver.py:388-407)"""
    _need_cuda(dom_t, dom_v, dom_a)
    lib = _lib.load()

    def run(*xs):
        d, B, _ = _stack(xs)
        loss = torch.zeros((), device=d.device)
        dd = torch.zeros_like(d)
        _lib.check(lib.mmda_loss_domain(d.data_ptr(), B, 1.0, loss.data_ptr(), dd.data_ptr(), _lib.stream_ptr()), "mmda_loss_domain")
        return loss, tuple(dd[i] for i in range(3))

    return _GradInForward.apply(run, dom_t, dom_v, dom_a)


class ReverseLayerF(Function):
    """Gradient reversal (reference functions.py:9-21).  Inside MISA the reversal is fused into the native backward;
    this standalone form is kept for API parity."""

    @staticmethod
    def forward(ctx, x, p):
        ctx.p = p
        return x.view_as(x)

    @staticmethod
    def backward(ctx, grad_output):
        return grad_output.neg() * ctx.p, None


def getBinaryTensor(imgTensor, boundary=0.35):
    """reference functions.py:112-115"""
    return (imgTensor > boundary).to(imgTensor.dtype)
