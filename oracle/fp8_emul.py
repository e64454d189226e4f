"""fp8-emulating CPU oracle for the fusion layer's feed-forward products.  TEST INFRASTRUCTURE ONLY (see misa_oracle.py's header).

BASELINE.json configs[4] runs linear1 / linear2 of the fusion transformer layer (reference models.py:160-161) on "mixed fp8 fusion
GEMMs".  The product's form (mmda_amd/csrc/gemm_mx8.hip) is OCP-MX block-scaled e4m3 in the FORWARD products only:

    q(x): per 32 consecutive elements along k:  e = floor(log2(amax)) - 8;  x / 2^e clamped to +-448, rounded to nearest even onto
          the e4m3 grid (torch.float8_e4m3fn), times 2^e
    f1 = relu(q(x1) . q(W1)^T + b1);   f2 = q(f1) . q(W2)^T + b2      (fp32 accumulate)

and a straight-through backward: the gradient GEMMs are the exact ones on the stored fp32 activations (x1, and f1 as computed above)
and the fp32 weights.  ``ffn_mx8`` restates exactly that (an autograd Function per product) to plug into
``misa_oracle.forward(..., ffn=ffn_mx8)``.
The reference itself has no fp8 path (it is stock fp32 torch): what pins this file is the exact oracle it perturbs -- the test bounds
the distance between the two -- and the e4m3 grid of torch.float8_e4m3fn.
"""
from __future__ import annotations

import torch

from . import misa_oracle as orc

BLOCK = 32


def mx_quant(x: torch.Tensor) -> torch.Tensor:
    """Dequantised OCP-MX e4m3 image of x (last dimension a multiple of 32)."""
    shp = x.shape
    xb = x.detach().reshape(-1, BLOCK).float()
    amax = xb.abs().amax(dim=1, keepdim=True)
    _, ex = torch.frexp(amax)                                # amax = m * 2^ex, m in [0.5, 1)  ->  floor(log2 amax) = ex - 1
    e = (ex - 1 - 8).clamp(min=-127, max=127).float()
    e = torch.where(amax > 0, e, torch.full_like(e, -126.0 - 8))
    scale = torch.exp2(e)
    q = (xb / scale).clamp(-448.0, 448.0).to(torch.float8_e4m3fn).float() * scale
    return q.reshape(shp)


class _MxLinear(torch.autograd.Function):
    """y = q(x) . q(W)^T + b in the forward pass; in the backward pass the gradients of the EXACT product with the unquantised
    operands (dx = dy . W, dW = dy^T . x): the kernels keep x and W in fp32 for the gradient GEMMs."""

    @staticmethod
    def forward(ctx, x, w, b):
        ctx.save_for_backward(x, w)
        return mx_quant(x) @ mx_quant(w).t() + b

    @staticmethod
    def backward(ctx, dy):
        x, w = ctx.saved_tensors
        dy2, x2 = dy.reshape(-1, dy.shape[-1]), x.reshape(-1, x.shape[-1])
        return (dy2 @ w).reshape(x.shape), dy2.t() @ x2, dy2.sum(0)


def ffn_mx8(x, P):
    te = "transformer_encoder.layers.0."
    f1 = torch.relu(_MxLinear.apply(x, P[te + "linear1.weight"], P[te + "linear1.bias"]))
    return _MxLinear.apply(f1, P[te + "linear2.weight"], P[te + "linear2.bias"])


def loss_and_grads(P, cfg, batch):
    return orc.loss_and_grads(P, cfg, batch, ffn=ffn_mx8)
