"""Optimizers.  ``Adam`` keeps torch.optim.Adam's constructor (the reference builds it as
``config.optimizer(params, lr=...)``, solver.py:97-99) but steps with the fused HIP clamp+Adam kernel.

When every parameter is a view into one flat bucket whose gradient/moment buckets are laid out identically (that is
how mmda_amd.models.MISA allocates them) the whole model is ONE kernel launch; otherwise one launch per tensor.
"""
from __future__ import annotations

import torch

from . import _lib


class Adam(torch.optim.Optimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, clip_value=None):
        if weight_decay != 0:
            raise NotImplementedError("the reference never passes weight_decay (config.py:143 is a dead flag)")
        defaults = dict(lr=lr, betas=betas, eps=eps, clip_value=clip_value)
        super().__init__(params, defaults)
        self._model = None
        self._t = 0

    def attach(self, model):
        """Bind to a MISA model so the step is one fused launch over its flat buckets."""
        self._model = model
        return self

    @torch.no_grad()
    def step(self, closure=None, clip_value=None, grad_scale=1.0):
        lib = _lib.load()
        self._t += 1
        s = _lib.stream_ptr()
        g0 = self.param_groups[0]
        clip = clip_value if clip_value is not None else g0["clip_value"]
        clip = float("inf") if clip is None else float(clip)
        b1, b2 = g0["betas"]
        m = self._model
        if m is not None and m._P is not None and m._views_valid():
            P, G, M, V = m.flat_buckets()
            _lib.check(lib.mmda_clamp_adam(P.data_ptr(), G.data_ptr(), M.data_ptr(), V.data_ptr(), P.numel(), g0["lr"], b1, b2,
                                           g0["eps"], clip, grad_scale, self._t, s), "mmda_clamp_adam")
            return None
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.device.type != "cuda":
                    raise _lib.MMDAError("mmda_amd.optim.Adam steps on the GPU only")
                st = self.state[p]
                if not st:
                    st["m"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                    st["v"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                g = p.grad.contiguous()
                ok = all(x.data_ptr() % 16 == 0 for x in (p, g, st["m"], st["v"])) and p.is_contiguous()
                if not ok:
                    raise _lib.MMDAError("parameter storage is not 16-byte aligned/contiguous")
                _lib.check(lib.mmda_clamp_adam(p.data_ptr(), g.data_ptr(), st["m"].data_ptr(), st["v"].data_ptr(), p.numel(),
                                               group["lr"], b1, b2, group["eps"], clip, grad_scale, self._t, s), "mmda_clamp_adam")
        return None


def clip_grad_value_(model_or_params, clip_value):
    """torch.nn.utils.clip_grad_value_ on the flat gradient bucket (solver.py:185)."""
    lib = _lib.load()
    m = model_or_params
    if hasattr(m, "flat_buckets") and m._G is not None:
        _lib.check(lib.mmda_clamp(m._G.data_ptr(), m._G.numel(), float(clip_value), _lib.stream_ptr()), "mmda_clamp")
        return
    for p in m:
        if p.grad is not None:
            _lib.check(lib.mmda_clamp(p.grad.data_ptr(), p.grad.numel(), float(clip_value), _lib.stream_ptr()), "mmda_clamp")


optimizer_dict = {"Adam": Adam}
