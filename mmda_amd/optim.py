"""Optimizers.  ``Adam`` / ``RMSprop`` keep the torch.optim constructors (the reference builds its optimizer as
``config.optimizer(params, lr=...)`` out of ``optimizer_dict = {'RMSprop', 'Adam'}``, config.py:24, solver.py:97-99) but step
with the fused HIP clamp+update kernels.

When every parameter is a view into one flat bucket whose gradient/moment buckets are laid out identically (that is
how mmda_amd.models.MISA allocates them) the whole model is ONE kernel launch; otherwise one launch per tensor.
``state_dict()`` / ``load_state_dict()`` of an attached optimizer carry the flat moment buckets and the step count, so a
run can be resumed from ``checkpoints/optim_{name}.std`` (solver.py:220 saves it beside the model).
"""
from __future__ import annotations

import torch

from . import _lib


class _FlatOptimizer(torch.optim.Optimizer):
    """Shared plumbing: binding to a MISA model's flat buckets, the step counter, (de)serialisation of the flat state."""

    def __init__(self, params, defaults):
        super().__init__(params, defaults)
        self._model = None
        self._t = 0

    def attach(self, model):
        """Bind to a MISA model so the step is one fused launch over its flat buckets (and shares the model's step counter
        with the native fused train step)."""
        self._model = model
        return self

    def _flat(self):
        m = self._model
        if m is not None and m._P is not None and m._views_valid():
            return m
        return None

    def _next_step(self) -> int:
        m = self._model
        if m is not None:
            m._step += 1
            self._t = m._step
        else:
            self._t += 1
        return self._t

    # ---- checkpointing (solver.py:220: torch.save(self.optimizer.state_dict(), 'checkpoints/optim_{name}.std'))
    def _flat_state(self):
        return {}

    def _load_flat_state(self, st):
        pass

    def state_dict(self):
        m = self._flat()
        if m is None:
            return super().state_dict()
        groups = [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]
        # the moment buckets are raw images of the flat bucket: its (name, offset, shape) layout travels with them, so that a checkpoint
        # written under another bucket order (another configuration or build) is re-ordered by name on load instead of loading wrongly
        layout = [[k, int(off), [int(x) for x in shape]] for k, (off, shape) in m._layout.items()]
        out = {"mmda_flat": 2, "step": int(m._step), "param_groups": groups, "layout": layout}
        out.update({k: v.detach().cpu().clone() for k, v in self._flat_state().items()})
        return out

    def load_state_dict(self, sd):
        m = self._flat()
        if not (isinstance(sd, dict) and sd.get("mmda_flat")):
            if m is not None:
                # a torch-format state (the reference writes one, solver.py:220): the attached fused step never reads self.state, so
                # the moments would be dropped silently -- scatter them into the flat buckets by parameter order instead
                return self._load_torch_state(sd, m)
            return super().load_state_dict(sd)
        if m is None:
            raise _lib.MMDAError("load_state_dict of a flat optimizer state needs the optimizer attached to a MISA model on the GPU")
        for g, saved in zip(self.param_groups, sd["param_groups"]):
            g.update(saved)
        m._step = int(sd["step"]); self._t = m._step
        mine = {k: (int(off), tuple(int(x) for x in shape)) for k, (off, shape) in m._layout.items()}
        saved = sd.get("layout")
        if saved is None:                                   # version 1: no layout recorded -- only safe when nothing else can differ
            for v in self._flat_state().values():
                for k, t in sd.items():
                    if torch.is_tensor(t) and t.numel() != v.numel():
                        raise _lib.MMDAError("flat optimizer state of another size (and without a layout map): cannot be loaded")
            return self._load_flat_state(sd)
        theirs = {k: (int(off), tuple(shape)) for k, off, shape in saved}
        if theirs == mine:
            return self._load_flat_state(sd)
        if set(theirs) != set(mine) or any(theirs[k][1] != mine[k][1] for k in mine):
            raise _lib.MMDAError("flat optimizer state belongs to a model with other parameters / shapes")
        # same parameters, another bucket order: move every tensor's moments to its offset here
        remapped = {}
        for name, img in sd.items():
            if not torch.is_tensor(img) or name not in self._flat_state():
                continue
            out = torch.zeros_like(img)
            for k, (off, shape) in mine.items():
                n = 1
                for x in shape:
                    n *= x
                o2 = theirs[k][0]
                out[off:off + n] = img[o2:o2 + n]
            remapped[name] = out
        self._load_flat_state(dict(sd, **remapped))

    def _load_torch_state(self, sd, m):
        names = [k for k, _ in m.named_parameters()]
        params = [p for g in self.param_groups for p in g["params"]]
        if len(sd.get("param_groups", [{}])[0].get("params", [])) != len(params):
            raise _lib.MMDAError("torch-format optimizer state with another number of parameters")
        by_id = {id(p): k for k, p in m.named_parameters()}
        flat = self._flat_state()
        keymap = {"exp_avg": "exp_avg", "exp_avg_sq": "exp_avg_sq", "square_avg": "square_avg"}
        step = 0
        for idx, p in zip(sd["param_groups"][0]["params"], params):
            st = sd["state"].get(idx)
            if not st:
                continue
            off, shape = m._layout[by_id[id(p)]]
            for src, dst in keymap.items():
                if src in st and dst in flat:
                    flat[dst][off:off + p.numel()].copy_(st[src].reshape(-1).to(flat[dst].device))
            step = max(step, int(st.get("step", 0)))
        for g, saved in zip(self.param_groups, sd["param_groups"]):
            g.update({k: v for k, v in saved.items() if k != "params"})
        m._step = step; self._t = step
        del names


class Adam(_FlatOptimizer):
    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=0, clip_value=None):
        if weight_decay != 0:
            raise NotImplementedError("the reference never passes weight_decay (config.py:143 is a dead flag)")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, clip_value=clip_value))

    def _flat_state(self):
        _, _, M, V = self._model.flat_buckets()
        return {"exp_avg": M, "exp_avg_sq": V}

    def _load_flat_state(self, sd):
        _, _, M, V = self._model.flat_buckets()
        M.copy_(sd["exp_avg"].to(M.device)); V.copy_(sd["exp_avg_sq"].to(V.device))

    @torch.no_grad()
    def step(self, closure=None, clip_value=None, grad_scale=1.0):
        lib = _lib.load()
        t = self._next_step()
        s = _lib.stream_ptr()
        g0 = self.param_groups[0]
        clip = clip_value if clip_value is not None else g0["clip_value"]
        clip = float("inf") if clip is None else float(clip)
        b1, b2 = g0["betas"]
        m = self._flat()
        if m is not None:
            P, G, M, V = m.flat_buckets()
            _lib.check(lib.mmda_clamp_adam(P.data_ptr(), G.data_ptr(), M.data_ptr(), V.data_ptr(), P.numel(), g0["lr"], b1, b2,
                                           g0["eps"], clip, grad_scale, t, s), "mmda_clamp_adam")
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
                                               group["lr"], b1, b2, group["eps"], clip, grad_scale, t, s), "mmda_clamp_adam")
        return None


class RMSprop(_FlatOptimizer):
    """torch.optim.RMSprop(params, lr) with torch's defaults (alpha 0.99, eps 1e-8, no momentum, not centered, no weight decay):
    the form the reference can construct (config.py:24, solver.py:97-99)."""

    def __init__(self, params, lr=1e-2, alpha=0.99, eps=1e-8, weight_decay=0, momentum=0, centered=False, clip_value=None):
        if weight_decay != 0 or momentum != 0 or centered:
            raise NotImplementedError("the reference passes lr only (solver.py:97-99)")
        super().__init__(params, dict(lr=lr, alpha=alpha, eps=eps, clip_value=clip_value))
        self._sq = None

    def _square_avg(self, like):
        if self._sq is None or self._sq.shape != like.shape or self._sq.device != like.device:
            self._sq = torch.zeros_like(like)
        return self._sq

    def _flat_state(self):
        return {"square_avg": self._square_avg(self._model.flat_buckets()[0])}

    def _load_flat_state(self, sd):
        sq = self._square_avg(self._model.flat_buckets()[0])
        sq.copy_(sd["square_avg"].to(sq.device))

    @torch.no_grad()
    def step(self, closure=None, clip_value=None, grad_scale=1.0):
        lib = _lib.load()
        self._next_step()
        s = _lib.stream_ptr()
        g0 = self.param_groups[0]
        clip = clip_value if clip_value is not None else g0["clip_value"]
        clip = float("inf") if clip is None else float(clip)
        m = self._flat()
        if m is not None:
            P, G, _, _ = m.flat_buckets()
            sq = self._square_avg(P)
            _lib.check(lib.mmda_clamp_rmsprop(P.data_ptr(), G.data_ptr(), sq.data_ptr(), P.numel(), g0["lr"], g0["alpha"], g0["eps"],
                                              clip, grad_scale, s), "mmda_clamp_rmsprop")
            return None
        for group in self.param_groups:
            for p in group["params"]:
                if p.grad is None:
                    continue
                if p.device.type != "cuda":
                    raise _lib.MMDAError("mmda_amd.optim.RMSprop steps on the GPU only")
                st = self.state[p]
                if not st:
                    st["square_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                g = p.grad.contiguous()
                _lib.check(lib.mmda_clamp_rmsprop(p.data_ptr(), g.data_ptr(), st["square_avg"].data_ptr(), p.numel(), group["lr"],
                                                  group["alpha"], group["eps"], clip, grad_scale, s), "mmda_clamp_rmsprop")
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


optimizer_dict = {"RMSprop": RMSprop, "Adam": Adam}       # reference config.py:24
