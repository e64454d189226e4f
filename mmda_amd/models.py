"""MISA model with the reference's call surface, computed by hand-written HIP kernels (gfx950).

Mirrors reference ``src/models.py:15-285``: same constructor (``MISA(config)``), same ``forward`` signature and return
value, same post-forward side-channel attributes the solver reads (``utt_shared_*``, ``utt_private_*``,
``utt_*_orig``, ``utt_*_recon``, ``domain_label_*``, ``tcp``), same ``state_dict`` keys.  ``Model`` is an alias
(BASELINE.json's north star names it so).

What is different by design: there are no ``nn.LSTM``/``nn.Linear`` submodules.  Every parameter is a view into ONE
flat fp32 device bucket (gradients and Adam moments likewise), the layout of which is defined by the native runtime
(``mmda_misa_param_info``); all arithmetic happens in ``libmmda_hip.so`` through the C ABI in ``include/mmda_hip.h``.
PyTorch only owns the memory and the autograd tape entry.  There is no CPU fallback: forward on a CPU tensor raises.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Dict, List, Tuple

import torch
import torch.nn as nn

from . import _lib
from .config import activation_name

FFN_DIM = 2048          # torch default dim_feedforward of nn.TransformerEncoderLayer (reference models.py:160)
FUSION_DROPOUT = 0.1    # torch default dropout of nn.TransformerEncoderLayer (not a reference flag)

_TOP_ORDER = ["activation", "embed", "trnn1", "trnn2", "vrnn1", "vrnn2", "arnn1", "arnn2", "project_t", "project_v", "project_a",
              "private_t", "private_v", "private_a", "shared", "recon_t", "recon_v", "recon_a", "discriminator",
              "sp_discriminator", "confidence", "classifier", "tlayer_norm", "vlayer_norm", "alayer_norm",
              "transformer_encoder"]
_RNN_ORDER = ["weight_ih_l0", "weight_hh_l0", "bias_ih_l0", "bias_hh_l0", "weight_ih_l0_reverse", "weight_hh_l0_reverse",
              "bias_ih_l0_reverse", "bias_hh_l0_reverse"]

# outputs of the autograd entry, in order (labels last, non-differentiable)
_PUB = ["scores", "tcp", "utt_t_orig", "utt_v_orig", "utt_a_orig", "utt_private_t", "utt_private_v", "utt_private_a",
        "utt_shared_t", "utt_shared_v", "utt_shared_a", "utt_t_recon", "utt_v_recon", "utt_a_recon",
        "domain_label_t", "domain_label_v", "domain_label_a"]


_SIG_CACHE: Dict[int, bool] = {}


def _takes_model(fn) -> bool:
    """True if ``fn`` accepts a third positional argument (the model)."""
    import inspect
    key = id(getattr(fn, "__func__", fn))
    if key not in _SIG_CACHE:
        try:
            ps = list(inspect.signature(fn).parameters.values())
            npos = sum(p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD) for p in ps)
            _SIG_CACHE[key] = npos >= 3 or any(p.kind == p.VAR_POSITIONAL for p in ps)
        except (TypeError, ValueError):
            _SIG_CACHE[key] = True
    return _SIG_CACHE[key]


class _Bag(nn.Module):
    """Parameter container; exists only to reproduce the reference's dotted state_dict names."""


def _reference_sort_key(name: str):
    top = name.split(".")[0]
    rest = name[len(top) + 1:]
    sub = _RNN_ORDER.index(rest) if rest in _RNN_ORDER else 0
    return (_TOP_ORDER.index(top), sub)


class MISA(nn.Module):
    """MISA for CMU-MOSEI emotion multi-label classification (reference models.py:15-17), HIP-backed."""

    def __init__(self, config):
        super().__init__()
        self.config = config
        self.text_size = config.embedding_size
        self.visual_size = config.visual_size
        self.acoustic_size = config.acoustic_size
        self.input_sizes = [self.text_size, self.visual_size, self.acoustic_size]
        self.hidden_sizes = [int(self.text_size), int(self.visual_size), int(self.acoustic_size)]
        self.output_size = config.num_classes
        self.dropout_rate = config.dropout
        if getattr(config, "extractor", "lstm") == "transformer":
            # the reference prints a TODO and calls exit() here (models.py:33-36)
            raise NotImplementedError("extractor='transformer' is a TODO in the reference as well")
        # reference models.py:39: nn.LSTM if config.rnncell == 'lstm' else nn.GRU
        self.rnncell = "lstm" if getattr(config, "rnncell", "lstm") == "lstm" else "gru"
        if getattr(config, "use_bert", False):
            raise NotImplementedError("use_bert=True needs a hub download; the GloVe/LSTM text branch is the hot path")
        self.activation_name = activation_name(config.activation)
        self.precision = getattr(config, "precision", "bf16")
        if self.precision not in ("bf16", "fp32"):
            raise ValueError("config.precision must be 'bf16' or 'fp32'")

        lib = _lib.load()
        cc = _lib.MisaConfig(
            vocab=len(config.word2id), d_t=self.text_size, d_v=self.visual_size, d_a=self.acoustic_size,
            hidden=config.hidden_size, ncls=config.num_classes, act=_lib.ACT[self.activation_name],
            use_cmd_sim=int(bool(config.use_cmd_sim)), use_confidNet=int(bool(getattr(config, "use_confidNet", False))),
            dropout=float(config.dropout), fusion_dropout=FUSION_DROPOUT, threshold=float(config.threshold),
            reverse_grad_weight=float(getattr(config, "reverse_grad_weight", 1.0)),
            diff_weight=float(getattr(config, "diff_weight", 0.3)), sim_weight=float(getattr(config, "sim_weight", 0.7)),
            recon_weight=float(getattr(config, "recon_weight", 0.7)), conf_weight=float(getattr(config, "conf_weight", 0.3)),
            mode=_lib.BF16 if self.precision == "bf16" else _lib.F32, rnncell=_lib.CELL[self.rnncell])
        h = C.c_void_p()
        _lib.check(lib.mmda_misa_create(C.byref(cc), C.byref(h)), "mmda_misa_create")
        self._h = h
        self._lib = lib
        # BASELINE configs[4]: the fusion layer's feed-forward products on block-scaled fp8 (forward only; off by default)
        self.fusion_fp8 = bool(getattr(config, "fusion_fp8", False))
        if self.fusion_fp8:
            _lib.check(lib.mmda_misa_set_fusion_fp8(h, 1), "set_fusion_fp8")
        self._layout: Dict[str, Tuple[int, Tuple[int, ...]]] = {}
        for i in range(lib.mmda_misa_num_params(h)):
            name, off, rows, cols = C.c_char_p(), C.c_int64(), C.c_int(), C.c_int()
            _lib.check(lib.mmda_misa_param_info(h, i, C.byref(name), C.byref(off), C.byref(rows), C.byref(cols)))
            shape = (rows.value, cols.value) if cols.value > 0 else (rows.value,)
            self._layout[name.value.decode()] = (off.value, shape)
        self._flat_floats = lib.mmda_misa_flat_floats(h)
        self._dense_floats = lib.mmda_misa_dense_floats(h)
        self._names: List[str] = sorted(self._layout, key=_reference_sort_key)
        # config.activation = prelu: the reference instantiates ONE nn.PReLU() (models.py:30) and adds that module to the three
        # projections and to the discriminator, so its slope shows up in state_dict() under every one of those names: aliases of the
        # single native parameter "activation.weight", registered as the same nn.Parameter
        self._aliases: Dict[str, str] = {}
        if "activation.weight" in self._layout:
            after = {f"project_{m_}.project_{m_}.bias": f"project_{m_}.project_{m_}_activation.weight" for m_ in "tva"}
            after["discriminator.discriminator_layer_1.bias"] = "discriminator.discriminator_layer_1_activation.weight"
            names = []
            for n_ in self._names:
                names.append(n_)
                if n_ in after:
                    names.append(after[n_])
                    self._aliases[after[n_]] = "activation.weight"
            self._names = names
        for name in self._names:
            if name in self._aliases:
                self._layout[name] = self._layout[self._aliases[name]]
                self._register(name, None, shared=self._get(self._aliases[name]))
            else:
                self._register(name, self._layout[name][1])
        self._plist = [(n, self._get(n)) for n in self._names]
        self.reset_parameters()

        # device state (created lazily on the first forward / .to())
        self._P = self._G = self._M = self._V = None
        self._ws = None
        self._ws_shape = None
        self._len_dev = None
        self._len_cache = None
        self._fwd_id = 0
        self._step = 0
        self._seed = 0x5EED
        self._anchor = None
        self._last = {}
        self._abort_seen = False
        self._seg_work = None

    # ------------------------------------------------------------------ parameters
    def _register(self, dotted: str, shape, shared=None):
        mod = self
        parts = dotted.split(".")
        for p in parts[:-1]:
            if not hasattr(mod, p):
                mod.add_module(p, _Bag())
            mod = getattr(mod, p)
        mod.register_parameter(parts[-1], shared if shared is not None else nn.Parameter(torch.empty(shape, dtype=torch.float32)))

    def _get(self, dotted: str) -> nn.Parameter:
        mod = self
        for p in dotted.split("."):
            mod = getattr(mod, p)
        return mod

    @torch.no_grad()
    def reset_parameters(self):
        """Same distributions as the torch modules the reference instantiates (nn.LSTM, nn.Linear, nn.LayerNorm,
        nn.Embedding, nn.MultiheadAttention); the reference's solver then applies orthogonal_ to weight_hh*."""
        for name, p in self._plist:
            if name.endswith("activation.weight"):
                p.fill_(0.25)                               # nn.PReLU() default
            elif name == "embed.weight":
                p.normal_(0.0, 1.0)
            elif "rnn" in name.split(".")[0]:
                h = self._layout[name.rsplit(".", 1)[0] + ".weight_hh_l0"][1][1]
                k = 1.0 / math.sqrt(h)
                p.uniform_(-k, k)
            elif "layer_norm" in name or ".norm1." in name or ".norm2." in name:
                p.fill_(1.0 if name.endswith("weight") else 0.0)
            elif name.endswith("in_proj_weight"):
                nn.init.xavier_uniform_(p)
            elif name.endswith("in_proj_bias") or name.endswith("out_proj.bias"):
                p.zero_()
            elif name.endswith("weight"):
                k = 1.0 / math.sqrt(p.shape[1])
                p.uniform_(-k, k)
            else:   # Linear bias: U(-1/sqrt(fan_in), 1/sqrt(fan_in))
                w = self._get(name[:-4] + "weight")
                k = 1.0 / math.sqrt(w.shape[1])
                p.uniform_(-k, k)

    def _apply(self, fn, *args, **kwargs):
        out = super()._apply(fn, *args, **kwargs)
        self._P = None           # parameter storages were replaced: re-flatten lazily
        return out

    def _views_valid(self) -> bool:
        if self._P is None:
            return False
        base = self._P.data_ptr()
        for name, p in self._plist:
            if p.data_ptr() != base + 4 * self._layout[name][0]:
                return False
        return True

    @torch.no_grad()
    def _materialize(self, device):
        """(Re)build the flat device buckets and point every Parameter at its slice."""
        if device.type != "cuda":
            raise _lib.MMDAError("mmda_amd.MISA runs on an MI355X only (no CPU fallback); move the model with .to('cuda')")
        P = torch.zeros(self._flat_floats, dtype=torch.float32, device=device)
        for name, p in self._plist:
            off, shape = self._layout[name]
            n = p.numel()
            P[off:off + n].copy_(p.data.reshape(-1))
            p.data = P[off:off + n].view(shape)
        self._P = P
        if self._G is None or self._G.device != device:
            self._G = torch.zeros_like(P)
            self._M = torch.zeros_like(P)
            self._V = torch.zeros_like(P)
        for name, p in self._plist:
            off, shape = self._layout[name]
            p.grad = None
        _lib.check(self._lib.mmda_misa_bind(self._h, P.data_ptr(), self._G.data_ptr(), self._M.data_ptr(), self._V.data_ptr()),
                   "mmda_misa_bind")
        self._ws_shape = None

    def _assign_grad_views(self):
        for name, p in self._plist:
            if p.grad is None:
                off, shape = self._layout[name]
                p.grad = self._G[off:off + p.numel()].view(shape)

    def zero_grad(self, set_to_none: bool = False):
        """Zeroes the flat gradient bucket with one memset (the reference calls model.zero_grad() per batch,
        solver.py:139).  Gradients stay views of the bucket."""
        if self._G is not None and self._P is not None:
            _lib.check(self._lib.mmda_misa_zero_grad(self._h, _lib.stream_ptr()), "zero_grad")
        else:
            super().zero_grad(set_to_none=True)

    # ------------------------------------------------------------------ device plumbing
    def _prepare(self, sentences, video, acoustic, lengths):
        dev = sentences.device
        if dev.type != "cuda":
            raise _lib.MMDAError("inputs are on the CPU: mmda_amd.MISA has no CPU path (HIP kernels only)")
        if not self._views_valid():
            self._materialize(dev)
        T, B = sentences.shape
        if video.shape[0] != T or video.shape[1] != B or acoustic.shape[0] != T or acoustic.shape[1] != B:
            raise ValueError("sentences/video/acoustic must share (T, B)")
        if video.shape[2] != self.visual_size or acoustic.shape[2] != self.acoustic_size:
            raise ValueError("feature width does not match config.visual_size / acoustic_size")
        lens = torch.as_tensor(lengths)
        if lens.numel() != B:
            raise ValueError("lengths must have B entries")
        lmin, lmax = int(lens.min()), int(lens.max())
        if lmin <= 0:
            raise RuntimeError("Length of all samples has to be greater than 0")     # pack_padded_sequence's rule
        if lmax > T:
            raise RuntimeError("a length exceeds the padded sequence length")
        if self._ws_shape != (B, T):
            need = self._lib.mmda_misa_workspace_floats(self._h, B, T)
            if self._ws is None or self._ws.numel() < need or self._ws.device != dev:
                if self._ws is not None and self._ws_shape is not None:
                    # the old buffer's sticky abort words would be lost with it (the native side never touches a buffer it was
                    # not handed): look at them first.  Buffers only grow, so this synchronous read happens a few times per run.
                    self._abort_seen = self._abort_seen or self.cluster_aborted()
                self._ws = torch.zeros(need, dtype=torch.float32, device=dev)
            _lib.check(self._lib.mmda_misa_set_workspace_async(self._h, self._ws.data_ptr(), self._ws.numel(), B, T, _lib.stream_ptr()),
                       "set_workspace")
            self._ws_shape = (B, T)
        # lengths arrive on the CPU (reference: l = to_cpu(l), solver.py:149).  Convert on the host, stage through a pinned
        # buffer and copy asynchronously; an unchanged batch (benchmark loops) reuses the device copy.
        lens32 = lens.to(device="cpu", dtype=torch.int32)
        cached = self._len_cache
        if cached is not None and cached[0].shape == lens32.shape and torch.equal(cached[0], lens32) and cached[1].device == dev:
            len_dev = cached[1]
        else:
            pin = torch.empty(B, dtype=torch.int32, pin_memory=True)
            pin.copy_(lens32)
            len_dev = pin.to(device=dev, non_blocking=True)
            self._len_cache = (lens32.clone(), len_dev, pin)
        t = sentences.contiguous()
        if t.dtype != torch.int64:
            t = t.long()
        v = video.contiguous().float()
        a = acoustic.contiguous().float()
        return t, v, a, len_dev

    def _off(self, name: str) -> int:
        o = self._lib.mmda_misa_tensor_offset(self._h, name.encode())
        if o < 0:
            raise KeyError(name)
        return o

    def _ws_view(self, name: str, shape):
        n = 1
        for s in shape:
            n *= s
        o = self._off(name)
        return self._ws[o:o + n].view(shape)

    def _public(self) -> Dict[str, torch.Tensor]:
        """Views (into the workspace) of everything the reference's solver reads after a forward."""
        B, _ = self._ws_shape
        hs, nc = self.config.hidden_size, self.config.num_classes
        x6 = self._ws_view("x6", (6, B, hs))
        orig = self._ws_view("orig", (3, B, hs))
        recon = self._ws_view("recon", (3, B, hs))
        out = {"scores": self._ws_view("scores", (B, nc)), "tcp": self._ws_view("tcp", (B, 6)),
               "labels": self._ws_view("labels", (B, nc))}
        for i, m in enumerate("tva"):
            out[f"utt_{m}_orig"] = orig[i]
            out[f"utt_private_{m}"] = x6[i]
            out[f"utt_shared_{m}"] = x6[3 + i]
            out[f"utt_{m}_recon"] = recon[i]
        if not self.config.use_cmd_sim:
            dom = self._ws_view("dom", (3, B, 3))
            for i, m in enumerate("tva"):
                out[f"domain_label_{m}"] = dom[i]
        return out

    def _grad_slots(self) -> Dict[str, torch.Tensor]:
        B, _ = self._ws_shape
        hs, nc = self.config.hidden_size, self.config.num_classes
        dx6 = self._ws_view("d_x6", (6, B, hs))
        dorig = self._ws_view("d_orig", (3, B, hs))
        drec = self._ws_view("d_recon", (3, B, hs))
        out = {"scores": self._ws_view("d_scores", (B, nc)), "tcp": self._ws_view("d_tcp", (B, 6))}
        for i, m in enumerate("tva"):
            out[f"utt_{m}_orig"] = dorig[i]
            out[f"utt_private_{m}"] = dx6[i]
            out[f"utt_shared_{m}"] = dx6[3 + i]
            out[f"utt_{m}_recon"] = drec[i]
        if not self.config.use_cmd_sim:
            ddom = self._ws_view("d_dom", (3, B, 3))
            for i, m in enumerate("tva"):
                out[f"domain_label_{m}"] = ddom[i]
        return out

    def _next_seed(self) -> int:
        self._seed = (self._seed * 6364136223846793005 + 1442695040888963407) & 0xFFFFFFFFFFFFFFFF
        return self._seed

    def _forward_raw(self, t, v, a, len_dev, training: bool, seed: int, inference: bool = False):
        # inference: a forward that no backward will follow (torch.no_grad()): no stash, no backward-only operand copies
        _lib.check(self._lib.mmda_misa_set_inference(self._h, int(inference)), "set_inference")
        _lib.check(self._lib.mmda_misa_forward(self._h, t.data_ptr(), v.data_ptr(), a.data_ptr(), len_dev.data_ptr(),
                                               int(training), seed, _lib.stream_ptr()), "mmda_misa_forward")
        self._fwd_id += 1
        self._last = dict(t=t, v=v, a=a, len_dev=len_dev)

    # ------------------------------------------------------------------ reference call surface
    def alignment(self, sentences, visual, acoustic, lengths, bert_sent=None, bert_sent_type=None, bert_sent_mask=None):
        """reference models.py:182-250 (whole encoder + fusion + heads in one native call)."""
        t, v, a, len_dev = self._prepare(sentences, visual, acoustic, lengths)
        seed = self._next_seed()
        if torch.is_grad_enabled():
            if self._anchor is None or self._anchor.device != t.device:
                self._anchor = torch.zeros(1, device=t.device, requires_grad=True)
            outs = _MISAFn.apply(self._anchor, self, t, v, a, len_dev, self.training, seed)
            named = dict(zip(_PUB, outs[:-1]))
            labels = outs[-1]
        else:
            self._forward_raw(t, v, a, len_dev, self.training, seed, inference=True)
            pub = self._public()
            named = {k: pub[k].clone() if k in pub else None for k in _PUB}
            labels = pub["labels"].clone()
        for k in _PUB:
            if k in ("scores",):
                continue
            val = named.get(k)
            if k.startswith("domain_label") and self.config.use_cmd_sim:
                val = None
            setattr(self, k, val)
        return named["scores"], labels

    def forward(self, sentences, video, acoustic, lengths, bert_sent=None, bert_sent_type=None, bert_sent_mask=None):
        """reference models.py:282-285: returns (predicted_scores (B,6), predicted_labels (B,6) in {0,1})."""
        return self.alignment(sentences, video, acoustic, lengths, bert_sent, bert_sent_type, bert_sent_mask)

    # written-but-unread attributes of the reference (models.py:234-237,256-258), materialised on demand with the HIP GEMM
    @property
    def utt_t(self):
        return self.utt_private_t + self.utt_shared_t

    @property
    def utt_v(self):
        return self.utt_private_v + self.utt_shared_v

    @property
    def utt_a(self):
        return self.utt_private_a + self.utt_shared_a

    # sp_discriminator outputs (models.py:234-237): the reference computes them every forward and never reads them (no loss uses
    # them), so the hot path skips the dead GEMMs; the attributes are materialised on demand on the HIP GEMM.
    def _sp_disc(self, x):
        from . import ops
        w = self.sp_discriminator.sp_discriminator_layer_1.weight
        b = self.sp_discriminator.sp_discriminator_layer_1.bias
        return ops.gemm(x.detach().contiguous(), w.detach(), mode="fp32", bias=b.detach())

    @property
    def shared_or_private_p_t(self):
        return self._sp_disc(self.utt_private_t)

    @property
    def shared_or_private_p_v(self):
        return self._sp_disc(self.utt_private_v)

    @property
    def shared_or_private_p_a(self):
        return self._sp_disc(self.utt_private_a)

    @property
    def shared_or_private_s(self):
        return self._sp_disc((self.utt_shared_t + self.utt_shared_v + self.utt_shared_a) / 3.0)

    # ------------------------------------------------------------------ fused fast path (Solver.train_epoch)
    def train_step(self, sentences, video, acoustic, lengths, emo_label, lr: float, clip: float, do_adam: bool = True,
                   training: bool = True, seed=None, grad_sync=None, optimizer=None):
        """One reference loop iteration (solver.py:139-186) in native code: zero_grad, forward, six losses, backward,
        clip + Adam.  ``grad_sync(flat_grad_bucket, dense_floats, model)`` is called between backward and Adam for the
        data-parallel all-reduce (mmda_amd/dist.py) and must return the gradient scale (1/world).
        ``optimizer``: an mmda_amd.optim optimizer attached to this model.  Adam (or None) is stepped by the native fused
        clamp+Adam with ``lr``; any other (RMSprop, config.py:24) by its own fused kernel after the gradient exchange.
        Losses stay on the device (read them with ``read_losses()``; one sync, not six)."""
        from . import optim as _optim
        t, v, a, len_dev = self._prepare(sentences, video, acoustic, lengths)
        emo = emo_label.to(device=t.device, dtype=torch.float32).contiguous()
        if seed is None:
            seed = self._next_seed()
        custom = do_adam and optimizer is not None and not isinstance(optimizer, _optim.Adam)
        if not custom:
            self._step += 1                        # (a custom optimizer counts its own steps on the same counter)
        s = _lib.stream_ptr()
        fused_adam = do_adam and grad_sync is None and not custom
        gs_owner = getattr(grad_sync, "__self__", None)
        if gs_owner is not None and getattr(gs_owner, "global_stats", False) and (gs_owner.world > 1 or gs_owner.force_collectives):
            self._global_stats_step(t, v, a, len_dev, emo, training, seed, gs_owner)
        else:
            _lib.check(self._lib.mmda_misa_train_step(self._h, t.data_ptr(), v.data_ptr(), a.data_ptr(), len_dev.data_ptr(),
                                                      emo.data_ptr(), int(training), seed, int(fused_adam), lr, clip, max(self._step, 1), s),
                       "mmda_misa_train_step")
            self._fwd_id += 1
        self._last = dict(t=t, v=v, a=a, len_dev=len_dev, emo=emo)
        if custom:
            scale = 1.0
            if grad_sync is not None:
                scale = grad_sync(self._G, self._dense_floats, self) if _takes_model(grad_sync) else grad_sync(self._G, self._dense_floats)
            optimizer.step(clip_value=clip, grad_scale=float(scale))
        elif do_adam and grad_sync is not None:
            # (bucket, dense_floats, model) or a plain (bucket, dense_floats) callable: decided from its signature, once -- never by
            # retrying after a TypeError, which could come from inside the exchange after a collective was already issued
            # A DataParallelSync (mmda_amd/dist.py) steps the early-reduced prefix of the bucket on its communication stream, beside the
            # rest of the backward pass; the remainder is stepped here, behind the exchange.
            owner = getattr(grad_sync, "__self__", None)
            hook = owner is not None and hasattr(owner, "early_step") and hasattr(owner, "early_stepped") and hasattr(owner, "world")
            if hook:
                gs = 1.0 / float(owner.world)
                step_no = max(self._step, 1)

                def _early(n_floats, stream, _gs=gs, _k=step_no):
                    _lib.check(self._lib.mmda_clamp_adam(self._P.data_ptr(), self._G.data_ptr(), self._M.data_ptr(), self._V.data_ptr(),
                                                         int(n_floats), lr, 0.9, 0.999, 1e-8, clip, _gs, _k, stream.cuda_stream), "adam(early)")
                owner.early_step = _early
            try:
                if _takes_model(grad_sync):
                    scale = grad_sync(self._G, self._dense_floats, self)
                else:
                    scale = grad_sync(self._G, self._dense_floats)
            finally:
                if hook:
                    owner.early_step = None
            done = int(owner.early_stepped) if hook else 0
            if done > 0:
                n = self._P.numel() - done
                o = done * 4
                _lib.check(self._lib.mmda_clamp_adam(self._P.data_ptr() + o, self._G.data_ptr() + o, self._M.data_ptr() + o,
                                                     self._V.data_ptr() + o, n, lr, 0.9, 0.999, 1e-8, clip, float(scale), max(self._step, 1), s),
                           "adam(rest)")
            else:
                _lib.check(self._lib.mmda_misa_adam_step(self._h, lr, clip, float(scale), self._step, s), "adam_step")

    def _global_stats_step(self, t, v, a, len_dev, emo, training: bool, seed: int, dp) -> None:
        """forward + losses + backward of one step with the batch-statistic losses on the batch of ALL ranks (DataParallelSync
        global_stats=True).  Reference: on one device DiffLoss (utils/functions.py:64-76), CMD (:89-108) and the
        confidence loss (solver.py:451-462) see the whole batch; here every rank gathers the (6, B, hs) private / shared utterance
        vectors -- and scores, tcp, labels for the confidence loss -- of all ranks, runs the SAME loss entry points on the gathered
        batch, keeps the loss sums and adds ITS rows of the gradients, times the world size (the gradient exchange averages over
        ranks; cls and recon are means over samples, for which the average of the shard gradients already is the global gradient).
        Equal batch shapes on all ranks."""
        lib, h, cfg = self._lib, self._h, self.config
        if not cfg.use_cmd_sim:
            raise NotImplementedError("global_stats: the CMD similarity branch only (config.use_cmd_sim)")
        s = _lib.stream_ptr()
        W, r = int(dp.world), int(dp.rank)
        _lib.check(lib.mmda_misa_zero_grad(h, s), "zero_grad")
        self._forward_raw(t, v, a, len_dev, training, seed, inference=False)
        _lib.check(lib.mmda_misa_zero_act_grads(h, s), "zero_act_grads")
        B, _ = self._ws_shape
        hs, nc = int(cfg.hidden_size), int(cfg.num_classes)
        Bg = W * B
        L = self._ws_view("losses", (8,))
        X = dp.gather_rows(self._ws_view("x6", (6, B, hs)), dim=1)                    # (6, W B, hs): rank r at rows [r B, (r + 1) B)
        dX = torch.zeros_like(X)
        work = torch.empty(int(lib.mmda_loss_diff_work_floats(Bg, hs)), dtype=torch.float32, device=X.device)
        _lib.check(lib.mmda_loss_diff(X.data_ptr(), Bg * hs, Bg, hs, float(getattr(cfg, "diff_weight", 0.3)), L.data_ptr() + 4, dX.data_ptr(),
                                      work.data_ptr(), s), "loss_diff(global)")
        _lib.check(lib.mmda_loss_cmd(X[3:].data_ptr(), Bg * hs, Bg, hs, float(getattr(cfg, "sim_weight", 0.7)), L.data_ptr() + 8,
                                     dX[3:].data_ptr(), s), "loss_cmd(global)")
        self._ws_view("d_x6", (6, B, hs)).add_(dX[:, r * B:(r + 1) * B], alpha=float(W))
        if nc == 6:
            S = dp.gather_rows(self._ws_view("scores", (B, nc)))
            Tc = dp.gather_rows(self._ws_view("tcp", (B, 6)))
            E = dp.gather_rows(emo)
            with_g = bool(getattr(cfg, "use_confidNet", False))
            dS = torch.zeros_like(S) if with_g else None
            dT = torch.zeros_like(Tc) if with_g else None
            _lib.check(lib.mmda_loss_conf(S.data_ptr(), Tc.data_ptr(), E.data_ptr(), Bg, nc, float(getattr(cfg, "conf_weight", 0.3)),
                                          L.data_ptr() + 16, dS.data_ptr() if with_g else None, dT.data_ptr() if with_g else None, s),
                       "loss_conf(global)")
            if with_g:
                self._ws_view("d_scores", (B, nc)).add_(dS[r * B:(r + 1) * B], alpha=float(W))
                self._ws_view("d_tcp", (B, 6)).add_(dT[r * B:(r + 1) * B], alpha=float(W))
        _lib.check(lib.mmda_misa_set_external_batch_losses(h, 1), "set_external_batch_losses")
        try:
            _lib.check(lib.mmda_misa_losses(h, emo.data_ptr(), 1, s), "mmda_misa_losses")
        finally:
            lib.mmda_misa_set_external_batch_losses(h, 0)
        _lib.check(lib.mmda_misa_backward(h, t.data_ptr(), v.data_ptr(), a.data_ptr(), len_dev.data_ptr(), s), "mmda_misa_backward")
        # (keep the gathered tensors alive until the stream has run the launches that read them)
        self._gs_keep = (X, dX, work)

    # ------------------------------------------------------------------ early part of the gradient bucket (data parallel)
    def early_grad_floats(self) -> int:
        """Length of the gradient-bucket prefix (fusion block, LayerNorms, layer-2 recurrent layers) whose gradients are final
        beside the layer-1 backward recurrence of the step that was just issued; 0 if nothing is early."""
        return int(self._lib.mmda_misa_early_grad_floats(self._h))

    def wait_early_grads(self, stream) -> None:
        """Make ``stream`` (a torch.cuda.Stream) wait for the event after which that prefix may be read."""
        _lib.check(self._lib.mmda_misa_wait_early_grads(self._h, stream.cuda_stream), "wait_early_grads")

    # ------------------------------------------------------------------ sparse view of the embedding gradient (data parallel)
    def embedding_grad_rows(self):
        """(ids (R,) int64, rows (R, d_t) fp32) of the last backward: embed.weight.grad == sum over the list of rows[p] into row
        ids[p].  The dense gradient is non-zero in at most R = T*B of its V rows, so data-parallel ranks exchange these instead of
        V x d_t.  Positions past a sample's length carry id -1 (their rows are exactly zero: no gradient flows through padding)."""
        t = self._last["t"]
        T, B = t.shape
        R = T * B
        pad = torch.arange(T, device=t.device, dtype=torch.int32).unsqueeze(1) >= self._last["len_dev"].unsqueeze(0)
        ids = torch.where(pad, torch.full_like(t, -1), t)
        return ids.reshape(R), self._ws_view("d_x_t", (R, self._layout["embed.weight"][1][1]))

    def embedding_rows_per_step(self):
        """T * B of the last step: the rows the (ids, rows) form of the embedding gradient holds (None before the first step)."""
        return None if not getattr(self, "_last", None) else int(self._last["t"].numel())

    def embedding_table_rows(self):
        """V: the rows a dense exchange of embed.weight.grad moves."""
        return int(self._layout["embed.weight"][1][0])

    def scatter_embedding_rows(self, ids: torch.Tensor, rows: torch.Tensor):
        """embed.weight.grad[ids] += rows with the native atomic scatter-add (summation order not fixed; ids < 0 not allowed)."""
        if ids.numel() == 0:
            return
        off, (V, D) = self._layout["embed.weight"]
        g = self._G[off:off + V * D]
        _lib.check(self._lib.mmda_embed_scatter_add(g.data_ptr(), ids.contiguous().data_ptr(), ids.numel(), D,
                                                    rows.contiguous().data_ptr(), _lib.stream_ptr()), "embed_scatter_add")

    def set_embedding_grad_rows(self, ids: torch.Tensor, rows: torch.Tensor):
        """embed.weight.grad[id] = sum of rows[p] over the positions with ids[p] == id, added in list order by the native
        deterministic segment sum (rows of ids that occur are overwritten, ids < 0 skipped): what every data-parallel rank runs on
        the same all-gathered list, so that replicas stay bit-identical."""
        n = ids.numel()
        if n == 0:
            return
        off, (V, D) = self._layout["embed.weight"]
        g = self._G[off:off + V * D]
        need = int(self._lib.mmda_embed_segment_sum_work_bytes(n, D))
        if self._seg_work is None or self._seg_work.numel() < need or self._seg_work.device != g.device:
            self._seg_work = torch.empty(need, dtype=torch.uint8, device=g.device)
        _lib.check(self._lib.mmda_embed_segment_sum(g.data_ptr(), ids.contiguous().data_ptr(), n, D, rows.contiguous().data_ptr(),
                                                    self._seg_work.data_ptr(), self._seg_work.numel(), _lib.stream_ptr()),
                   "embed_segment_sum")

    def read_losses(self) -> Dict[str, float]:
        """cls, diff, sim, recon, conf, total of the last losses pass (one device->host sync)."""
        L = self._ws_view("losses", (8,)).tolist()
        return dict(cls=L[0], diff=L[1], sim=L[2], recon=L[3], conf=L[4], total=L[5])

    def flat_buckets(self):
        """(params, grads, adam_m, adam_v) flat fp32 device tensors; the first `dense_floats` entries exclude embed.weight."""
        return self._P, self._G, self._M, self._V

    @property
    def dense_floats(self) -> int:
        return self._dense_floats

    def cluster_aborted(self) -> bool:
        """True if a resident-weights recurrence ever timed out waiting for its cluster (results are invalid after that).
        Synchronous device->host read: call it off the step path."""
        import ctypes
        if self._ws is None or self._ws_shape is None:
            return self._abort_seen
        flag = ctypes.c_int(0)
        _lib.check(self._lib.mmda_misa_cluster_status(self._h, ctypes.byref(flag)), "cluster_status")
        self._abort_seen = self._abort_seen or bool(flag.value)
        self._abort_bits = getattr(self, "_abort_bits", 0) | int(flag.value)
        return self._abort_seen

    def check_cluster(self, where: str = ""):
        """Raises MMDAError if a recurrence ever gave up waiting for its cluster (every result since then is invalid).
        One synchronous device->host read: Solver calls it once per epoch / evaluation pass and before saving a checkpoint."""
        if self.cluster_aborted():
            # The hand-off tags every published bf16 value with its epoch in a spare bit: a NaN / an overflowing activation reads as a tag
            # that never matches, so a DIVERGED run ends here too (the reference would print NaN losses).  Say which one it was.
            nonfinite = False
            try:
                L = self._ws_view("losses", (8,)).tolist()
                nonfinite = any(x != x or abs(x) == float("inf") for x in L[:6]) or not bool(torch.isfinite(self._P).all())
            except Exception:
                pass
            what = ("non-finite values reached a recurrence (diverged run: NaN / inf in the losses or parameters), which its data-tagged "
                    "hand-off reports as a cluster time-out" if nonfinite else
                    "a resident-weights recurrence timed out waiting for its workgroup cluster")
            if not nonfinite and (getattr(self, "_abort_bits", 0) & 2):
                what = ("a kernel of the fused training step timed out waiting on the device for the side stream's chain (flag join): are "
                        "kernel dispatches being serialised (profiler counters)?  Run such tools with MMDA_FLAG_JOIN=0 MMDA_SORT_EARLY=0")
            raise _lib.MMDAError(what + (f" ({where})" if where else "") + ": results since then are invalid")

    def set_recurrence(self, resident_weights: bool):
        """bf16 recurrences: W_hh resident in LDS across a workgroup cluster (default) or streamed from L2 per step."""
        _lib.check(self._lib.mmda_misa_set_recurrence(self._h, int(resident_weights)), "set_recurrence")

    def set_gemm_operands(self, bf16_copies: bool):
        """bf16 mode: LSTM-sized GEMMs on bf16 operand copies (default) or on fp32 tensors staged through the generic kernel."""
        _lib.check(self._lib.mmda_misa_set_gemm_operands(self._h, int(bf16_copies)), "set_gemm_operands")

    def set_fusion_fp8(self, on: bool):
        """Forward feed-forward products of the fusion transformer layer (linear1 / linear2, reference models.py:160-161) on
        block-scaled fp8 MFMA operands; the backward pass stays exact (straight-through)."""
        self.fusion_fp8 = bool(on)
        _lib.check(self._lib.mmda_misa_set_fusion_fp8(self._h, int(on)), "set_fusion_fp8")

    def set_precision(self, precision: str):
        self.precision = precision
        _lib.check(self._lib.mmda_misa_set_mode(self._h, _lib.BF16 if precision == "bf16" else _lib.F32), "set_mode")

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                self._lib.mmda_misa_destroy(self._h)
                self._h = None
        except Exception:
            pass


class _MISAFn(torch.autograd.Function):
    """Autograd tape entry for the compat path: forward and backward are single native calls; the anchor input only
    exists so that the outputs require grad.  Parameter gradients are ACCUMULATED into the flat bucket (and exposed as
    ``p.grad`` views), like ``loss.backward()`` accumulates in the reference."""

    @staticmethod
    def forward(ctx, anchor, model, t, v, a, len_dev, training, seed):
        model._forward_raw(t, v, a, len_dev, training, seed)
        pub = model._public()
        outs = []
        for k in _PUB:
            outs.append(pub[k].clone() if k in pub else torch.zeros(0, device=t.device))
        labels = pub["labels"].clone()
        ctx.model = model
        ctx.fwd_id = model._fwd_id
        ctx.io = (t, v, a, len_dev)
        ctx.mark_non_differentiable(labels)
        return tuple(outs) + (labels,)

    @staticmethod
    def backward(ctx, *grads):
        model = ctx.model
        if ctx.fwd_id != model._fwd_id:
            raise RuntimeError("backward through a stale MISA forward: activations live in a per-model workspace that the "
                               "next forward overwrites (run forward -> backward in order, as the reference loop does)")
        lib = model._lib
        s = _lib.stream_ptr()
        _lib.check(lib.mmda_misa_zero_act_grads(model._h, s), "zero_act_grads")
        slots = model._grad_slots()
        for k, g in zip(_PUB, grads[:-1]):
            if g is not None and k in slots and g.numel() > 0:
                slots[k].copy_(g)
        t, v, a, len_dev = ctx.io
        _lib.check(lib.mmda_misa_backward(model._h, t.data_ptr(), v.data_ptr(), a.data_ptr(), len_dev.data_ptr(), s),
                   "mmda_misa_backward")
        model._assign_grad_views()
        return (torch.zeros_like(model._anchor),) + (None,) * 7


Model = MISA   # BASELINE.json north_star: "keeping the Model(config) constructor"
