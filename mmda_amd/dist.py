"""Data-parallel gradient exchange: one process per GPU, torch.distributed ('nccl' backend = RCCL over xGMI on ROCm).

The reference is single-device (its only multi-GPU trace is a commented-out nn.DataParallel, solver.py:88-91).  Here each
rank runs the reference loop on its own minibatch shard and the ranks exchange ONE flat fp32 gradient bucket per step
(SURVEY.md 8e): the dense prefix (all non-embedding parameters, 19.25 MB at MOSEI sizes) is all-reduced; the embedding gradient
travels as all-gathered (ids, rows) and is summed by every rank with the native deterministic segment sum (or, on request, dense
inside the same all-reduce); the average is taken by passing grad_scale = 1/world to the fused clamp+Adam kernel, so the clip
happens AFTER averaging exactly like the single-process order (solver.py:183-186).

The bucket is ordered by completion time of the backward pass; with ``overlap=True`` (default) the prefix that is final beside
the layer-1 backward recurrence (fusion block, LayerNorms, layer-2 recurrent layers) is reduced on its own stream behind an
event the native step records, the rest after the step's last kernel (``sync``).

Semantics = DDP: every rank is the reference at batch_size = B_local; the batch-statistic losses (DiffLoss, CMD, conf)
are per-shard, gradients are averaged.  Works on CPU tensors with the gloo backend (tests) and on GPU with RCCL.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class DataParallelSync:
    def __init__(self, group=None, bucket_mb: float = 0.0, sparse_embedding=None, overlap: bool = True, equal_shapes: bool = False,
                 force_collectives: bool = False, global_stats: bool = False):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.bucket_floats = int(bucket_mb * (1 << 20) / 4) if bucket_mb > 0 else 0
        # The embedding gradient travels as all-gathered (ids, rows) -- 2 * T*B*d_t*4 bytes per rank instead of an all-reduce of
        # V*d_t*4 (1.9 MB against 24 MB at MOSEI sizes, V = 20 000) -- and every rank then sums the SAME gathered list in the SAME
        # order with the native deterministic segment sum (mmda_embed_segment_sum), so replicas stay bit-identical exactly as after an
        # all-reduce.  None (default): on whenever the model exposes the sparse view and its gradients live on a GPU; False: the
        # dense all-reduce of the whole bucket.
        self.sparse_embedding = sparse_embedding
        self.overlap = bool(overlap)          # reduce the early-finished prefix of the bucket beside the rest of the backward pass
        # every rank holds a batch of the same (B, T) on every step (benchmarks): the row counts need not be exchanged first
        self.equal_shapes = bool(equal_shapes)
        # test hook: a ONE-rank group still walks the whole exchange (communication stream, early event wait, early clip + Adam,
        # all-reduces, both all-gathers, segment sum) instead of returning at once -- the way to run the RCCL branch on a box with one GPU
        self.force_collectives = bool(force_collectives)
        # "global statistics" (optional mode): DiffLoss (batch mean + Gram contracted over the batch, reference
        # utils/functions.py:64-76), CMD (batch moments, :89-108) and the confidence loss (per-class nnz and batch softmax,
        # solver.py:451-462) are functions of the WHOLE batch, so the mean of the ranks' shard gradients is not the gradient of the
        # global-batch loss.  Default (False): DDP semantics -- every rank is the reference at batch_size = B_local.  True: the
        # model's training step gathers the 128-wide utterance vectors (and scores / tcp / labels) of all ranks, evaluates those three
        # losses on the gathered batch with the same loss kernels, and back-propagates its own rows of their gradients (times the world
        # size: the exchange averages), so world x B_local behaves like ONE batch (mmda_amd/models.py: MISA._global_stats_step).  A
        # fidelity mode: the step is no longer one native call, and the gathered losses run on every rank.
        self.global_stats = bool(global_stats)
        self._comm = None
        # Optional optimizer hook for the early prefix: `early_step(n_floats, stream)` is called on the communication stream right behind
        # the early all-reduce (the prefix's gradients are then final and summed), and `early_stepped` tells the caller how many leading
        # floats of the bucket have been stepped already.  Set per step by the model's fused path; None = no early step.
        self.early_step = None
        self.early_stepped = 0

    def broadcast_parameters(self, model):
        """Rank 0's weights everywhere (one broadcast of the flat bucket when the model has one)."""
        gloo_cuda = dist.get_backend(self.group) == "gloo"
        ts = [model.flat_buckets()[0]] if (hasattr(model, "flat_buckets") and model.flat_buckets()[0] is not None) \
            else [p.data for p in model.parameters()]
        for t in ts:
            if t.is_cuda and gloo_cuda:
                h = t.cpu()
                dist.broadcast(h, src=0, group=self.group)
                t.copy_(h)
            else:
                dist.broadcast(t, src=0, group=self.group)

    def agree(self, value: float, src: int = 0) -> float:
        """Rank `src`'s value on every rank (a decision every rank must take alike -- e.g. "is this epoch the best so far" -- may not
        depend on a float each rank computed for itself)."""
        if self.world == 1:
            return float(value)
        nccl = dist.get_backend(self.group) == "nccl"
        t = torch.tensor([float(value)], dtype=torch.float64, device=torch.device("cuda", torch.cuda.current_device()) if nccl else "cpu")
        dist.broadcast(t, src=src, group=self.group)
        return float(t.item())

    def any_rank(self, flag: bool) -> bool:
        """True on every rank if `flag` is true on any rank (one small MAX all-reduce)."""
        if self.world == 1:
            return bool(flag)
        nccl = dist.get_backend(self.group) == "nccl"
        t = torch.tensor([1.0 if flag else 0.0], dtype=torch.float32, device=torch.device("cuda", torch.cuda.current_device()) if nccl else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=self.group)
        return bool(t.item() > 0)

    def _all_reduce(self, t: torch.Tensor, async_op: bool = False):
        """all-reduce(sum) in place.  With the gloo backend (CPU tests, or a 2-rank rehearsal on one GPU) device tensors are
        staged through host memory; with nccl (= RCCL) the tensor is reduced in place over xGMI."""
        if t.is_cuda and dist.get_backend(self.group) == "gloo":
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)
            return None
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    def _all_reduce_chunked(self, t: torch.Tensor):
        """With bucket_floats > 0 the range is cut into chunks so the first all-reduce can start while later chunks are still
        being enqueued (xGMI rings are per-link bound; >= 8 MB chunks keep them at bandwidth)."""
        n = t.numel()
        if n == 0:
            return
        if self.bucket_floats <= 0 or self.bucket_floats >= n:
            self._all_reduce(t)
            return
        works = [self._all_reduce(t[s:s + self.bucket_floats], async_op=True) for s in range(0, n, self.bucket_floats)]
        for w in works:
            if w is not None:
                w.wait()

    def _early_split(self, flat_grads: torch.Tensor, model) -> int:
        """Floats of the bucket that may be reduced ahead of the rest (0: none / not applicable)."""
        if not self.overlap or model is None or not flat_grads.is_cuda or not hasattr(model, "early_grad_floats"):
            return 0
        early = int(model.early_grad_floats())
        if early <= 0 or early >= flat_grads.numel():
            return 0
        return early

    def _all_gather(self, t: torch.Tensor) -> torch.Tensor:
        """(world, *t.shape) gathered from every rank (equal shapes; t has at least one dimension)."""
        flat_shape = (self.world * t.shape[0],) + tuple(t.shape[1:])      # concatenation along dim 0: the form every backend takes
        if t.is_cuda and dist.get_backend(self.group) == "gloo":
            h = torch.empty(flat_shape, dtype=t.dtype)
            dist.all_gather_into_tensor(h, t.cpu().contiguous(), group=self.group)
            out = h.to(t.device)
        else:
            out = torch.empty(flat_shape, dtype=t.dtype, device=t.device)
            dist.all_gather_into_tensor(out, t.contiguous(), group=self.group)
        return out.view((self.world,) + tuple(t.shape))

    def gather_rows(self, t: torch.Tensor, dim: int = 0) -> torch.Tensor:
        """The ranks' tensors (equal shapes) concatenated along `dim` in rank order -- rank r's rows at [r * n, (r + 1) * n) --, on
        every rank, contiguous."""
        g = self._all_gather(t.contiguous())                      # (world, *t.shape)
        if dim == 0:
            return g.reshape((self.world * t.shape[0],) + tuple(t.shape[1:]))
        return torch.cat(list(g.unbind(0)), dim=dim).contiguous()

    def _use_sparse(self, flat_grads, dense_floats, model) -> bool:
        if self.sparse_embedding is False or model is None or dense_floats >= flat_grads.numel():
            return False
        hooks = hasattr(model, "embedding_grad_rows") and hasattr(model, "set_embedding_grad_rows")
        if not hooks:
            return False
        if self.sparse_embedding is True:
            return True
        if not flat_grads.is_cuda:
            return False
        # None = decide by volume: the (ids, rows) form moves world * T*B rows to every rank, the dense form V rows through the all-reduce
        # (MOSEI, V = 20 000, 8 ranks: B = 32 -> 12 800 gathered rows: sparse; B = 256 -> 102 400: dense).  Every rank decides alike: the
        # row count is the padded (T, B) of the batch, agreed on beforehand when shapes may differ (equal_shapes or the count exchange).
        rows_per_rank = getattr(model, "embedding_rows_per_step", lambda: None)()
        vocab_rows = getattr(model, "embedding_table_rows", lambda: None)()
        if rows_per_rank is None or vocab_rows is None or not self.equal_shapes:
            return True
        return self.world * int(rows_per_rank) < int(vocab_rows)

    def _exchange_embedding_rows(self, model) -> None:
        """Every rank's (ids, rows) all-gathered (rank-major list), then the deterministic segment sum of that list into the dense
        embedding gradient -- the same list, the same order, the same sums on every rank."""
        ids, rows = model.embedding_grad_rows()
        if not self.equal_shapes:
            # ranks may hold batches of different length (the reference's collate pads to the batch maximum): agree on the row
            # count, pad with (id -1, zero row) -- skipped by the segment sum.  One 8-byte collective and one host read per step.
            n = torch.tensor([ids.numel()], dtype=torch.int64, device=ids.device)
            cap = max(self._all_gather(n).view(-1).tolist())
            if ids.numel() < cap:
                pad = cap - ids.numel()
                ids = torch.cat([ids, ids.new_full((pad,), -1)])
                rows = torch.cat([rows, rows.new_zeros(pad, rows.shape[1])])
        # ONE all-gather: the ids ride along as an extra fp32 column (exact below 2^24 rows; -1 = padding) -- a second collective
        # costs its launch latency (~15 us over RCCL), which at these sizes is as much as the transfer itself
        d = rows.shape[1]
        if ids.numel() > 0 and int(getattr(model, "embedding_table_rows", lambda: 0)() or 0) < (1 << 24):
            packed = torch.cat([rows, ids.to(rows.dtype).unsqueeze(1)], dim=1)            # (cap, d_t + 1)
            allp = self._all_gather(packed).reshape(-1, d + 1)
            model.set_embedding_grad_rows(allp[:, d].to(torch.int64).contiguous(), allp[:, :d].contiguous())
            return
        all_ids = self._all_gather(ids)               # (world, cap)
        all_rows = self._all_gather(rows)             # (world, cap, d_t)
        model.set_embedding_grad_rows(all_ids.reshape(-1), all_rows.reshape(-1, d))

    def sync(self, flat_grads: torch.Tensor, dense_floats: int, model=None) -> float:
        """Exchange the gradient bucket in place (sum over ranks); returns the scale (1/world) the optimizer applies.

        The bucket is laid out in the order the backward pass completes it.  On a GPU the whole exchange runs on a communication
        stream of its own: the prefix that is final beside the layer-1 backward recurrence (fusion block, LayerNorms, layer-2
        recurrent layers; the native step records an event behind it) is all-reduced as soon as that event fires -- the recurrence
        leaves more than half of the CUs idle -- and the remainder (layer 1; the embedding gradient in its dense or its (ids, rows)
        form) behind the step's last kernel.  The calling stream waits for the communication stream before the optimizer runs."""
        self.early_stepped = 0
        if self.world == 1 and not self.force_collectives:
            return 1.0
        n = flat_grads.numel()
        sparse = self._use_sparse(flat_grads, dense_floats, model)
        end = dense_floats if sparse else n           # floats that travel by all-reduce
        if not flat_grads.is_cuda:
            self._all_reduce_chunked(flat_grads[:end])
            if sparse:
                self._exchange_embedding_rows(model)
            return 1.0 / self.world
        early = min(self._early_split(flat_grads, model), end)
        cur = torch.cuda.current_stream(flat_grads.device)
        if self._comm is None:
            self._comm = torch.cuda.Stream(device=flat_grads.device)
        if early:
            # `early` is a function of the model configuration and the build, identical on every rank; a failure of the event wait
            # is raised, not handled: a rank that fell back to ONE all-reduce while its peers issue TWO would hang the job or,
            # worse, sum mismatched ranges.
            model.wait_early_grads(self._comm)
            with torch.cuda.stream(self._comm):
                self._all_reduce(flat_grads[:early])
                if self.early_step is not None:
                    # clip + Adam of the prefix while the rest of the backward pass is still running on the main stream (nothing issued
                    # after the early event reads these fp32 parameters; the single-GPU fused step does the same on its side stream)
                    self.early_step(early, self._comm)
                    self.early_stepped = early
        self._comm.wait_stream(cur)                   # the rest is final once everything issued so far is through
        with torch.cuda.stream(self._comm):
            self._all_reduce_chunked(flat_grads[early:end])
            if sparse:
                self._exchange_embedding_rows(model)
        cur.wait_stream(self._comm)
        return 1.0 / self.world
