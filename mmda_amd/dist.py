"""Data-parallel gradient exchange: one process per GPU, torch.distributed ('nccl' backend = RCCL over xGMI on ROCm).

The reference is single-device (its only multi-GPU trace is a commented-out nn.DataParallel, solver.py:88-91).  Here each
rank runs the reference loop on its own minibatch shard and the ranks exchange ONE flat fp32 gradient bucket per step
(SURVEY.md 8e): the dense prefix (all non-embedding parameters, 19.25 MB at MOSEI sizes) plus the dense embedding
gradient, summed with a single all-reduce and averaged by passing grad_scale = 1/world to the fused clamp+Adam kernel, so
the clip happens AFTER averaging exactly like the single-process order (solver.py:183-186).

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
    def __init__(self, group=None, bucket_mb: float = 0.0, sparse_embedding: bool = False, overlap: bool = True):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.bucket_floats = int(bucket_mb * (1 << 20) / 4) if bucket_mb > 0 else 0
        # Opt-in: exchange the embedding gradient as (ids, rows) instead of all-reducing V x d_t.  Less traffic (1.9 MB per rank
        # against 24 MB at V = 20 000), but every rank then scatter-adds with float atomics in its own order, so replicas agree
        # only to rounding, not bit for bit as after an all-reduce.  Default: the dense, bit-identical exchange.
        self.sparse_embedding = bool(sparse_embedding)
        self.overlap = bool(overlap)          # reduce the early-finished prefix of the bucket beside the rest of the backward pass
        self._comm = None

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

    def _all_reduce(self, t: torch.Tensor, async_op: bool = False):
        """all-reduce(sum) in place.  With the gloo backend (CPU tests, or a 2-rank rehearsal on one GPU) device tensors are
        staged through host memory; with nccl (= RCCL) the tensor is reduced in place over xGMI."""
        if t.is_cuda and dist.get_backend(self.group) == "gloo":
            h = t.cpu()
            dist.all_reduce(h, op=dist.ReduceOp.SUM, group=self.group)
            t.copy_(h)
            return None
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    def _early_split(self, flat_grads: torch.Tensor, model) -> int:
        """Floats of the bucket that may be reduced ahead of the rest (0: none / not applicable)."""
        if not self.overlap or model is None or not flat_grads.is_cuda or not hasattr(model, "early_grad_floats"):
            return 0
        early = int(model.early_grad_floats())
        if early <= 0 or early >= flat_grads.numel():
            return 0
        if self._comm is None:
            self._comm = torch.cuda.Stream(device=flat_grads.device)
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

    def _sync_sparse_embedding(self, flat_grads: torch.Tensor, dense_floats: int, model) -> None:
        """Dense prefix: one all-reduce.  Embedding gradient: every rank's (ids, rows) are all-gathered and the other ranks'
        rows scatter-added into the local dense gradient (which already holds the local rows) -- 2 * T*B*d_t*4 bytes per
        rank on the wire instead of an all-reduce of V*d_t*4 (1.9 MB against 24 MB at MOSEI sizes, V = 20 000)."""
        self._all_reduce(flat_grads[:dense_floats])
        ids, rows = model.embedding_grad_rows()
        # ranks may hold batches of different length: agree on the row count, pad with (id 0, zero row)
        n = torch.tensor([ids.numel()], dtype=torch.int64, device=ids.device)
        counts = self._all_gather(n).view(-1).tolist()
        cap = max(counts)
        if ids.numel() < cap:
            pad = cap - ids.numel()
            ids = torch.cat([ids, ids.new_zeros(pad)])
            rows = torch.cat([rows, rows.new_zeros(pad, rows.shape[1])])
        all_ids = self._all_gather(ids)               # (world, cap)
        all_rows = self._all_gather(rows)             # (world, cap, d_t)
        if self.rank > 0:
            model.scatter_embedding_rows(all_ids[:self.rank].reshape(-1), all_rows[:self.rank].reshape(-1, rows.shape[1]))
        if self.rank + 1 < self.world:
            model.scatter_embedding_rows(all_ids[self.rank + 1:].reshape(-1), all_rows[self.rank + 1:].reshape(-1, rows.shape[1]))

    def sync(self, flat_grads: torch.Tensor, dense_floats: int, model=None) -> float:
        """All-reduce(sum) the gradient bucket in place; returns the scale (1/world) the optimizer applies.
        With ``sparse_embedding=True`` and a model that exposes ``embedding_grad_rows`` the embedding gradient travels in its
        sparse form (see __init__).
        With bucket_floats > 0 the bucket is cut into chunks so the first all-reduce can start while later chunks are
        still being enqueued (xGMI rings are per-link bound; >=8 MB chunks keep them at bandwidth)."""
        if self.world == 1:
            return 1.0
        if model is not None and self.sparse_embedding and hasattr(model, "embedding_grad_rows") \
                and dense_floats < flat_grads.numel():
            self._sync_sparse_embedding(flat_grads, dense_floats, model)
            return 1.0 / self.world
        n = flat_grads.numel()
        early = self._early_split(flat_grads, model)
        if early:
            # The bucket is laid out in the order the backward pass completes it; the native step recorded an event behind the
            # last kernel that writes its first `early` floats (fusion block, LayerNorms, layer-2 recurrent layers).  That part
            # is reduced on a stream of its own as soon as the event fires -- beside the layer-1 backward recurrence, which
            # leaves more than half of the CUs idle -- and the rest (layer 1, embedding) after the step's last kernel.
            # `early` is a function of the model configuration and the build, identical on every rank; a failure of the event wait
            # is raised, not handled: a rank that fell back to ONE all-reduce while its peers issue TWO would hang the job or,
            # worse, sum mismatched ranges.
            cur = torch.cuda.current_stream(flat_grads.device)
            model.wait_early_grads(self._comm)
            with torch.cuda.stream(self._comm):
                self._all_reduce(flat_grads[:early])
            self._all_reduce(flat_grads[early:])
            cur.wait_stream(self._comm)
            return 1.0 / self.world
        if self.bucket_floats <= 0 or self.bucket_floats >= n:
            self._all_reduce(flat_grads)
        else:
            works = []
            for s in range(0, n, self.bucket_floats):
                works.append(self._all_reduce(flat_grads[s:s + self.bucket_floats], async_op=True))
            for w in works:
                if w is not None:
                    w.wait()
        return 1.0 / self.world
