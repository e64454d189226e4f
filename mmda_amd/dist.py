"""Data-parallel gradient exchange: one process per GPU, torch.distributed ('nccl' backend = RCCL over xGMI on ROCm).

The reference is single-device (its only multi-GPU trace is a commented-out nn.DataParallel, solver.py:88-91).  Here each
rank runs the reference loop on its own minibatch shard and the ranks exchange ONE flat fp32 gradient bucket per step
(SURVEY.md 8e): the dense prefix (all non-embedding parameters, 19.25 MB at MOSEI sizes) plus the dense embedding
gradient, summed with a single all-reduce and averaged by passing grad_scale = 1/world to the fused clamp+Adam kernel, so
the clip happens AFTER averaging exactly like the single-process order (solver.py:183-186).

Semantics = DDP: every rank is the reference at batch_size = B_local; the batch-statistic losses (DiffLoss, CMD, conf)
are per-shard, gradients are averaged.  Works on CPU tensors with the gloo backend (tests) and on GPU with RCCL.
"""
from __future__ import annotations

import torch
import torch.distributed as dist


class DataParallelSync:
    def __init__(self, group=None, bucket_mb: float = 0.0):
        if not dist.is_initialized():
            raise RuntimeError("torch.distributed is not initialised")
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        self.bucket_floats = int(bucket_mb * (1 << 20) / 4) if bucket_mb > 0 else 0

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

    def sync(self, flat_grads: torch.Tensor, dense_floats: int) -> float:
        """All-reduce(sum) the gradient bucket in place; returns the scale (1/world) the optimizer applies.
        With bucket_floats > 0 the bucket is cut into chunks so the first all-reduce can start while later chunks are
        still being enqueued (xGMI rings are per-link bound; >=8 MB chunks keep them at bandwidth)."""
        if self.world == 1:
            return 1.0
        n = flat_grads.numel()
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
