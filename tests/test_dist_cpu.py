"""CPU, world_size 2, gloo: the data-parallel gradient exchange (mmda_amd/dist.py) on a flat bucket.
DDP contract: all-reduce(sum) of the bucket, then grad_scale = 1/world inside the optimizer, so that
averaged-gradient == mean of the shard gradients, and every rank ends with identical parameters."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from mmda_amd.dist import DataParallelSync


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, bucket_mb, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dp = DataParallelSync(bucket_mb=bucket_mb)
        n_dense, n_emb = 100_003, 5_000
        g = torch.Generator().manual_seed(100 + rank)
        grads = torch.randn(n_dense + n_emb, generator=g)
        mine = grads.clone()
        scale = dp.sync(grads, n_dense)
        # oracle: in-process "fake collective" = sum of every rank's shard gradient
        expect = sum(torch.randn(n_dense + n_emb, generator=torch.Generator().manual_seed(100 + r)) for r in range(world))
        ok_sum = torch.allclose(grads, expect, atol=1e-6)
        ok_scale = abs(scale - 1.0 / world) < 1e-12
        # parameters broadcast from rank 0

        class M:
            def __init__(self, p): self.p = p
            def flat_buckets(self): return (self.p, None, None, None)
        p = torch.full((1000,), float(rank))
        dp.broadcast_parameters(M(p))
        ok_bcast = bool((p == 0).all())
        # clip AFTER averaging (solver.py:183-186 order): identical update on every rank
        upd = (grads * scale).clamp(-1, 1)
        gathered = [torch.zeros_like(upd) for _ in range(world)]
        dist.all_gather(gathered, upd)
        ok_same = all(torch.equal(gathered[0], x) for x in gathered)
        q.put((rank, ok_sum, ok_scale, ok_bcast, ok_same, bool((mine != grads).any())))
    finally:
        dist.destroy_process_group()


def _run(bucket_mb):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, bucket_mb, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in res:
        assert all(r[1:]), r


def test_flat_bucket_allreduce_world2_single_bucket():
    _run(0.0)


def test_flat_bucket_allreduce_world2_chunked():
    _run(0.1)


def _sparse_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dp = DataParallelSync(sparse_embedding=True)
        n_dense, V, D = 1003, 40, 6
        R = 7 + 3 * rank                              # ranks hold batches of different length: the exchange pads
        g = torch.Generator().manual_seed(200 + rank)
        ids = torch.randint(0, V, (R,), generator=g)
        rows = torch.randn(R, D, generator=g)
        dense = torch.randn(n_dense, generator=g)

        class FakeModel:                              # the two hooks of mmda_amd.models.MISA that the sparse exchange uses
            def __init__(self):
                self.G = torch.cat([dense, torch.zeros(V * D).index_add_(0, (ids[:, None] * D + torch.arange(D)).reshape(-1), rows.reshape(-1))])
            def embedding_grad_rows(self): return ids, rows
            def set_embedding_grad_rows(self, i, r):          # rows of the ids that occur are overwritten by the list-order sums; ids < 0 skipped
                E = self.G[n_dense:].view(V, D)
                for v in sorted(set(int(x) for x in i.tolist() if x >= 0)):
                    acc = torch.zeros(D)
                    for p in range(i.numel()):
                        if int(i[p]) == v:
                            acc = acc + r[p]
                    E[v] = acc
        m = FakeModel()
        scale = dp.sync(m.G, n_dense, m)
        # oracle: sum over ranks of the DENSE gradients
        exp = torch.zeros(n_dense + V * D)
        for r in range(world):
            gg = torch.Generator().manual_seed(200 + r)
            Rr = 7 + 3 * r
            i2 = torch.randint(0, V, (Rr,), generator=gg); r2 = torch.randn(Rr, D, generator=gg); d2 = torch.randn(n_dense, generator=gg)
            exp[:n_dense] += d2
            exp[n_dense:].index_add_(0, (i2[:, None] * D + torch.arange(D)).reshape(-1), r2.reshape(-1))
        q.put((rank, torch.allclose(m.G, exp, atol=1e-5), abs(scale - 1.0 / world) < 1e-12))
    finally:
        dist.destroy_process_group()


def test_sparse_embedding_exchange_world2_ragged_shards():
    """sparse form (the default on GPUs): dense prefix all-reduced, embedding gradient as all-gathered (ids, rows) with padding to the
    longest shard (id -1), summed in list order on every rank -- the result must equal the all-reduce of the dense gradients."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sparse_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for r in res:
        assert all(r[1:]), r
