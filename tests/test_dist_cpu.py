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


def _agree_worker(rank, world, port, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        dp = DataParallelSync()
        # each rank computed "its" dev loss; the decision must come out the same everywhere (rank 0's value)
        mine = 0.5 + 1e-7 * rank
        q.put((rank, dp.agree(mine), dp.any_rank(rank == 1), dp.any_rank(False)))
    finally:
        dist.destroy_process_group()


def test_collective_decisions_world2():
    """Solver.train()'s best-epoch decision and its cluster check are collective (mmda_amd/solver.py): rank 0's loss everywhere, an
    error flag raised by any rank seen by all."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_agree_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    assert res[0][1] == res[1][1] == 0.5
    assert res[0][2] is True and res[1][2] is True
    assert res[0][3] is False and res[1][3] is False


def test_bench_launches_its_own_ranks():
    """`python bench.py --gpus N` without torchrun (how the driver calls it): the script starts N fresh rank processes with the
    rendezvous environment set; rank 0's output is the launcher's stdout.  --launch-check stops every rank before its first GPU call."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "3", "--launch-check"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr
    out = [json.loads(x) for x in r.stdout.splitlines() if x.startswith("{")]
    err = [json.loads(x) for x in r.stderr.splitlines() if x.startswith("{")]
    assert len(out) == 1 and out[0]["rank"] == 0 and out[0]["world"] == 3 and out[0]["master"] == "127.0.0.1"
    assert sorted(e["rank"] for e in err) == [1, 2] and all(e["port"] == out[0]["port"] for e in err)
    # under a launcher (WORLD_SIZE set) it must NOT spawn again
    env2 = dict(env, RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1", MASTER_PORT="1")
    r2 = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--launch-check"], env=env2,
                        capture_output=True, text=True, timeout=300)
    assert r2.returncode == 0 and len([x for x in r2.stdout.splitlines() if x.startswith("{")]) == 1 and not r2.stderr.strip().startswith("{")


def _global_stats_worker(rank, world, port, q):
    """The protocol of DataParallelSync(global_stats=True) (mmda_amd/models.py: MISA._global_stats_step) with the oracle's loss
    functions standing in for the HIP loss kernels: gather the ranks' rows, evaluate the batch-statistic loss on the gathered batch,
    back-propagate THIS rank's rows of its gradient times the world size, average the parameter gradients."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from types import SimpleNamespace
        from oracle import misa_oracle as orc
        dp = DataParallelSync(global_stats=True)
        B, D = 5, 16
        W = torch.randn(6, D, D, generator=torch.Generator().manual_seed(7)).requires_grad_(True)      # the "model": six D x D maps
        u = torch.randn(B, D, generator=torch.Generator().manual_seed(50 + rank))                      # this rank's inputs

        def x6_of(inp, Wm):
            return torch.sigmoid(torch.einsum("bd,kde->kbe", inp, Wm))                                 # (6, B, D)

        def batch_losses(x6):
            o = SimpleNamespace(utt_private_t=x6[0], utt_private_v=x6[1], utt_private_a=x6[2], utt_shared_t=x6[3], utt_shared_v=x6[4],
                                utt_shared_a=x6[5])
            return 0.3 * orc.diff_loss(o) + 0.7 * orc.cmd_loss(o), x6.mean()                           # (batch-statistic, per-sample mean)

        x6 = x6_of(u, W)
        X = dp.gather_rows(x6.detach(), dim=1).requires_grad_(True)                                    # (6, world B, D)
        ok_rows = torch.equal(X[:, rank * B:(rank + 1) * B], x6.detach())
        Lg, _ = batch_losses(X)
        Lg.backward()
        mine = X.grad[:, rank * B:(rank + 1) * B] * world                                              # this rank's rows, times world
        (x6 * mine).sum().backward(retain_graph=True)                                                  # into dW through the local graph
        x6.mean().backward()                                                                           # the per-sample-mean part
        g = W.grad.reshape(-1).clone()
        scale = dp.sync(g, g.numel())
        got = g * scale
        # one "device": the same losses on the concatenated batch
        W1 = W.detach().clone().requires_grad_(True)
        U = torch.cat([torch.randn(B, D, generator=torch.Generator().manual_seed(50 + r)) for r in range(world)], 0)
        Lb, Lm = batch_losses(x6_of(U, W1))
        (Lb + Lm).backward()
        ref = W1.grad.reshape(-1)
        q.put((rank, ok_rows, float((got - ref).abs().max() / ref.abs().max()), float(Lg), float(Lb)))
    finally:
        dist.destroy_process_group()


def test_global_statistics_protocol_world2():
    """SURVEY.md 8e's optional mode, host side: with the gathered rows, `own rows x world` and the averaging exchange, two ranks give
    the gradient (and the loss) of ONE batch of both shards."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_global_stats_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, ok_rows, err, Lg, Lb in res:
        assert ok_rows
        assert err < 1e-5, err
        assert abs(Lg - Lb) < 1e-6 * abs(Lb) + 1e-7
