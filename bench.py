#!/usr/bin/env python3
"""Throughput of the MISA training hot path on MI355X:  python bench.py --gpus N --steps K --warmup W

One "step" = one full reference loop iteration (solver.py:139-186: zero_grad, forward, six losses, backward,
clip_grad_value_, Adam; dropout ON) on one synthetic MOSEI-shaped batch per GPU that is already resident in HBM.
N=1 workload = BASELINE.json configs[1]: B=32, T=50, (d_t,d_v,d_a)=(300,35,74), hidden 128, V=20000, bf16 MFMA operands
with fp32 accumulate.  N>1: one process per GPU -- started by torchrun (RANK / WORLD_SIZE in the environment) or, when called
bare as `python bench.py --gpus N`, by this script itself (launch_ranks: N fresh children, the parent never touches a GPU) --, the
same per-GPU batch shape on every rank (weak scaling; default B=32/GPU so the curve continues the N=1 point, `--batch 256` is
BASELINE configs[2]), gradient exchange over RCCL per step.  Rank 0 prints ONE JSON line.

Extra legs on the same line (rank 0, N=1 only for cpu_baseline):
  roofline     - the dominant kernel (the biLSTM recurrence), timed with HIP events on its own stream inside the timed region
  cpu_baseline - the oracle's stock-PyTorch CPU training loop on the same shapes, bounded sample
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402


def launch_ranks(n: int) -> int:
    """`python bench.py --gpus N` without torchrun: start one child process per GPU (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in
    its environment, rendezvous on 127.0.0.1) and relay rank 0's JSON line.  Returns the largest exit code of the ranks.  A rank
    that dies takes the others down (they would wait in a collective for ever)."""
    import socket
    import subprocess
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=os.environ.get("MASTER_PORT", str(port)))
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        # rank 0 prints the line on this process's stdout; whatever the other ranks print goes to stderr
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else sys.stderr))
    rc = 0
    alive = list(procs)
    while alive:
        for p in list(alive):
            code = p.poll()
            if code is None:
                continue
            alive.remove(p)
            rc = max(rc, abs(code))
            if code != 0:
                for q in alive:
                    q.terminate()
        time.sleep(0.05)
    return rc


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--batch", type=int, default=32, help="per-GPU batch (BASELINE configs[1]: 32; configs[2]: 256)")
    ap.add_argument("--seq-len", type=int, default=50)
    ap.add_argument("--vocab", type=int, default=20000)
    ap.add_argument("--precision", default="bf16", choices=["bf16", "fp32"])
    ap.add_argument("--ragged", type=int, default=0)
    ap.add_argument("--confidnet", type=int, default=0)
    ap.add_argument("--rnncell", default="lstm", choices=["lstm", "gru"], help="config.rnncell (the headline config is lstm)")
    ap.add_argument("--fp8-fusion", type=int, default=0, help="BASELINE configs[4]: the fusion layer's feed-forward products on block-scaled fp8 (forward)")
    ap.add_argument("--global-stats", type=int, default=0, help="N > 1: DiffLoss / CMD / conf on the batch of all ranks (config.dp_global_stats; the headline runs keep DDP semantics)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--streaming-recurrence", action="store_true", help="bf16: stream W_hh from L2 per step instead of LDS-resident")
    ap.add_argument("--cpu-steps", type=int, default=0, help="0 = pick a count that takes ~10-30 s")
    ap.add_argument("--launch-check", action="store_true", help="every rank prints its rendezvous environment and exits (no GPU call): tests the launcher")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and world > 1:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.gpus > 1 and world == 1 and "WORLD_SIZE" in os.environ:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE=1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # called as `python bench.py --gpus N` without a launcher: this process becomes the launcher.  It has not touched the GPU
        # (importing torch does not) and never will; the ranks are fresh child processes.
        raise SystemExit(launch_ranks(args.gpus))
    if args.launch_check:
        print(json.dumps({"rank": rank, "local_rank": local_rank, "world": world, "master": os.environ.get("MASTER_ADDR"),
                          "port": os.environ.get("MASTER_PORT")}), file=sys.stdout if rank == 0 else sys.stderr)
        return
    # rehearsal switches (not for measurements): MMDA_BENCH_BACKEND=gloo + MMDA_BENCH_ONE_DEVICE=1 run the N > 1 code path with
    # every rank on cuda:0 (RCCL refuses two ranks on one device), gradients staged through the host
    backend = os.environ.get("MMDA_BENCH_BACKEND", "nccl")
    if os.environ.get("MMDA_BENCH_ONE_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev)
        else:
            dist.init_process_group(backend)

    from mmda_amd import make_config
    from mmda_amd.solver import Solver
    from mmda_amd.data import synth_batch
    from mmda_amd import _lib

    torch.manual_seed(0)
    cfg = make_config(vocab_size=args.vocab, precision=args.precision, device=str(dev), batch_size=args.batch,
                      seq_len=args.seq_len, use_confidNet=bool(args.confidnet), rnncell=args.rnncell, pretrained_emb=torch.randn(args.vocab, 300),
                      fusion_fp8=bool(args.fp8_fusion), dp_global_stats=bool(args.global_stats))
    solver = Solver(cfg, cfg, cfg, None, None, None, is_train=True).build()
    model = solver.model
    model.train()
    if args.streaming_recurrence:
        model.set_recurrence(False)
    t, v, a, y, emo, lengths, *_ = synth_batch(cfg, args.batch, args.seq_len, seed=rank, ragged=bool(args.ragged), device=dev)
    if solver.dp is not None:
        solver.dp.equal_shapes = True            # every rank steps the same (B, T): no row-count exchange in front of the all-gathers
    sync = solver.dp.sync if solver.dp is not None else None

    def step():
        model.train_step(t, v, a, lengths, emo, lr=cfg.learning_rate, clip=cfg.clip, grad_sync=sync)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            torch.distributed.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    lib = model._lib
    if rank == 0 and not os.environ.get("MMDA_BENCH_NO_KERNEL_TIMING"):
        # HIP events around the four recurrent launches of every 8th timed step (25 samples at the default 200 steps): the eight
        # event records cost ~35 us per step, which the metric should not carry on every step.  A short run (the driver's 20 steps)
        # samples four of its steps, never every one.
        # A short run samples FOUR of its steps (four samples per launch).  (mmda_misa_timing_rotate spreads the same number of event
        # pairs over more steps -- one launch per sampled step -- at the same total cost; off here.)
        stride = 8 if args.steps >= 64 else max(1, -(-args.steps // 4))
        _lib.check(lib.mmda_misa_timing_stride(model._h, stride), "timing_stride")
        _lib.check(lib.mmda_misa_timing_begin(model._h, (args.steps + stride - 1) // stride), "timing_begin")
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tt = torch.tensor([elapsed], device=dev if backend == "nccl" else "cpu", dtype=torch.float64)
        torch.distributed.all_reduce(tt, op=torch.distributed.ReduceOp.MAX)
        elapsed = float(tt.item())
    losses = model.read_losses()
    if not all(x == x for x in losses.values()):
        raise SystemExit(f"non-finite losses: {losses}")
    if model.cluster_aborted():
        raise SystemExit("a resident-weights recurrence timed out waiting for its workgroup cluster: results invalid")

    if rank != 0:
        if world > 1:
            torch.distributed.destroy_process_group()
        return

    if os.environ.get("MMDA_BENCH_NO_KERNEL_TIMING"):        # diagnostic: the step rate without the per-kernel HIP events
        print(json.dumps({"ms_per_step_without_kernel_events": round(1e3 * elapsed / args.steps, 4)}))
        return
    # ---- roofline of the dominant kernel (biLSTM recurrence; 4 launches per step with equal algorithmic FLOPs)
    ms4 = (C.c_float * 4)()
    nst = C.c_int()
    _lib.check(lib.mmda_misa_timing_collect(model._h, C.byref(ms4), C.byref(nst)), "timing_collect")
    lib.mmda_misa_timing_end(model._h)
    ms = [float(x) for x in ms4]
    sumH2 = 300 ** 2 + 35 ** 2 + 74 ** 2
    real_steps = float(lengths.sum().item())            # sum_b len_b (= T*B when not ragged)
    flops_per_launch = 16.0 * real_steps * sumH2        # 2 dirs * 2 FLOP/MAC * 4H*H per (sample, step), summed over modalities
    if args.rnncell == "gru":
        flops_per_launch *= 0.75                        # three gate blocks (the padded fourth slot is not algorithmic work)
    names = ["lstm_fwd_kernel(layer1)", "lstm_fwd_kernel(layer2)", "lstm_bwd_kernel(layer2)", "lstm_bwd_kernel(layer1)"]
    k = max(range(4), key=lambda i: ms[i])
    peak = 2500.0 if args.precision == "bf16" else 157.3
    achieved = flops_per_launch / (ms[k] * 1e-3) / 1e12 if ms[k] > 0 else 0.0
    # HBM bytes per launch of that kernel from the PMC counters (FETCH_SIZE + WRITE_SIZE, separate rocprofv3 passes, see
    # profiles/): only valid for the configuration the profile was taken on, otherwise null
    # (PMC counters cannot be read from inside the process: they come from the committed rocprofv3 passes of THIS command,
    # tools/prof_pmc.sh -> profiles/pmc_traffic.json; FETCH_SIZE doubled as the MI355X guide prescribes for gfx950)
    traffic = mfma_busy = None
    tfile = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if os.path.exists(tfile):
        try:
            tj = json.load(open(tfile))
            if (tj.get("batch"), tj.get("seq_len"), tj.get("precision")) == (args.batch, args.seq_len, args.precision) and not args.ragged:
                kn = names[k].split("(")[0]
                traffic = tj["bytes_per_launch"].get(kn)
                mfma_busy = {"chip_percent": tj.get("mfma_util_chip_percent", {}).get(kn),
                             "share_of_wave_cycles_percent": tj.get("mfma_busy_share_of_wave_cycles_percent", {}).get(kn),
                             "source": tj.get("source")}
        except Exception:
            traffic = mfma_busy = None
    roofline = {"bound": "mfma", "kernel": names[k], "achieved": round(achieved, 3), "peak": peak, "unit": "TFLOP/s",
                "frac": round(achieved / peak, 6), "traffic": traffic, "launch_ms": round(ms[k], 4),
                "all_launch_ms": {n: round(m, 4) for n, m in zip(names, ms)},
                "recurrent_share_of_step": round(sum(ms) / (elapsed / args.steps * 1e3), 3), "timed_steps": nst.value,
                "mfma_busy": mfma_busy}
    # What actually bounds that kernel: neither MFMA nor HBM but the chain of T dependent cross-CU hand-offs (h_t needs every hidden
    # tile of h_{t-1}).  Its floor is the guide's handoff-1to1 price (0.8 - 1.0 us per hop on an idle chip, MI355X_MICROARCH.md
    # price list); frac = T x floor / launch time.
    hop_floor_us = 0.9
    chain_steps = int(lengths.max().item())
    roofline_chain = {"bound": "serial_chain", "kernel": names[k], "steps": chain_steps, "floor_us_per_step": hop_floor_us,
                      "achieved_us_per_step": round(ms[k] * 1e3 / max(chain_steps, 1), 3),
                      "frac": round(chain_steps * hop_floor_us / (ms[k] * 1e3), 4) if ms[k] > 0 else 0.0,
                      "all_us_per_step": {n: round(m * 1e3 / max(chain_steps, 1), 3) for n, m in zip(names, ms)}}

    total_samples = args.batch * world * args.steps
    value = total_samples / elapsed
    out = {
        "metric": "training samples/sec on MOSEI-shaped synthetic batch", "value": round(value, 2), "unit": "samples/s",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(elapsed / args.steps * 1e3, 4),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.precision, "data": "synthetic",
        "config": {"workload": f"MISA train step, MOSEI shapes B={args.batch}/GPU T={args.seq_len} (d_t,d_v,d_a)=(300,35,74) "
                               f"hidden=128 V={args.vocab}, {args.precision} MFMA operands fp32 accumulate"
                               f"{', fusion FFN products on block-scaled fp8' if args.fp8_fusion else ''}, dropout on, "
                               f"{'ragged' if args.ragged else 'full'} lengths",
                   "global_batch": args.batch * world, "seq_len": args.seq_len,
                   "parallelism": f"dp{world}" if world > 1 else "single", "use_confidNet": bool(args.confidnet),
                   "rnncell": args.rnncell, "dp_global_stats": bool(args.global_stats) and world > 1},
        "gflop_per_sample": 1.184 if args.seq_len == 50 else round(3 * (7736080 * args.seq_len + 7832576) / 1e9, 3),
        "roofline": roofline,
        "roofline_serial_chain": roofline_chain,
        "losses": {k_: round(v_, 5) for k_, v_ in losses.items()},
    }

    # ---- CPU baseline: the oracle's stock-PyTorch port of the reference loop on this box's host cores (N=1 only)
    if world == 1 and not args.no_cpu_baseline:
        from oracle import misa_oracle as orc
        # host cores this process may actually use (cgroup/affinity share, 16 on a 1-GPU box), not the machine's count:
        # oversubscribing torch's intra-op pool makes the CPU loop orders of magnitude slower
        try:
            ncores = len(os.sched_getaffinity(0))
        except AttributeError:
            ncores = os.cpu_count() or 1
        ncores = max(1, min(ncores, int(os.environ.get("MMDA_CPU_THREADS", "16"))))
        torch.set_num_threads(ncores)
        ocfg = orc.default_config(vocab_size=args.vocab, use_confidNet=bool(args.confidnet), rnncell=args.rnncell)
        cb = {"t": t.cpu(), "v": v.cpu(), "a": a.cpu(), "l": lengths.cpu(), "emo": emo.cpu()}
        nsteps = args.cpu_steps
        if nsteps <= 0:
            probe = orc.baseline_train_steps(ocfg, cb, steps=1, warmup=1)
            nsteps = int(max(2, min(60, 15.0 / max(probe, 1e-3))))
        secs = orc.baseline_train_steps(ocfg, cb, steps=nsteps, warmup=1)
        out["cpu_baseline"] = {"value": round(args.batch * nsteps / secs, 2), "unit": "samples/s", "cores": torch.get_num_threads(),
                               "kind": "port",
                               "sample": f"{nsteps} training steps of the same B={args.batch},T={args.seq_len} batch "
                                         f"(oracle.ModuleBaseline: nn.LSTM/nn.TransformerEncoderLayer/torch.optim.Adam fp32, "
                                         f"pinned to the reference-generated fixtures by tests/test_oracle_golden.py)"}
        out["speedup_vs_cpu"] = round(value / out["cpu_baseline"]["value"], 1)
    print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
