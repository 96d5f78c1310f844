#!/usr/bin/env python3
"""ONE R-MAT graph (config C5) split by rows over the ranks: the strong-scaling counterpart of bench.py's
one-graph-per-rank weak scaling (third row; graph_odenet_amd/partition.py).

  python tools/partition_bench.py                         one rank: the per-stage solver path on the whole graph
  python tools/partition_bench.py --gpus N                starts N ranks itself (graph_odenet_amd/launch.py), or
  python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/partition_bench.py --gpus N
      N ranks, one GPU each: every f-eval all-gathers the N x d operand over RCCL; the parameter gradients are
      summed with one flat all-reduce per step.  With fewer GPUs than ranks the ranks share the GPUs and exchange
      over gloo (rehearsal).

bench.py --gpus N calls run() for its `secondary.strong_scaling` entry.

Same model, optimiser and step as bench.py (ODEGCN3, 16 rk4 steps = 64 f-evals, forward + adjoint + Adam).  Prints one
JSON line on rank 0: steps/s of the whole job, the time of the exchange alone and the bytes it moves.
Development / secondary measurement; the contract line is bench.py's.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def run(dev, rank, world, backend, scale=20, edges=10_000_000, nfeat=128, hidden=128, nclass=16, ode_steps=16,
        steps=3, warmup=1, cyclic=False):
    """The strong-scaling measurement; every rank calls it (collectives inside), every rank gets the same dict back."""
    import torch.distributed as dist
    from graph_odenet_amd import models
    from graph_odenet_amd.optim import Adam
    from graph_odenet_amd.parallel import GradBucket, broadcast_parameters
    from graph_odenet_amd.partition import PartitionedGraph, RowPartition, global_sum
    from graph_odenet_amd.synth import rmat_coo

    # every rank draws the same graph / features (seed 0) and keeps its rows
    r, c, v, n = rmat_coo(scale, edges, seed=0, device=dev)
    part = RowPartition(n, world, rank) if cyclic else RowPartition.balanced(n, r, c, world, rank)
    pg = PartitionedGraph.from_coo(r, c, v, n, part, device=dev)
    pg.transpose()
    nnz_total = int(r.numel())
    del r, c, v
    gen = torch.Generator(device=dev).manual_seed(1000)
    x = part.take(torch.randn(n, nfeat, generator=gen, device=dev))
    labels = part.take(torch.randint(0, nclass, (n,), generator=gen, device=dev))
    train = torch.randperm(n, generator=gen, device=dev)[: n // 10]
    pos = part.local_positions(train)
    n_train = train.numel()

    torch.manual_seed(42)
    model = models.ODEGCN3(nfeat=nfeat, nhid=hidden, nclass=nclass, dropout=0.5,
                           method="rk4", step_size=1.0 / ode_steps).to(dev)
    broadcast_parameters(model, 0)
    opt = Adam(model.parameters(), lr=0.01, weight_decay=5e-4)
    bucket = GradBucket(model)

    def step():
        model.train()
        opt.zero_grad(set_to_none=False)
        out = model(x, pg)
        loss = F.nll_loss(out[pos], labels[pos], reduction="sum") / n_train
        loss.backward()
        bucket.allreduce_sum()
        opt.step()
        return loss

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(warmup):
        step()
    model.nfe = 0
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        loss = step()
    barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    el = el.item()
    nfe = model.nfe / max(steps, 1)
    loss = float(global_sum(loss.detach().clone()))

    # the exchange alone
    probe = torch.randn(part.n_per, hidden, device=dev)
    for _ in range(3):
        pg.gather(probe)
    barrier()
    t0 = time.perf_counter()
    for _ in range(20):
        pg.gather(probe)
    barrier()
    t_g = torch.tensor([(time.perf_counter() - t0) / 20], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(t_g, op=dist.ReduceOp.MAX)
    t_g = t_g.item()
    recv = (world - 1) * part.n_per * hidden * 4
    gathers = 3 * 4 * ode_steps + 4       # fwd eval + (recompute + transpose) per adjoint stage, + 2 layers fwd/bwd
    return {
        "metric": "ONE row-partitioned graph: ODE-GCN forward+backward steps/sec at 64 RK4 evals",
        "value": round(steps / el, 4), "unit": "steps/s", "n_gpus": world, "steps": steps,
        "warmup": warmup, "ms_per_step": round(1e3 * el / steps, 2), "higher_is_better": True,
        "scaling": "strong", "dtype": "f32", "data": "synthetic", "loss": round(loss, 5), "backend": backend,
        "config": {"workload": "C5 R-MAT scale %d, one graph over %d rank(s), %s row partition"
                               % (scale, world, "cyclic" if cyclic else "degree-balanced"),
                   "nodes": n, "nnz": nnz_total, "rows_per_rank": part.n_per, "nnz_this_rank": pg.nnz,
                   "nnz_this_rank_transposed": pg.transpose().nnz, "hidden": hidden, "nfe_per_step": nfe},
        "exchange": {"ms_per_gather": round(1e3 * t_g, 4), "bytes_received_per_rank_per_gather": recv,
                     "gathers_per_step": gathers, "bytes_received_per_rank_per_step": recv * gathers,
                     "exchange_ms_per_step": round(1e3 * t_g * gathers, 3) if world > 1 else 0.0,
                     "GB_per_s_received": round(recv / t_g / 1e9, 1) if world > 1 else None,
                     "share_of_step": round(gathers * t_g / (el / steps), 3) if world > 1 else 0.0}}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--scale", type=int, default=20)
    ap.add_argument("--edges", type=int, default=10_000_000)
    ap.add_argument("--nfeat", type=int, default=128)
    ap.add_argument("--hidden", type=int, default=128)
    ap.add_argument("--nclass", type=int, default=16)
    ap.add_argument("--ode-steps", type=int, default=16)
    ap.add_argument("--cyclic", action="store_true", help="plain cyclic map instead of the degree-balanced deal")
    ap.add_argument("--backend", choices=["auto", "nccl", "gloo"], default="auto",
                    help="auto: nccl (RCCL) with a GPU per rank, else gloo with the ranks sharing the GPUs (rehearsal)")
    args = ap.parse_args()
    from graph_odenet_amd import launch
    if launch.needs_self_launch(args.gpus):
        sys.exit(launch.self_launch(__file__, sys.argv[1:], args.gpus))
    import torch.distributed as dist
    rank, _, world, dev, backend = launch.init_ranks(args.backend)
    res = run(dev, rank, world, backend, args.scale, args.edges, args.nfeat, args.hidden, args.nclass, args.ode_steps,
              args.steps, args.warmup, args.cyclic)
    if rank == 0:
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
