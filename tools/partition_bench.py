#!/usr/bin/env python3
"""ONE R-MAT graph (config C5) split by rows over the ranks: the strong-scaling counterpart of bench.py's
one-graph-per-rank weak scaling (third row; graph_odenet_amd/partition.py).

  python tools/partition_bench.py                         one rank: the per-stage solver path on the whole graph
  python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/partition_bench.py --gpus N
      N ranks, one GPU each: every f-eval all-gathers the N x d operand over RCCL; the parameter gradients are
      summed with one flat all-reduce per step.

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
    ap.add_argument("--backend", choices=["nccl", "gloo"], default="nccl",
                    help="gloo: rehearsal with all ranks on GPU 0 and the exchange staged through the host")
    args = ap.parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1 and args.gpus > 1:
        raise SystemExit("launch with torch.distributed.run, one rank per GPU")
    import torch.distributed as dist
    if args.backend == "gloo":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "gloo":
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    from graph_odenet_amd import models
    from graph_odenet_amd.parallel import GradBucket, broadcast_parameters
    from graph_odenet_amd.partition import PartitionedGraph, RowPartition, global_sum
    from graph_odenet_amd.synth import rmat_coo

    # every rank draws the same graph / features (seed 0) and keeps its rows
    r, c, v, n = rmat_coo(args.scale, args.edges, seed=0, device=dev)
    part = RowPartition(n, world, rank) if args.cyclic else RowPartition.balanced(n, r, c, world, rank)
    pg = PartitionedGraph.from_coo(r, c, v, n, part, device=dev)
    pg.transpose()
    nnz_total = int(r.numel())
    del r, c, v
    gen = torch.Generator(device=dev).manual_seed(1000)
    x = part.take(torch.randn(n, args.nfeat, generator=gen, device=dev))
    labels = part.take(torch.randint(0, args.nclass, (n,), generator=gen, device=dev))
    train = torch.randperm(n, generator=gen, device=dev)[: n // 10]
    pos = part.local_positions(train)
    n_train = train.numel()

    torch.manual_seed(42)
    model = models.ODEGCN3(nfeat=args.nfeat, nhid=args.hidden, nclass=args.nclass, dropout=0.5,
                           method="rk4", step_size=1.0 / args.ode_steps).to(dev)
    broadcast_parameters(model, 0)
    opt = torch.optim.Adam(model.parameters(), lr=0.01, weight_decay=5e-4)
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

    for _ in range(args.warmup):
        step()
    model.nfe = 0
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        loss = step()
    barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    el = el.item()
    nfe = model.nfe / max(args.steps, 1)
    loss = float(global_sum(loss.detach().clone()))

    # the exchange alone
    probe = torch.randn(part.n_per, args.hidden, device=dev)
    for _ in range(3):
        pg.gather(probe)
    barrier()
    t0 = time.perf_counter()
    for _ in range(20):
        pg.gather(probe)
    barrier()
    t_g = (time.perf_counter() - t0) / 20
    recv = (world - 1) * part.n_per * args.hidden * 4
    if rank == 0:
        gathers = args.steps and (3 * 4 * args.ode_steps + 4)       # fwd eval + (recompute + transpose) per adjoint stage, + 2 layers fwd/bwd
        print(json.dumps({
            "metric": "ONE row-partitioned graph: ODE-GCN forward+backward steps/sec at 64 RK4 evals",
            "value": round(args.steps / el, 4), "unit": "steps/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": round(1e3 * el / args.steps, 2), "higher_is_better": True,
            "scaling": "strong", "dtype": "f32", "data": "synthetic", "loss": round(loss, 5),
            "config": {"workload": "C5 R-MAT scale %d, one graph over %d rank(s), cyclic row partition" % (args.scale, world),
                       "nodes": n, "nnz": nnz_total, "rows_per_rank": part.n_per, "nnz_this_rank": pg.nnz, "nnz_this_rank_transposed": pg.transpose().nnz,
                       "hidden": args.hidden, "nfe_per_step": nfe},
            "exchange": {"ms_per_gather": round(1e3 * t_g, 4), "bytes_received_per_rank": recv,
                         "GB_per_s_received": round(recv / t_g / 1e9, 1) if world > 1 else None,
                         "gathers_per_step": gathers,
                         "share_of_step": round(gathers * t_g / (el / args.steps), 3) if world > 1 else 0.0}}), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
