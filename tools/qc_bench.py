#!/usr/bin/env python3
"""QC / QM9 data-parallel step timing (config C4 of SURVEY.md §8(d), configs[3] of BASELINE.json).

Model: the reference's `MPNN_ENN_K_Set2Set` shape (QC/layer_models.py: edge encoder 5 -> 2667 -> 73*73, input
Linear 13 -> 73, MPNN_enn_edge T=3, Set2Set(12), output Linear -> 12 targets; 14.4 M parameters, 57 MB of
gradients), batch of 20 synthetic QM9-like molecules per GPU (RDKit / QM9 files are absent), MSE loss,
Adam(lr 1e-3) as in QC/train_egcn.py:122.

  python tools/qc_bench.py                                       one GPU (+ the oracle's message step on the host)
  python tools/qc_bench.py --gpus N                              starts N ranks itself (graph_odenet_amd/launch.py), or
  python -m torch.distributed.run --nproc-per-node N --master-addr 127.0.0.1 tools/qc_bench.py --gpus N
      one rank per GPU, every rank its own batches (seed = rank), the bucketed RCCL all-reduce of the 57 MB of
      gradients per step (parallel.GradBucket); weak scaling, value = N*20*K / max-over-ranks time.  With fewer GPUs
      than ranks the ranks share the GPUs and exchange over gloo (rehearsal).

bench.py --gpus N calls run() for its `secondary.qc_data_parallel` entry.

Prints one JSON line on rank 0.  Development / secondary measurement; the contract line is bench.py's.
"""
import argparse
import json
import os
import sys
import time

os.environ.setdefault("DEBUG_CLR_GRAPH_PACKET_CAPTURE", "0")     # graph_odenet_amd/hipgraph.py: memset nodes must work for --captured

import torch
import torch.nn.functional as F

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(dev, model_name, seed, n_batches, batch_size, bucket=False, by_index=False):
    from graph_odenet_amd import qc_models
    from graph_odenet_amd.qc_batch import pad_batch
    from graph_odenet_amd.synth import qm9_like_batch
    torch.manual_seed(0)
    net = getattr(qc_models, model_name)(node_features=13, edge_features=5, target_features=12,
                                         hidden_features=73, num_layers=3).to(dev)
    batches = []
    for b in range(n_batches):
        x, ef, Esrc, Etgt, batch = qm9_like_batch(batch_size, seed=seed * 1000 + b, device=dev)
        if by_index:          # what a loader has before it builds the dense matrix: the per-edge target index
            Etgt = Etgt.argmax(0)
        if bucket:            # collate-time padding to the shape bucket (qc_batch.py): part of data loading, not of the step
            x, ef, Esrc, Etgt, batch, _ = pad_batch(x, ef, Esrc, Etgt, batch)
        tgt = torch.randn(batch_size, 12, generator=torch.Generator().manual_seed(seed * 1000 + b)).to(dev)
        batches.append((x, ef, Esrc, Etgt, batch, tgt))
    return net, batches


def cpu_oracle_step(model_name, batch_size, reps):
    """Same model on the host cores with the oracle's message step instead of the HIP one."""
    from oracle import layers_ref as R
    net, batches = build(torch.device("cpu"), model_name, 0, 1, batch_size)

    def mpnn_cpu(x, Esrc, Etgt, A):
        return R.mpnn_enn_edge(x, Esrc, Etgt, A, net.mpnn.update_net, net.mpnn.T)
    net.mpnn.forward = mpnn_cpu
    opt = torch.optim.Adam(net.parameters(), lr=1e-3)         # host baseline: torch's own optimiser
    x, ef, Esrc, Etgt, batch, tgt = batches[0]
    ts = []
    for _ in range(reps + 1):
        t0 = time.perf_counter()
        opt.zero_grad()
        F.mse_loss(net(x, ef, Esrc, Etgt, batch), tgt).backward()
        opt.step()
        ts.append(time.perf_counter() - t0)
    return sorted(ts[1:])[len(ts[1:]) // 2]


def run(dev, rank, world, backend, model="MPNN_ENN_K_Set2Set", steps=50, warmup=5, batch_size=20, captured=False,
        prepared=False, bucket_pad=False):
    """The data-parallel QC measurement; every rank calls it (collectives inside) and gets the same dict back."""
    import torch.distributed as dist
    from graph_odenet_amd.optim import Adam
    from graph_odenet_amd.parallel import GradBucket, broadcast_parameters

    # one distinct batch per step, as in training: the per-batch graph conversion is inside the timed region
    bucket_pad = bucket_pad or captured
    net, batches = build(dev, model, rank, steps + warmup, batch_size, bucket_pad, prepared)
    broadcast_parameters(net, 0)
    opt = Adam(net.parameters(), lr=1e-3)
    bucket = GradBucket(net, overlap=not captured) if (world > 1 or captured) else None     # one rank: gradients stay autograd's own tensors
    cstep = None
    if captured:
        from graph_odenet_amd.qc_step import CapturedQCStep
        cstep = CapturedQCStep(net, opt, F.mse_loss, exchange=(lambda: bucket.allreduce_mean(assume_all=True)) if world > 1 else None)

    def step(i):
        x, ef, Esrc, Etgt, batch, tgt = batches[i]
        if cstep is not None:
            return cstep(x, ef, Esrc, Etgt, batch, tgt)            # index-vector batches are converted inside the graph
        if prepared:
            from graph_odenet_amd.qc_batch import prepare
            n_graphs = tgt.shape[0] + (1 if bucket_pad else 0)
            Etgt, batch = prepare(Esrc, Etgt, batch, x.shape[0], n_graphs)          # timed: part of every step
        opt.zero_grad(set_to_none=bucket is None)                # (as qc_train.TrainStep: dropped, not zeroed, without an exchange)
        loss = F.mse_loss(net(x, ef, Esrc, Etgt, batch)[:tgt.shape[0]], tgt)
        loss.backward()
        if bucket is not None:
            bucket.allreduce_mean()
        opt.step()
        return loss

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(warmup):
        step(i)
    barrier()
    t0 = time.perf_counter()
    for i in range(steps):
        loss = step(warmup + i)
    barrier()
    el = torch.tensor([time.perf_counter() - t0], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
    if world > 1:
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
    el = el.item()
    res = {"metric": "QC edge-MPNN training throughput (graphs/s), batch %d per GPU" % batch_size,
           "value": round(world * batch_size * steps / el, 1), "unit": "graphs/s",
           "n_gpus": world, "steps": steps, "warmup": warmup,
           "ms_per_step": round(1e3 * el / steps, 3), "scaling": "weak", "dtype": "f32",
           "data": "synthetic", "loss": round(float(loss.detach()), 5), "backend": backend,
           "config": {"workload": "%s h=73 T=3, %d QM9-like molecules per rank, a new batch every step (graph conversion timed)%s%s%s"
                                  % (model, batch_size, ", padded to shape buckets" if bucket_pad else "",
                                     ", edges handed over as index vectors" if prepared else "",
                                     ", one HIP-graph replay per step and bucket" if captured else ""),
                      "params": sum(p.numel() for p in net.parameters()),
                      "gradient_bytes_allreduced_per_step": 4 * bucket.flat.numel() if world > 1 else 0,
                      "exchange_buckets": len(bucket.buckets) if bucket is not None else 0}}
    if cstep is not None:
        res["config"]["shape_buckets_captured"] = sum(1 for b in cstep.buckets.values() if b.graph is not None)
        res["config"]["shape_buckets_seen"] = len(cstep.buckets)
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch-size", type=int, default=20)
    ap.add_argument("--model", default="MPNN_ENN_K_Set2Set")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--captured", action="store_true",
                    help="one HIP-graph replay per step and shape bucket (qc_step.CapturedQCStep; implies --bucket)")
    ap.add_argument("--prepared", action="store_true",
                    help="hand the edges over as index vectors (qc_batch.prepare: no dense Etgt, no host synchronisation)")
    ap.add_argument("--bucket", action="store_true",
                    help="pad every batch to its shape bucket (multiples of 64 atoms / 128 edges, one dummy graph)")
    ap.add_argument("--backend", choices=["auto", "nccl", "gloo"], default="auto")
    args = ap.parse_args()
    from graph_odenet_amd import launch
    if launch.needs_self_launch(args.gpus):
        sys.exit(launch.self_launch(__file__, sys.argv[1:], args.gpus))
    import torch.distributed as dist
    rank, _, world, dev, backend = launch.init_ranks(args.backend)
    res = run(dev, rank, world, backend, args.model, args.steps, args.warmup, args.batch_size, args.captured, args.prepared,
              args.bucket)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            t = cpu_oracle_step(args.model, args.batch_size, 3)
            res["cpu_baseline"] = {"value": round(args.batch_size / t, 1), "unit": "graphs/s",
                                   "cores": torch.get_num_threads(), "kind": "port",
                                   "sample": "median of 3 steps after 1 warm-up, %.1f ms/step" % (1e3 * t)}
        print(json.dumps(res), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
