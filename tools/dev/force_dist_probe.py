#!/usr/bin/env python3
"""Where do the ~8 ms per step go that `bench.py --force-dist` costs on ONE rank?  The benchmark
step with the gradient exchange (a) off, (b) on with the backward-overlapped bucket, (c) on without overlap; host time of
the three phases of a step (forward + backward issue, allreduce_mean, optimiser) and the device time of the whole step."""
import os
import sys
import time

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
os.environ.setdefault("MASTER_PORT", "29541")
from graph_odenet_amd import models, parallel  # noqa: E402
from graph_odenet_amd.optim import Adam  # noqa: E402
from graph_odenet_amd.parallel import GradBucket  # noqa: E402
from graph_odenet_amd.synth import rmat_graph  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    g = rmat_graph(20, 10_000_000, seed=0, device=dev)
    g.transpose()
    n = g.n_rows
    gen = torch.Generator(device=dev).manual_seed(1000)
    x = torch.randn(n, 128, generator=gen, device=dev)
    labels = torch.randint(0, 16, (n,), generator=gen, device=dev)
    idx = torch.randperm(n, generator=gen, device=dev)[: n // 10]
    torch.manual_seed(42)
    model = models.ODEGCN3(nfeat=128, nhid=128, nclass=16, dropout=0.5, method="rk4", step_size=1 / 16).to(dev)
    opt = Adam(model.parameters(), lr=0.01, weight_decay=5e-4)
    for mode in ("alone", "overlap", "no overlap", "alone"):
        parallel.FORCE_COLLECTIVES = mode != "alone"
        bucket = GradBucket(model, overlap=(mode != "no overlap"))
        host = [0.0, 0.0, 0.0]
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]

        def step(record):
            t0 = time.perf_counter()
            model.train()
            opt.zero_grad(set_to_none=False)
            out = model(x, g)
            loss = torch.nn.functional.nll_loss(out[idx], labels[idx])
            loss.backward()
            t1 = time.perf_counter()
            bucket.allreduce_mean()
            t2 = time.perf_counter()
            opt.step()
            t3 = time.perf_counter()
            if record:
                host[0] += t1 - t0; host[1] += t2 - t1; host[2] += t3 - t2
        for _ in range(2):
            step(False)
        torch.cuda.synchronize()
        reps = 5
        t0 = time.perf_counter()
        ev[0].record()
        for _ in range(reps):
            step(True)
        ev[1].record()
        torch.cuda.synchronize()
        wall = (time.perf_counter() - t0) / reps * 1e3
        print("%-11s wall %.2f ms/step  device %.2f ms/step | host: fwd+bwd issue %.2f  allreduce_mean %.2f  optimiser %.3f ms"
              % (mode, wall, ev[0].elapsed_time(ev[1]) / reps, host[0] / reps * 1e3, host[1] / reps * 1e3, host[2] / reps * 1e3),
              flush=True)
        for h in bucket._hooks:
            h.remove()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
