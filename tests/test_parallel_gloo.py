"""N>1 path on CPU: world_size-2 gloo processes exercise the gradient bucket, parameter broadcast
and shard arithmetic that bench.py / training use with RCCL on the GPUs."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from graph_odenet_amd.parallel import GradBucket, broadcast_parameters, shard_range


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                       # different init per rank ...
    m = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
    broadcast_parameters(m, 0)                          # ... made identical here
    x = torch.randn(16, 6, generator=torch.Generator().manual_seed(rank))   # each rank its own "graph"
    y = torch.randn(16, 3, generator=torch.Generator().manual_seed(50 + rank))
    bucket = GradBucket(m)
    opt = torch.optim.Adam(m.parameters(), lr=0.01)
    for it in range(3):
        opt.zero_grad(set_to_none=(it == 1))            # step 1 drops the gradients: the bucket re-attaches its views
        ((m(x) - y) ** 2).mean().backward()
        bucket.allreduce_mean()
        assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(bucket.params, bucket.views))
        opt.step()
    q.put((rank, torch.cat([p.detach().reshape(-1) for p in m.parameters()]).tolist()))
    dist.destroy_process_group()


def test_two_rank_data_parallel_matches_single_process():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    res = {r: torch.tensor(v) for r, v in (q.get(timeout=120) for _ in range(world))}
    [p.join(60) for p in procs]
    assert torch.equal(res[0], res[1])                  # replicas stay in lock-step
    # single-process reference: mean of the two per-rank gradients each step
    torch.manual_seed(100)
    m = torch.nn.Sequential(torch.nn.Linear(6, 5), torch.nn.ReLU(), torch.nn.Linear(5, 3))
    xs = [torch.randn(16, 6, generator=torch.Generator().manual_seed(r)) for r in range(world)]
    ys = [torch.randn(16, 3, generator=torch.Generator().manual_seed(50 + r)) for r in range(world)]
    opt = torch.optim.Adam(m.parameters(), lr=0.01)
    for _ in range(3):
        opt.zero_grad()
        (sum(((m(x) - y) ** 2).mean() for x, y in zip(xs, ys)) / world).backward()
        opt.step()
    ref = torch.cat([p.detach().reshape(-1) for p in m.parameters()])
    assert (res[0] - ref).abs().max() < 1e-6


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 8, 250):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
