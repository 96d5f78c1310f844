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


class _Shape(torch.nn.Module):
    """CPU stand-in with the parameter list of the benchmark model (ODEGCN3: gc1, ODE block with GroupNorm + (d+1) x d
    weight, gc3) - the product kernels have no CPU path, the DISTRIBUTED logic under test does not care what computes
    the gradients - plus one parameter no rank ever uses and one that only rank 0 uses."""

    def __init__(self, nfeat=6, d=8, ncls=3):
        super().__init__()
        self.gc1 = torch.nn.Linear(nfeat, d)
        self.norm1 = torch.nn.GroupNorm(4, d)
        self.ode_w = torch.nn.Parameter(torch.randn(d + 1, d) * 0.3)
        self.ode_b = torch.nn.Parameter(torch.zeros(d))
        self.gc3 = torch.nn.Linear(d, ncls)
        self.unused = torch.nn.Parameter(torch.ones(5))
        self.rank0_only = torch.nn.Parameter(torch.ones(d))

    def forward(self, x, use_extra):
        h = torch.relu(self.gc1(x))
        for t in (0.0, 0.5):
            z = torch.cat([torch.full_like(h[:, :1], t), self.norm1(h)], 1) @ self.ode_w + self.ode_b
            h = h + 0.5 * torch.relu(z)
        if use_extra:
            h = h * self.rank0_only
        return torch.log_softmax(self.gc3(h), 1)


def _data(rank):
    g = torch.Generator().manual_seed(rank)
    return torch.randn(16, 6, generator=g), torch.randint(0, 3, (16,), generator=g)


def _worker(rank, world, port, q, overlap, bucket_bytes):
    from graph_odenet_amd.parallel import run_timed
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(100 + rank)                       # different init per rank ...
    m = _Shape()
    broadcast_parameters(m, 0)                          # ... made identical here
    x, y = _data(rank)                                  # each rank its own "graph"
    bucket = GradBucket(m, bucket_bytes=bucket_bytes, overlap=overlap)
    assert len(bucket.buckets) == (1 if bucket_bytes > 4096 else len(bucket.buckets)) and len(bucket.buckets) >= 1
    opt = torch.optim.Adam(m.parameters(), lr=0.01, weight_decay=5e-4)
    it = [0]

    def step():                                         # the step function of bench.py, on the stand-in
        opt.zero_grad(set_to_none=(it[0] == 1))         # step 1 drops the gradients: the bucket re-attaches its views
        it[0] += 1
        loss = torch.nn.functional.nll_loss(m(x, rank == 0), y)
        loss.backward()
        bucket.allreduce_mean()
        assert m.unused.grad is None                    # no rank produced one: stays None, Adam must not decay it
        assert all(p.grad is None or p.grad.data_ptr() == v.data_ptr() for p, v in zip(bucket.params, bucket.views))
        opt.step()
        return loss
    elapsed, _ = run_timed(step, 2, 1)                  # warm-up + timed steps, barrier-bracketed, max over ranks
    q.put((rank, torch.cat([p.detach().reshape(-1) for p in m.parameters()]).tolist(), elapsed))
    dist.destroy_process_group()


import pytest  # noqa: E402


@pytest.mark.parametrize("overlap,bucket_bytes", [(True, 64), (True, 16 << 20), (False, 64)])
def test_two_rank_data_parallel_matches_single_process(overlap, bucket_bytes):
    """bench.py's training step (broadcast, step = zero_grad / forward / backward / GradBucket.allreduce_mean / Adam
    with weight decay, run_timed) on two gloo ranks against the single-process mean-gradient run; several small
    buckets launched from the backward hooks, one big bucket, and the non-overlapped mode."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q, overlap, bucket_bytes)) for r in range(world)]
    [p.start() for p in procs]
    got = [q.get(timeout=120) for _ in range(world)]
    [p.join(60) for p in procs]
    res = {r: torch.tensor(v) for r, v, _ in got}
    times = [t for _, _, t in got]
    assert times[0] == times[1] and times[0] > 0        # the max over ranks is what every rank reports
    assert torch.equal(res[0], res[1])                  # replicas stay in lock-step
    # single-process reference: mean of the two per-rank gradients each step
    torch.manual_seed(100)
    m = _Shape()
    opt = torch.optim.Adam(m.parameters(), lr=0.01, weight_decay=5e-4)
    for _ in range(3):
        opt.zero_grad()
        (sum(torch.nn.functional.nll_loss(m(*_data(r)[:1], r == 0), _data(r)[1]) for r in range(world)) / world).backward()
        opt.step()
    assert m.unused.grad is None and bool((m.unused == 1).all())
    ref = torch.cat([p.detach().reshape(-1) for p in m.parameters()])
    assert (res[0] - ref).abs().max() < 1e-6


def test_shard_range_covers_everything():
    for n in (0, 1, 7, 8, 250):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert max(h - l for l, h in spans) - min(h - l for l, h in spans) <= 1
