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


class _Three(torch.nn.Module):
    """Three equal-sized parameters, each alone in its bucket (bucket_bytes = 8): `b`, in the MIDDLE bucket, is used on
    rank 0 only.  Round 2's bucket launched whatever completed first, so rank 0 issued c, b, a and rank 1 c, a, (b in
    _finish): equal sizes pair up silently and a.grad / b.grad came out swapped-and-summed."""

    def __init__(self):
        super().__init__()
        self.a = torch.nn.Parameter(torch.tensor([1.0, 2.0]))
        self.b = torch.nn.Parameter(torch.tensor([3.0, 4.0]))
        self.c = torch.nn.Parameter(torch.tensor([5.0, 6.0]))

    def forward(self, x, use_b):
        y = (self.a * x).sum() * 3.0 + (self.c * x).sum() * 7.0
        if use_b:
            y = y + (self.b * x).sum() * 9.0
        return y


def _worker_middle(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = _Three()
    bucket = GradBucket(m, bucket_bytes=8, overlap=True)
    assert len(bucket.buckets) == 3 and [idx for (_, _, _, idx) in bucket.buckets] == [[2], [1], [0]]
    x = torch.tensor([1.0, 0.5]) * (rank + 1)
    out = []
    for step in range(2):
        for p in m.parameters():
            if p.grad is not None:
                p.grad.zero_()
        m(x, rank == 0).backward()
        during_backward = list(bucket.launch_log)
        bucket.allreduce_mean()
        out.append((during_backward, [None if p.grad is None else p.grad.tolist() for p in m.parameters()]))
    q.put((rank, out))
    dist.destroy_process_group()


def test_rank_local_parameter_alone_in_a_middle_bucket():
    """Launch order must be the bucket index order on every rank (RCCL / gloo pair collectives by issue order)."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_middle, args=(r, world, port, q)) for r in range(world)]
    [p.start() for p in procs]
    got = dict(q.get(timeout=120) for _ in range(world))
    [p.join(60) for p in procs]
    # x_0 = (1, .5), x_1 = (2, 1): mean gradients a: 3 * 1.5 x_0, c: 7 * 1.5 x_0, b: 9 * x_0 / 2 (rank 1 counts as zero)
    want = [[4.5, 2.25], [4.5, 2.25], [10.5, 5.25]]
    for r in range(world):
        for during, grads in got[r]:
            assert grads == want, (r, grads)
    # rank 0 had every bucket complete during backward; rank 1 had to hold a's bucket (index 2) back behind b's
    assert got[0][0][0] == [0, 1, 2] and got[1][0][0] == [0]


def _qc_shapes(h=73, hid=2667):
    """Parameter shapes of the reference's 14.3 M-parameter QC model (EdgeEncoderMLP 5 -> hid -> h*h dominates:
    QC/layers.py:65-86): 57 MB of fp32 gradients.  A parameter is never cut, so the 56.8 MB
    weight of the second encoder layer closes the bucket it falls in: 2 buckets at the default 16 MB, 4+ at 64 KB."""
    return [(h, 13), (h,), (hid, 5), (hid,), (h * h, hid), (h * h,), (3 * h, 2 * h), (3 * h, h), (3 * h,), (3 * h,),
            (h, h), (h,), (12, h), (12,)]


def _worker_qc(rank, world, port, q, bucket_bytes):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    m = torch.nn.Module()
    m.ps = torch.nn.ParameterList([torch.nn.Parameter(torch.zeros(s)) for s in _qc_shapes()])
    bucket = GradBucket(m, bucket_bytes=bucket_bytes, overlap=False)     # CapturedQCStep: no launch from a hook
    nb = len(bucket.buckets)
    sizes = [hi - lo for (lo, _, hi, _) in bucket.buckets]
    # a graph replay writes the gradients into the kept `.grad` views; no hook fires
    g = torch.Generator().manual_seed(7 + rank)
    for p in bucket.params:
        p.grad.copy_(torch.randn(p.shape, generator=g))
    bucket.allreduce_mean(assume_all=True)
    chk = [float(p.grad.double().sum()) for p in bucket.params] + [float(bucket.params[4].grad[123, 456])]
    q.put((rank, nb, sizes, chk, list(bucket.launch_log)))
    dist.destroy_process_group()


@pytest.mark.parametrize("bucket_bytes,n_buckets", [(16 << 20, 2), (64 << 10, 4)])
def test_qc_sized_gradient_exchange_after_a_graph_replay(bucket_bytes, n_buckets):
    """57 MB, gradients written behind the hooks' back (HIP-graph replay): allreduce_mean(assume_all=True)
    as qc_step.CapturedQCStep calls it."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_qc, args=(r, world, port, q, bucket_bytes)) for r in range(world)]
    [p.start() for p in procs]
    got = [q.get(timeout=300) for _ in range(world)]
    [p.join(60) for p in procs]
    (r0, nb, sizes, chk0, log0), (r1, _, _, chk1, log1) = got
    assert nb == n_buckets and sum(sizes) * 4 > 57e6 and log0 == log1 == []      # _finish reset the log
    assert chk0 == chk1
    ref = []
    for r in range(world):
        g = torch.Generator().manual_seed(7 + r)
        ref.append([torch.randn(s, generator=g) for s in _qc_shapes()])
    mean = [(a + b) / 2 for a, b in zip(*ref)]
    want = [float(t.double().sum()) for t in mean] + [float(mean[4][123, 456])]
    assert max(abs(a - b) for a, b in zip(chk0, want)) < 1e-3
