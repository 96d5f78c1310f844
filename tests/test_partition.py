"""Row partition of one graph over several ranks (graph_odenet_amd/partition.py): index logic on the CPU and the
host-side exchange over a world_size-2 gloo group.  The GPU parity test is tests/test_gpu_partition.py."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from graph_odenet_amd.partition import PartitionedGraph, RowPartition


def _random_adj(n, nnz, seed):
    g = torch.Generator().manual_seed(seed)
    r, c = torch.randint(0, n, (nnz,), generator=g), torch.randint(0, n, (nnz,), generator=g)
    v = torch.rand(nnz, generator=g) + 0.1
    return r, c, v


def _dense(r, c, v, n):
    return torch.zeros(n, n).index_put_((r, c), v, accumulate=True)


def test_cyclic_map_roundtrip_and_padding():
    for n, world in ((10, 2), (11, 4), (7, 8), (1, 3), (64, 8)):
        parts = [RowPartition(n, world, r) for r in range(world)]
        assert all(p.n_per * world == p.n_pad >= n for p in parts)
        x = torch.arange(n * 3, dtype=torch.float32).view(n, 3) + 1
        gathered = torch.cat([p.take(x) for p in parts])                 # what an all-gather of the blocks yields
        assert torch.equal(parts[0].scatter_back(gathered), x)
        assert sorted(torch.cat([p.local_ids() for p in parts]).tolist()) == list(range(n))
        idx = torch.tensor([i for i in (0, n // 2, n - 1)])
        for p in parts:
            pos = p.local_positions(idx)
            assert torch.equal(p.local_ids()[pos], idx[p.owned(idx)])


def test_degree_balanced_deal():
    n, world = 203, 4
    g = torch.Generator().manual_seed(5)
    # skewed pattern: low ids are hubs in both directions
    r = (torch.rand(4000, generator=g) ** 1.5 * n).long()
    c = (torch.rand(4000, generator=g) ** 1.5 * n).long()
    parts = [RowPartition.balanced(n, r, c, world, k) for k in range(world)]
    x = torch.arange(n, dtype=torch.float32).view(n, 1)
    gathered = torch.cat([p.take(x) for p in parts])
    assert torch.equal(parts[0].scatter_back(gathered), x)
    assert sorted(torch.cat([p.local_ids() for p in parts]).tolist()) == list(range(n))
    idx = torch.tensor([0, 5, 77, n - 1, 5])
    for p in parts:
        assert torch.equal(p.local_ids()[p.local_positions(idx)] if p.local_positions(idx).numel() else idx[:0],
                           idx[p.owned(idx)])
    share = [int(p.owned(r).sum()) for p in parts]
    cyc = [int(RowPartition(n, world, k).owned(r).sum()) for k in range(world)]
    assert max(share) - min(share) <= 0.08 * len(r)                      # near-equal shares of the non-zeros ...
    assert max(share) - min(share) <= max(cyc) - min(cyc)                # ... at least as even as the cyclic map


@pytest.mark.parametrize("balanced", [False, True])
def test_blocks_reassemble_to_the_matrix_and_its_transpose(balanced):
    n, world = 37, 3
    r, c, v = _random_adj(n, 300, 0)                                     # with duplicates: summed as in from_coo
    A = _dense(r, c, v, n)
    x = torch.randn(n, 5, generator=torch.Generator().manual_seed(1))
    parts = [RowPartition.balanced(n, r, c, world, k) if balanced else RowPartition(n, world, k) for k in range(world)]
    pgs = [PartitionedGraph.from_coo(r, c, v, n, p) for p in parts]
    x_full = torch.cat([p.take(x) for p in parts])
    y = torch.cat([pg.local.to_dense() @ x_full for pg in pgs])          # every rank's block product, gathered
    yt = torch.cat([pg.transpose().local.to_dense() @ x_full for pg in pgs])
    assert (parts[0].scatter_back(y) - A @ x).abs().max() < 1e-5
    assert (parts[0].scatter_back(yt) - A.t() @ x).abs().max() < 1e-5
    assert pgs[0].transpose().transpose() is pgs[0]
    assert sum(pg.nnz for pg in pgs) == int((A != 0).sum())


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    n = 23
    r, c, v = _random_adj(n, 120, 3)
    part = RowPartition(n)                                               # world / rank from the process group
    pg = PartitionedGraph.from_coo(r, c, v, n, part)
    x = torch.randn(n, 4, generator=torch.Generator().manual_seed(4))
    full = pg.gather(part.take(x))                                       # the exchange step, on host tensors
    y = pg.local.to_dense() @ full
    q.put((rank, full.tolist(), y.tolist()))
    dist.destroy_process_group()


def test_two_rank_gather_over_gloo():
    world, n = 2, 23
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(k, world, port, q)) for k in range(world)]
    [p.start() for p in procs]
    res = {k: (torch.tensor(f), torch.tensor(y)) for k, f, y in (q.get(timeout=120) for _ in range(world))}
    [p.join(60) for p in procs]
    r, c, v = _random_adj(n, 120, 3)
    x = torch.randn(n, 4, generator=torch.Generator().manual_seed(4))
    part = RowPartition(n, world, 0)
    assert torch.equal(res[0][0], res[1][0])                             # both ranks see the same gathered operand
    assert torch.equal(part.scatter_back(res[0][0]), x)
    y = part.scatter_back(torch.cat([res[0][1], res[1][1]]))
    assert (y - _dense(r, c, v, n) @ x).abs().max() < 1e-5
