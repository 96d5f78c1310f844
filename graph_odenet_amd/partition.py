"""One graph over several GPUs: 1-D row partition with one exchange step per aggregation (third row).

The reference has no distributed code; this is the single-model counterpart of the replica / independent-graph modes
of parallel.py for a graph (or a per-step latency target) that one GPU does not serve.

Layout.  A RowPartition maps node i to (rank, slot): cyclic by default (rank i mod P, slot i div P), or dealt by
decreasing degree (RowPartition.balanced) so that the row blocks of A and A^T carry equal shares of the non-zeros -
R-MAT and citation graphs keep their hubs at low indices and R-MAT skews every index bit, so neither contiguous
blocks nor the cyclic map balance them.  Every rank holds
  * its n_per = ceil(N / P) rows of every N x d node matrix (features, ODE state, adjoint state; tail slots of a
    ragged last block are empty rows that no edge references),
  * the rows of A it owns AND the rows of A^T it owns, both as n_per x (P n_per) CSR blocks with columns renumbered
    to (owner, slot) order, i.e. to the order an all-gather of the local blocks produces.
Everything that is row-local in the path (GroupNorm, the dense products, relu, the Runge-Kutta combinations, the
loss) runs unchanged on the local rows.  The one non-local operation is the aggregation  Y = A X:
    X_full = all_gather(X_local)                 (RCCL over xGMI; (P-1)/P x N x d x 4 bytes received per rank)
    Y_local = A[rows of this rank, :] @ X_full   (the ordinary SpMM kernel on the rectangular block)
and its transpose in the backward pass is the same pattern on the block of A^T (an all-gather of the cotangent rows
instead of a reduce-scatter of N x d partial sums: same volume, no atomics, no second kernel).  Parameter gradients
are per-rank partial sums over the local rows: one flat all-reduce(SUM) per step (parallel.GradBucket.allreduce_sum).

`ops.spmm` recognises a PartitionedGraph and performs the gather, so layers.py / gcn_ode.py / functional.py need no
second code path; the whole-solve C drivers and HIP-graph capture are bypassed (a collective sits between the two
kernels of every f-eval) and the solvers take their per-stage paths.  Fixed grid (rk4): nothing else is exchanged.
Adaptive (dopri5): the step-size controller must see the same error norms on every rank, so the per-component sums of
squares of the row-sliced components are all-reduced once per step (a handful of doubles), and the small components of
the adjoint state (a_t, parameter gradients) are kept GLOBAL by summing every stage's small derivatives over the ranks
(66 KB at d = 128); every rank then takes identical accept / reject decisions.

Cost model at the benchmark scale (N = 2^20, d = 128, P = 8): 448 MB received per rank per aggregation over seven
xGMI links (~1 TB/s aggregate at best, ~0.45 ms) against ~0.12 ms for the local SpMM block: communication-bound, as
SURVEY §8(e) predicts; the mode is for capacity (graphs beyond one GPU's 288 GB), not for speed at this size.
"""
import torch
import torch.distributed as dist

from .graph import from_coo


class RowPartition:
    """Node -> (rank, slot) map for N nodes over `world` ranks, every rank with n_per = ceil(N / world) slots.

    Default: cyclic (node i on rank i mod P, slot i div P).  With `order` (a permutation of the nodes, identical on
    every rank) the nodes are dealt in that order, boustrophedon over the ranks (0..P-1, P-1..0, ...): `balanced()`
    passes the nodes by decreasing degree, which evens out the non-zeros of the row blocks of A and of A^T - on an
    R-MAT graph every index bit is skewed, so the plain cyclic map still leaves rank 0 with 0.76^log2(P) of them."""

    def __init__(self, n, world=None, rank=None, group=None, order=None):
        if world is None:
            world = dist.get_world_size(group) if dist.is_initialized() else 1
        if rank is None:
            rank = dist.get_rank(group) if dist.is_initialized() else 0
        if not (0 <= rank < world) or n < 0:
            raise ValueError("RowPartition: rank %d of %d, n = %d" % (rank, world, n))
        self.n, self.world, self.rank, self.group = int(n), int(world), int(rank), group
        self.n_per = -(-self.n // self.world) if self.n else 0
        self.n_pad = self.n_per * self.world
        self._pos = None                            # node -> position in the gathered (owner-major) order
        self._pos_by_device = {}
        if order is not None:
            order = order.to(torch.int64)
            if order.numel() != self.n:
                raise ValueError("RowPartition: order must be a permutation of the %d nodes" % self.n)
            k = torch.arange(self.n, device=order.device)
            rnd, j = k // self.world, k % self.world
            owner = torch.where(rnd % 2 == 0, j, self.world - 1 - j)
            self._pos = torch.empty(self.n, dtype=torch.int64, device=order.device)
            self._pos[order] = owner * self.n_per + rnd
            if int(torch.bincount(order, minlength=self.n).max()) != 1:
                raise ValueError("RowPartition: order is not a permutation")

    @staticmethod
    def balanced(n, rows, cols, world=None, rank=None, group=None):
        """Deal the nodes by decreasing (out-degree + in-degree) of the COO pattern (rows, cols)."""
        w = torch.bincount(rows.to(torch.int64), minlength=n) + torch.bincount(cols.to(torch.int64), minlength=n)
        order = torch.sort(w, descending=True, stable=True).indices
        return RowPartition(n, world, rank, group, order=order)

    def _pos_on(self, device):
        t = self._pos_by_device.get(device)
        if t is None:
            t = self._pos_by_device[device] = self._pos.to(device)
        return t

    def renumber(self, idx):
        """Global node ids -> positions in the gathered (owner-major) order."""
        if self._pos is not None:
            return self._pos_on(idx.device)[idx]
        return (idx % self.world) * self.n_per + idx // self.world

    def owned(self, idx):
        if self._pos is not None:
            return (self._pos_on(idx.device)[idx] // max(self.n_per, 1)) == self.rank
        return (idx % self.world) == self.rank

    def local_ids(self, device=None):
        """Global ids of the real nodes of this rank, in local slot order."""
        if self._pos is not None:
            pos = self._pos_on(torch.device(device) if device is not None else self._pos.device)
            mine = torch.nonzero((pos // max(self.n_per, 1)) == self.rank).flatten()
            return mine[torch.argsort(pos[mine])]
        if self.rank >= self.n:                   # more ranks than nodes: this rank holds padding only
            return torch.empty(0, dtype=torch.int64, device=device)
        return torch.arange(self.rank, self.n, self.world, device=device)

    def take(self, x):
        """Rows of a global node tensor that this rank owns, in slot order, zero-padded to n_per rows."""
        ids = self.local_ids(x.device)
        out = x.new_zeros((self.n_per,) + tuple(x.shape[1:]))
        out[self.renumber(ids) - self.rank * self.n_per] = x[ids]
        return out

    def local_positions(self, idx):
        """Local slots of those entries of the global id list `idx` that this rank owns (order kept)."""
        idx = idx[self.owned(idx)]
        return self.renumber(idx) - self.rank * self.n_per

    def scatter_back(self, x_full):
        """Inverse of the gathered order: rows of an (n_pad x ...) gathered tensor back in global node order."""
        ids = torch.arange(self.n, device=x_full.device)
        return x_full[self.renumber(ids)]


class PartitionedGraph:
    """This rank's rows of A and of A^T (see the module docstring).  Quacks like a square CSRGraph over the LOCAL
    rows: n_rows = n_cols = n_per; `ops.spmm(pg, X_local)` gathers and multiplies."""

    is_partitioned = True

    def __init__(self, block, block_t, part):
        if block.n_rows != part.n_per or block.n_cols != part.n_pad or block_t.n_rows != part.n_per \
                or block_t.n_cols != part.n_pad:
            raise ValueError("PartitionedGraph: blocks must be n_per x n_pad")
        self.local, self.local_t, self.part = block, block_t, part
        self.n_rows = self.n_cols = part.n_per
        self.device = block.device
        self._T = None
        self._bufs = {}

    @property
    def nnz(self):
        return self.local.nnz

    def transpose(self):
        if self._T is None:
            self._T = PartitionedGraph(self.local_t, self.local, self.part)
            self._T._T = self
            self._T._bufs = self._bufs           # one gather buffer per width serves both directions
        return self._T

    def gather(self, x):
        """(n_per x d) local rows -> (n_pad x d) rows of every rank in owner-major order."""
        p = self.part
        if x.dim() != 2 or x.shape[0] != p.n_per or x.dtype != torch.float32:
            raise ValueError("PartitionedGraph.gather: expected an fp32 %d x d block, got %s" % (p.n_per, tuple(x.shape)))
        x = x.contiguous()
        if p.world == 1:
            return x
        d = x.shape[1]
        buf = self._bufs.get(d)
        if buf is None or buf.device != x.device:
            buf = self._bufs[d] = torch.empty(p.n_pad, d, dtype=torch.float32, device=x.device)
        if dist.get_backend(p.group) == "gloo" and x.is_cuda:
            # rehearsal on one box (several ranks sharing a GPU): the exchange is staged through the host
            host = torch.empty(p.n_pad, d, dtype=torch.float32)
            dist.all_gather(list(host.chunk(p.world)), x.cpu(), group=p.group)
            buf.copy_(host)
        else:
            dist.all_gather_into_tensor(buf, x, group=p.group)
        return buf

    @staticmethod
    def from_coo(rows, cols, vals, n, part, device=None):
        """Build this rank's blocks from the (global) COO triplets of the N x N matrix; duplicates are summed as in
        graph.from_coo.  Every rank may pass the full edge list or only the entries whose row OR column it owns."""
        device = device if device is not None else rows.device
        rows, cols = rows.to(device=device, dtype=torch.int64), cols.to(device=device, dtype=torch.int64)
        vals = vals.to(device=device, dtype=torch.float32) if vals is not None else None
        if rows.numel() and (int(rows.max()) >= n or int(cols.max()) >= n or int(rows.min()) < 0 or int(cols.min()) < 0):
            raise ValueError("PartitionedGraph.from_coo: index out of range")
        r2, c2 = part.renumber(rows), part.renumber(cols)
        base = part.rank * part.n_per
        sel = part.owned(rows)
        block = from_coo(r2[sel] - base, c2[sel], vals[sel] if vals is not None else None, part.n_per, part.n_pad)
        sel = part.owned(cols)
        block_t = from_coo(c2[sel] - base, r2[sel], vals[sel] if vals is not None else None, part.n_per, part.n_pad)
        return PartitionedGraph(block, block_t, part)

    @staticmethod
    def from_adj(adj, part, device=None):
        """From a sparse COO tensor (coalesced or not) or a dense N x N tensor."""
        if adj.dim() != 2 or adj.shape[0] != adj.shape[1] or adj.shape[0] != part.n:
            raise ValueError("PartitionedGraph.from_adj: adjacency must be %d x %d" % (part.n, part.n))
        if adj.is_sparse:
            idx, v = adj._indices(), adj._values()
            return PartitionedGraph.from_coo(idx[0], idx[1], v, part.n, part, device)
        nz = torch.nonzero(adj)
        return PartitionedGraph.from_coo(nz[:, 0], nz[:, 1], adj[nz[:, 0], nz[:, 1]], part.n, part, device)


def global_sum(t, group=None):
    """Sum of a (scalar / small) tensor over the ranks, in place; identity without a process group."""
    if dist.is_initialized() and dist.get_world_size(group) > 1:
        if dist.get_backend(group) == "gloo" and t.is_cuda:
            h = t.cpu()
            dist.all_reduce(h, group=group)
            t.copy_(h)
        else:
            dist.all_reduce(t, group=group)
    return t
