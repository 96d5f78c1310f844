"""Graph-format normaliser: every adjacency format the reference passes to its layers
is converted ONCE per distinct tensor object into device-resident int32 CSR plus the
nnz-balanced record list the SpMM kernel walks.

Accepted:
  * uncoalesced torch sparse COO  N x N   (GCN/utils.py:222-229)
  * dense N x N                           (GCN-dense-paper/utils.py:87)
  * sparse COO incidence  N x E  (Mtgt)   (GAT/utils.py:194-209)
  * dense incidence       N x 2M (Etgt)   (QC/datasets/utils.py:213-214)

Duplicate COO entries are summed (torch.spmm semantics on uncoalesced input).
The conversion is index bookkeeping with torch ops on whatever device the tensor lives
on; the arithmetic of the hot path is in csrc/.
"""
import weakref

import torch

DEFAULT_SPLIT = None  # auto: see auto_split()


def auto_split(nnz):
    """Longest record the SpMM kernel walks with one lane group.  Large graphs: 256 - measured on the R-MAT
    2^20 / 10.8 M-nnz graph at d=128 (tools/kbench.py --split): 64 -> 1.005 ms, 128 -> 0.925, 160 -> 0.849,
    256 -> 0.818, 320 -> 0.820, 512 -> 0.865 (fewer partial slots and finish work vs. a longer tail).
    Small graphs (citation-graph size): 1024, i.e. practically no splitting - below 65 536 records the kernel
    gives every record a whole wave (spmm_vec4_wave_kernel), which shortens the tail without a finish pass."""
    return 256 if nnz >= (1 << 20) else 1024


class CSRGraph:
    """CSR of an (n_rows x n_cols) sparse matrix + balanced record list."""

    def __init__(self, rowptr, col, val, n_rows, n_cols, split=DEFAULT_SPLIT, records=True):
        self.rowptr = rowptr.to(torch.int32).contiguous()
        self.col = col.to(torch.int32).contiguous()
        self.val = None if val is None else val.to(torch.float32).contiguous()
        self.n_rows = int(n_rows)
        self.n_cols = int(n_cols)
        self.nnz = int(self.col.numel())
        self.device = self.col.device
        self.split = int(split) if split is not None else auto_split(self.nnz)
        if records:
            self._build_items()
        else:
            # no record list: the kernels walk whole rows (one lane group / wave per row).  For matrices that are
            # rebuilt all the time and have no long rows (a QM9 mini-batch): building the list costs several host syncs.
            self.items, self.n_items, self.long_rows, self.n_long, self.n_slots = None, self.n_rows, None, 0, 0
        self._partial = {}
        self._T = None

    # -- balanced record list -------------------------------------------------
    def _build_items(self):
        dev = self.device
        n = self.n_rows
        rp = self.rowptr.to(torch.int64)
        deg = rp[1:] - rp[:-1]
        L = self.split
        nseg = torch.clamp((deg + L - 1) // L, min=1)
        total = int(nseg.sum().item()) if n > 0 else 0
        rows = torch.repeat_interleave(torch.arange(n, device=dev, dtype=torch.int64), nseg)
        first = torch.cumsum(nseg, 0) - nseg
        k = torch.arange(total, device=dev, dtype=torch.int64) - first[rows]
        begin = rp[rows] + k * L
        end = torch.minimum(begin + L, rp[rows + 1])
        is_long = nseg[rows] > 1
        slot = torch.where(is_long, torch.cumsum(is_long.to(torch.int64), 0) - 1,
                           torch.full_like(rows, -1))
        self.n_slots = int(is_long.sum().item()) if total > 0 else 0
        items = torch.stack([rows, begin, end, slot], 1)
        # longest first: better tail behaviour and less divergence inside a wave
        order = torch.argsort(end - begin, descending=True, stable=True)
        self.items = items[order].to(torch.int32).contiguous()
        self.n_items = total
        long_rows = torch.nonzero(nseg > 1).flatten()
        self.n_long = int(long_rows.numel())
        if self.n_long:
            # slots of one row are consecutive in row order
            long_first = torch.cumsum(nseg[long_rows], 0) - nseg[long_rows]
            lr = torch.stack([long_rows, long_first, long_first + nseg[long_rows],
                              torch.zeros_like(long_rows)], 1)
            self.long_rows = lr.to(torch.int32).contiguous()
        else:
            self.long_rows = None

    def partial(self, d):
        """Scratch for split rows at feature width d (owned here, reused across calls)."""
        if self.n_slots == 0:
            return None
        buf = self._partial.get(d)
        if buf is None:
            buf = torch.empty(self.n_slots * d, dtype=torch.float32, device=self.device)
            self._partial[d] = buf
        return buf

    # -- transpose (CSR by source) for the backward pass -----------------------
    def transpose(self):
        if self._T is None:
            rp = self.rowptr.to(torch.int64)
            rows = torch.repeat_interleave(torch.arange(self.n_rows, device=self.device), rp[1:] - rp[:-1])
            self._T = from_coo(self.col.to(torch.int64), rows, self.val, self.n_cols, self.n_rows,
                               split=self.split, coalesce=False)
            self._T._T = self
        return self._T

    # -- node renumbering ---------------------------------------------------------
    def degree_order(self):
        """Nodes by decreasing (row + column) degree, ties in the given order: hubs first."""
        rp = self.rowptr.to(torch.int64)
        deg = rp[1:] - rp[:-1]
        deg = deg + torch.bincount(self.col.to(torch.int64), minlength=self.n_cols)[: self.n_rows]
        return torch.argsort(deg, descending=True, stable=True)

    def relabel(self, order):
        """P A P^T for the renumbering that puts old node order[k] at position k (square matrices).  Every row keeps
        its entries in the order they have here (a CSR row need not be column-sorted), and so does every row of the
        transpose, so A' (P x) = P (A x) and A'^T (P x) = P (A^T x) BIT FOR BIT: the kernels add the same numbers in
        the same order, only the addresses of the operand rows change."""
        if self.n_rows != self.n_cols:
            raise ValueError("relabel: square matrices only")
        order = order.to(torch.int64)
        new_id = torch.empty_like(order)
        new_id[order] = torch.arange(self.n_rows, device=self.device)

        def one(g):
            rp = g.rowptr.to(torch.int64)
            deg = (rp[1:] - rp[:-1])[order]
            rowptr = torch.zeros(g.n_rows + 1, dtype=torch.int64, device=g.device)
            rowptr[1:] = torch.cumsum(deg, 0)
            idx = torch.repeat_interleave(rp[order] - rowptr[:-1], deg) + torch.arange(g.nnz, device=g.device)
            return CSRGraph(rowptr, new_id[g.col.to(torch.int64)[idx]], None if g.val is None else g.val[idx],
                            g.n_rows, g.n_cols, split=g.split)
        out, out_t = one(self), one(self.transpose())
        out._T, out_t._T = out_t, out
        return out

    def to_dense(self):
        rp = self.rowptr.to(torch.int64)
        rows = torch.repeat_interleave(torch.arange(self.n_rows, device=self.device), rp[1:] - rp[:-1])
        out = torch.zeros(self.n_rows, self.n_cols, dtype=torch.float32, device=self.device)
        v = self.val if self.val is not None else torch.ones(self.nnz, device=self.device)
        out.index_put_((rows, self.col.to(torch.int64)), v, accumulate=True)
        return out

    def algorithmic_bytes(self, d):
        """nnz*(4+4+4d) + (N+1)*4 + N*d*4 (value term dropped when pattern-only)."""
        per = 4 + (4 if self.val is not None else 0) + 4 * d
        return self.nnz * per + (self.n_rows + 1) * 4 + self.n_rows * d * 4


def from_coo(rows, cols, vals, n_rows, n_cols, split=DEFAULT_SPLIT, coalesce=True):
    rows = rows.to(torch.int64)
    cols = cols.to(torch.int64)
    key = rows * n_cols + cols
    if coalesce:
        ukey, inv = torch.unique(key, sorted=True, return_inverse=True)
        if ukey.numel() != key.numel():
            if vals is None:
                vals = torch.ones(key.numel(), dtype=torch.float32, device=key.device)
            vals = torch.zeros(ukey.numel(), dtype=torch.float32, device=key.device).index_add_(0, inv, vals.float())
        elif vals is not None:
            vals = torch.empty_like(vals, dtype=torch.float32).index_copy_(0, inv, vals.float())
        key = ukey
    else:
        order = torch.argsort(key, stable=True)
        key = key[order]
        if vals is not None:
            vals = vals[order]
    r = key // n_cols
    c = key % n_cols
    counts = torch.bincount(r, minlength=n_rows)
    rowptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=key.device)
    rowptr[1:] = torch.cumsum(counts, 0)
    return CSRGraph(rowptr, c, vals, n_rows, n_cols, split=split)


_cache = {}


def _evict(key):
    _cache.pop(key, None)


def as_graph(adj, split=DEFAULT_SPLIT):
    """Normalise `adj` (sparse COO / dense / CSRGraph); cached per tensor object."""
    if isinstance(adj, CSRGraph) or getattr(adj, "is_partitioned", False):
        return adj                              # partition.PartitionedGraph: this rank's rows of A and A^T
    if not torch.is_tensor(adj):
        raise TypeError("adjacency must be a torch tensor (sparse COO or dense) or CSRGraph")
    key = id(adj)
    hit = _cache.get(key)
    if hit is not None and hit[0]() is adj and hit[2] == adj._version:
        return hit[1]
    if adj.dim() != 2:
        raise ValueError("adjacency must be 2-D, got shape %s" % (tuple(adj.shape),))
    if adj.is_sparse:
        idx = adj._indices()
        vals = adj._values()
        g = from_coo(idx[0], idx[1], vals, adj.shape[0], adj.shape[1], split=split)
    elif adj.layout == torch.sparse_csr:
        g = CSRGraph(adj.crow_indices(), adj.col_indices(), adj.values(), adj.shape[0], adj.shape[1], split=split)
    else:
        nz = torch.nonzero(adj)
        g = from_coo(nz[:, 0], nz[:, 1], adj[nz[:, 0], nz[:, 1]], adj.shape[0], adj.shape[1], split=split)
    try:
        ref = weakref.ref(adj, lambda _r, k=key: _evict(k))
        _cache[key] = (ref, g, adj._version)
    except TypeError:
        pass
    return g


RECORDS_MIN_NNZ = 1 << 16      # assignment matrices below this size skip the record list (see CSRGraph.__init__)


def csr_from_assignment(index, n_rows, vals=None, split=DEFAULT_SPLIT, records=None):
    """CSR of the (n_rows x E) matrix with exactly one entry per column e, at row index[e] (value vals[e], or
    pattern-only).  No duplicates are possible, so this is a stable sort plus a count - no `unique`, and no host
    synchronisation unless the record list is built."""
    idx = index.to(torch.int64)
    e = idx.numel()
    if records is None:
        records = e >= RECORDS_MIN_NNZ
    if idx.is_cuda and not records and n_rows > 0:
        # a mini-batch sized matrix: counts, prefix sum and stable order in ONE launch (csrc/convert.hip) instead of a
        # sort, a count, a scan and their casts (~14 launches)
        from . import _lib
        lib = _lib.load()
        if lib.gode_assign_csr_supported(e, n_rows):
            idx = idx.contiguous()
            rowptr = torch.empty(n_rows + 1, dtype=torch.int32, device=idx.device)
            order = torch.empty(e, dtype=torch.int32, device=idx.device)
            v_in = None if vals is None else vals.to(torch.float32).contiguous()
            v = None if vals is None else torch.empty(e, dtype=torch.float32, device=idx.device)
            _lib.check(lib.gode_assign_csr_i32(_lib.ptr(idx), e, n_rows, _lib.ptr(rowptr), _lib.ptr(order), _lib.ptr(v_in),
                                               _lib.ptr(v), _lib.stream_ptr()), "gode_assign_csr_i32")
            return CSRGraph(rowptr, order, v, n_rows, e, split=split, records=False)
    order = torch.argsort(idx, stable=True)                      # edge ids grouped by row, ascending inside a row
    counts = torch.zeros(n_rows, dtype=torch.int64, device=idx.device).index_add_(
        0, idx, torch.ones(e, dtype=torch.int64, device=idx.device))
    rowptr = torch.zeros(n_rows + 1, dtype=torch.int64, device=idx.device)
    rowptr[1:] = torch.cumsum(counts, 0)
    v = None if vals is None else vals.to(torch.float32)[order]
    return CSRGraph(rowptr, order, v, n_rows, e, split=split, records=records)


def incidence_from_index(index, n_rows, split=DEFAULT_SPLIT):
    """N x E incidence with a single 1.0 per column at row index[e] (Mtgt of GAT/utils.py:194-197),
    built directly from the index vector."""
    return csr_from_assignment(index, n_rows, None, split=split)
