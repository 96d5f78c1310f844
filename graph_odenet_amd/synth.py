"""Synthetic inputs of the benchmark configurations; data only, no arithmetic
of the hot path.  Generated on the device with torch's generator so that no host->device copy
of 10M edges is needed."""
import torch

from .graph import from_coo


def rmat_edges(scale, n_edges, seed=0, abcd=(0.57, 0.19, 0.19, 0.05), device="cpu"):
    """R-MAT (Chakrabarti et al. 2004) edge list with 2^scale nodes; duplicates NOT removed here."""
    g = torch.Generator(device=device).manual_seed(seed)
    a, b, c, _ = abcd
    src = torch.zeros(n_edges, dtype=torch.int64, device=device)
    dst = torch.zeros(n_edges, dtype=torch.int64, device=device)
    for _ in range(scale):
        u = torch.rand(n_edges, generator=g, device=device)
        right = (u >= a) & (u < a + b) | (u >= a + b + c)          # column bit
        down = (u >= a + b)                                        # row bit
        src = (src << 1) | down.to(torch.int64)
        dst = (dst << 1) | right.to(torch.int64)
    return src, dst


def rmat_coo(scale, n_edges, seed=0, device="cpu", self_loops=True, normalize=True):
    """The matrix of rmat_graph as sorted COO triplets (rows, cols, values or None, n)."""
    n = 1 << scale
    src, dst = rmat_edges(scale, n_edges, seed, device=device)
    if self_loops:
        ar = torch.arange(n, device=device)
        src = torch.cat([src, ar]); dst = torch.cat([dst, ar])
    key = torch.unique(src * n + dst)
    r, c = key // n, key % n
    v = 1.0 / torch.bincount(r, minlength=n).to(torch.float32)[r] if normalize else None
    return r, c, v, n


def rmat_graph(scale, n_edges, seed=0, device="cpu", self_loops=True, normalize=True, split=None):
    """C5 of SURVEY.md §8(d): R-MAT, duplicates removed, self-loops added, row-normalised fp32."""
    n = 1 << scale
    src, dst = rmat_edges(scale, n_edges, seed, device=device)
    if self_loops:
        ar = torch.arange(n, device=device)
        src = torch.cat([src, ar]); dst = torch.cat([dst, ar])
    key = torch.unique(src * n + dst)
    r, c = key // n, key % n
    if normalize:
        deg = torch.bincount(r, minlength=n).to(torch.float32)
        v = 1.0 / deg[r]
    else:
        v = None
    kw = {} if split is None else {"split": split}
    return from_coo(r, c, v, n, n, coalesce=False, **kw)


def qm9_like_batch(n_graphs=20, seed=0, device="cpu", offset_indices=True, h_edge=5, n_node_feat=13):
    """C4 of SURVEY.md §8(d): a batch of synthetic molecule-like graphs (RDKit / QM9 files are absent):
    per graph n ~ U{9..29} atoms, a random spanning tree plus ~2 ring closures, both edge directions,
    13 node features, 5 edge features [dist ~ U(1,1.6), one-hot(4) bond type].
    Returns (x[N,13], edge_feat[E,5], Esrc[E] int64, Etgt dense [N,E], batch[N]) in the layout of the
    reference's collate (QC/datasets/utils.py:153-217).  offset_indices=False reproduces quirk Q5
    (per-graph local node ids, so every edge indexes the first <= 29 rows)."""
    g = torch.Generator().manual_seed(seed)
    xs, efs, srcs, tgts, batch = [], [], [], [], []
    n_acc = 0
    for b in range(n_graphs):
        n = int(torch.randint(9, 30, (1,), generator=g))
        parent = [int(torch.randint(0, i, (1,), generator=g)) for i in range(1, n)]
        und = [(i + 1, p) for i, p in enumerate(parent)]
        for _ in range(2):
            a, c = (int(v) for v in torch.randint(0, n, (2,), generator=g))
            if a != c:
                und.append((a, c))
        e = torch.tensor(und + [(c, a) for a, c in und], dtype=torch.int64)
        off = n_acc if offset_indices else 0
        srcs.append(e[:, 0] + off)
        tgts.append(e[:, 1] + off)
        m = e.shape[0] // 2
        dist = torch.rand(m, generator=g) * 0.6 + 1.0
        bond = torch.nn.functional.one_hot(torch.randint(0, 4, (m,), generator=g), 4).float()
        ef = torch.cat([dist[:, None], bond], 1)
        efs.append(torch.cat([ef, ef]))
        xs.append(torch.randn(n, n_node_feat, generator=g))
        batch.append(torch.full((n,), b, dtype=torch.int64))
        n_acc += n
    x, ef = torch.cat(xs), torch.cat(efs)
    Esrc, etgt = torch.cat(srcs), torch.cat(tgts)
    Etgt = torch.zeros(n_acc, Esrc.numel())
    Etgt[etgt, torch.arange(Esrc.numel())] = 1.0
    return x.to(device), ef.to(device), Esrc.to(device), Etgt.to(device), torch.cat(batch).to(device)
