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
