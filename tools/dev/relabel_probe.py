#!/usr/bin/env python3
"""SpMM locality probe 2 (second suggestion): node renumbering of the R-MAT benchmark graph.
A' = P A P^T for P = identity / hubs first (by in-degree of the gathered side) / random; the SpMM's time and the
L2-side picture depend only on which operand rows are hot together, so this tells what a renumbering can buy."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from graph_odenet_amd import graph as G, ops  # noqa: E402
from graph_odenet_amd.synth import rmat_coo  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tools"))
from kbench import timeit  # noqa: E402

dev = torch.device("cuda:0")
r, c, v, n = rmat_coo(20, 10_000_000, seed=0, device=dev)
d = 128
X = torch.randn(n, d, device=dev)
Y = torch.empty(n, d, device=dev)
indeg = torch.bincount(c, minlength=n)
orders = {"natural": torch.arange(n, device=dev),
          "hubs first (column degree)": torch.argsort(indeg, descending=True, stable=True),
          "random": torch.randperm(n, device=dev, generator=torch.Generator(device=dev).manual_seed(1))}
# round 3: orders that keep the generator's locality (R-MAT ids share neighbourhoods by bit prefix) but break the
# power-of-two alignment of the hot ids (ids with few set bits all fall into the same 32 KiB residue)
ar = torch.arange(n, device=dev)
def from_new_id(nid):                       # order[k] = old id at position k
    o = torch.empty(n, dtype=torch.int64, device=dev); o[nid] = ar; return o
top = torch.argsort(indeg, descending=True, stable=True)
def hubs_then(rest_key, k):                 # the k hottest first, the others by rest_key
    is_top = torch.zeros(n, dtype=torch.bool, device=dev); is_top[top[:k]] = True
    rest = ar[~is_top]
    return torch.cat([top[:k], rest[torch.argsort(rest_key[rest], stable=True)]])
sw10 = ar ^ ((ar >> 10) & 1023)
sw7 = ar ^ ((ar >> 7) & 127)
sw_all = ar ^ ((ar >> 10) & 1023) ^ ((ar >> 5) & 31)
orders.update({
    "xor-swizzle low10 ^= bits 10-19": from_new_id(sw10),
    "xor-swizzle low7 ^= bits 7-13": from_new_id(sw7),
    "xor-swizzle 10 + 5": from_new_id(sw_all),
    "8k hubs, rest natural": hubs_then(ar, 8192),
    "8k hubs, rest swizzled": hubs_then(sw10, 8192),
    "64k hubs, rest swizzled": hubs_then(sw10, 65536),
    "degree class, natural inside": torch.argsort(-(torch.log2(indeg.float().clamp_min(1)).floor().long()), stable=True),
})
only = sys.argv[1:]
for name, order in orders.items():
    if only and not any(o in name for o in only):
        continue
    new_id = torch.empty(n, dtype=torch.int64, device=dev)
    new_id[order] = torch.arange(n, device=dev)
    g = G.from_coo(new_id[r], new_id[c], v, n, n, coalesce=False)
    gb = g.algorithmic_bytes(d) / 1e9
    for nm, gr in (("A", g), ("A^T", g.transpose())):
        t = timeit(lambda: ops.spmm(gr, X, out=Y), n=20)
        print("%-28s %-4s %7.3f ms  %7.1f GB/s (B_alg)" % (name, nm, t, gb / t * 1e3), flush=True)
    if name not in ("natural", "hubs first (column degree)", "random"):
        continue
    # record order: longest first (default) vs row order
    g2 = G.from_coo(new_id[r], new_id[c], v, n, n, coalesce=False)
    it = g2.items
    order2 = torch.argsort(it[:, 0].long(), stable=True)
    g2.items = it[order2].contiguous()
    t = timeit(lambda: ops.spmm(g2, X, out=Y), n=20)
    print("%-28s %-4s %7.3f ms  %7.1f GB/s (records in row order)" % (name, "A", t, gb / t * 1e3), flush=True)
