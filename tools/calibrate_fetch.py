#!/usr/bin/env python3
"""Known-byte-count launches for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950
(the microarch guide: FETCH_SIZE reads 1/2 of a wide coalesced stream; calibrate on your own
access pattern).  Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`.
  A) lincomb of 2 x 512 MiB streams -> reads 1 GiB, writes 512 MiB
  B) SpMM over a random PERMUTATION matrix (one nnz per row, d=128): every X row gathered exactly
     once as 512 contiguous bytes -> reads N*512 B (+ N*24 B of records/indices), writes N*512 B."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from graph_odenet_amd import graph as G, ops  # noqa: E402

dev = torch.device("cuda:0")
n, d = 1 << 20, 128
a = torch.randn(n, d, device=dev)
b = torch.randn(n, d, device=dev)
out = torch.empty(n, d, device=dev)
big = torch.empty(1 << 28, device=dev)          # 1 GiB scrub between launches (evicts the 256 MiB MALL)
for _ in range(3):
    big.fill_(1.0)
    ops.lincomb_(out, [(1.0, a), (0.5, b)])
perm = torch.randperm(n, device=dev)
g = G.from_coo(torch.arange(n, device=dev), perm, torch.ones(n, device=dev), n, n, coalesce=False)
for _ in range(3):
    big.fill_(1.0)
    ops.spmm(g, a, out=out)
torch.cuda.synchronize()
print("calibration launches done: lincomb4_kernel reads %d B writes %d B; spmm_vec4_kernel<32> reads %d B writes %d B"
      % (2 * n * d * 4, n * d * 4, n * d * 4 + n * 24, n * d * 4))
