#!/usr/bin/env python3
"""SpMM locality probe at the C5 size: the d = 128 product as column slabs (w = 64 / 32 / 16 columns per launch), with
the gathered operand (a) left row-major N x 128 (strided slab, ldx = 128) or (b) stored slab-major [d/w][N][w]
(contiguous N x w tables of 256 / 128 / 64 MB - at or below the 256 MB Infinity Cache).  VERDICT r01 item 4."""
import ctypes
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from graph_odenet_amd import _lib, ops  # noqa: E402
from graph_odenet_amd.synth import rmat_graph  # noqa: E402
sys.path.insert(0, os.path.join(ROOT, "tools"))
from kbench import timeit  # noqa: E402


def raw_spmm(lib, g, xptr, ldx, yptr, ldy, d):
    part = g.partial(d)
    rc = lib.gode_spmm_csr_f32(_lib.ptr(g.rowptr), _lib.ptr(g.col), _lib.ptr(g.val), _lib.ptr(g.items), g.n_items,
                               _lib.ptr(g.long_rows), g.n_long, _lib.ptr(part), ctypes.c_void_p(xptr), ldx,
                               ctypes.c_void_p(yptr), ldy, g.n_rows, d, None, _lib.stream_ptr())
    _lib.check(rc, "spmm")


def main():
    dev = torch.device("cuda:0")
    lib = _lib.load()
    scale = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    edges = int(sys.argv[2]) if len(sys.argv) > 2 else 10_000_000
    g = rmat_graph(scale, edges, seed=0, device=dev)
    gt = g.transpose()
    n, d = g.n_rows, 128
    X = torch.randn(n, d, device=dev)
    Y = torch.empty(n, d, device=dev)
    gb = g.algorithmic_bytes(d) / 1e9
    for name, gr in (("A", g), ("A^T", gt)):
        ref = ops.spmm(gr, X)
        t = timeit(lambda: ops.spmm(gr, X, out=Y), n=20)
        print("%-4s whole d=128                     %7.3f ms  %7.1f GB/s (B_alg)" % (name, t, gb / t * 1e3), flush=True)
        for w in (64, 32, 16):
            ns = d // w
            # (a) strided slabs of the row-major operand
            def strided():
                for j in range(ns):
                    raw_spmm(lib, gr, X.data_ptr() + 4 * w * j, d, Y.data_ptr() + 4 * w * j, d, w)
            Y.zero_(); strided()
            err = (Y - ref).abs().max().item()
            t = timeit(strided, n=20)
            print("%-4s %d strided slabs of %3d columns  %7.3f ms  %7.1f GB/s   (err %.1e)" % (name, ns, w, t, gb / t * 1e3, err), flush=True)
            # (b) slab-major operand and result
            Xs = X.view(n, ns, w).permute(1, 0, 2).contiguous()
            Ys = torch.empty_like(Xs)

            def slabmajor():
                for j in range(ns):
                    raw_spmm(lib, gr, Xs[j].data_ptr(), w, Ys[j].data_ptr(), w, w)
            slabmajor()
            err = (Ys.permute(1, 0, 2).reshape(n, d) - ref).abs().max().item()
            t = timeit(slabmajor, n=20)
            print("%-4s %d slab-major tables N x %3d     %7.3f ms  %7.1f GB/s   (err %.1e)" % (name, ns, w, t, gb / t * 1e3, err), flush=True)
            # (c) slab-major operand, row-major result
            def mixed():
                for j in range(ns):
                    raw_spmm(lib, gr, Xs[j].data_ptr(), w, Y.data_ptr() + 4 * w * j, d, w)
            t = timeit(mixed, n=20)
            print("%-4s %d slab-major in, row-major out  %7.3f ms  %7.1f GB/s" % (name, ns, t, gb / t * 1e3), flush=True)


if __name__ == "__main__":
    main()
