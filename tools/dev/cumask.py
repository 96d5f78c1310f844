#!/usr/bin/env python3
"""Experiment: CU-masked streams (hipExtStreamCreateWithCUMask) to run the HBM-bound SpMM chain beside the
MFMA/VALU-bound dense kernels on disjoint CU sets.  Development aid, not part of the product."""
import ctypes
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from graph_odenet_amd import ops  # noqa: E402
from graph_odenet_amd.synth import rmat_graph  # noqa: E402

hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(bits):
    """bits: list of CU indices enabled."""
    words = (ctypes.c_uint32 * 8)()
    for b in bits:
        words[b // 32] |= (1 << (b % 32))
    st = ctypes.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(st), 8, words)
    if rc != 0:
        raise RuntimeError("hipExtStreamCreateWithCUMask rc=%d" % rc)
    return torch.cuda.ExternalStream(st.value)


def timed(streams_fns, reps):
    """streams_fns: list of (stream, fn, count).  Launches all, returns wall ms."""
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for st, fn, cnt in streams_fns:
            with torch.cuda.stream(st):
                for _ in range(cnt):
                    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3 / reps


def main():
    dev = torch.device("cuda:0")
    g = rmat_graph(20, 10_000_000, seed=0, device=dev)
    n, d = g.n_rows, 128
    X, Y, K1, OUT, OUT2, DX = [torch.randn(n, d, device=dev) for _ in range(6)]
    bias = torch.randn(d, device=dev)
    W = torch.randn(d + 1, d, device=dev) / d ** 0.5
    gam, bet = torch.rand(d, device=dev) + 0.5, torch.rand(d, device=dev) - 0.5
    t2 = [(1.0, X), (0.1, K1)]

    def spmm():
        ops.spmm(g, X, bias=bias, relu=True, out=OUT)

    def gb():
        ops.gn_time_gemm_bwd(t2, n, d, 32, 1e-5, gam, W, True, Y, out=DX)

    def gf():
        ops.gn_time_gemm(t2, n, d, 32, 1e-5, gam, bet, W, True, 0.3, out=OUT2)

    def wg():
        ops.wgrad(t2, n, d, 32, 1e-5, gam, bet, Y, True)

    cur = torch.cuda.current_stream()
    for f in (spmm, gb, gf, wg):
        f()
    print("full chip: spmm %.3f  gb %.3f  gf %.3f  wg %.3f ms" % tuple(timed([(cur, f, 1)], 10) for f in (spmm, gb, gf, wg)), flush=True)
    print("serial spmm x2 + gb + gf + wg on one stream: %.3f ms" % timed([(cur, spmm, 2), (cur, gb, 1), (cur, gf, 1), (cur, wg, 1)], 10), flush=True)
    plain = torch.cuda.Stream()
    print("unmasked two streams (spmm x2 | gb+gf+wg): %.3f ms" % timed([(cur, spmm, 2), (plain, gb, 1), (plain, gf, 1), (plain, wg, 1)], 10), flush=True)

    # how does a mask map to CUs?  time spmm and gb under several 128-CU patterns
    pats = {"low128": list(range(128)), "even": list(range(0, 256, 2)), "quads": [b for b in range(256) if (b // 4) % 2 == 0],
            "per32_lo16": [b for b in range(256) if b % 32 < 16], "low64": list(range(64)), "low192": list(range(192)),
            "all": list(range(256))}
    for name, bits in pats.items():
        st = masked_stream(bits)
        with torch.cuda.stream(st):
            spmm(); gb()
        print("mask %-10s (%3d CUs): spmm %.3f  gb %.3f ms" % (name, len(bits), timed([(st, spmm, 1)], 6), timed([(st, gb, 1)], 6)), flush=True)

    # concurrency on disjoint sets: spmm chain on `a` CUs, dense chain on the rest
    for name, sel in (("even/odd", lambda b: b % 2 == 0), ("per32 20/12", lambda b: b % 32 < 20), ("per32 16/16", lambda b: b % 32 < 16),
                      ("per32 24/8", lambda b: b % 32 < 24), ("low160/high96", lambda b: b < 160), ("per8 5/3", lambda b: b % 8 < 5)):
        A = masked_stream([b for b in range(256) if sel(b)])
        B = masked_stream([b for b in range(256) if not sel(b)])
        with torch.cuda.stream(A):
            spmm()
        with torch.cuda.stream(B):
            gb(); gf(); wg()
        ta = timed([(A, spmm, 2)], 6)
        tb = timed([(B, gb, 1), (B, gf, 1), (B, wg, 1)], 6)
        both = timed([(A, spmm, 2), (B, gb, 1), (B, gf, 1), (B, wg, 1)], 6)
        print("split %-14s: spmm x2 alone %.3f | dense alone %.3f | together %.3f ms" % (name, ta, tb, both), flush=True)


if __name__ == "__main__":
    main()
